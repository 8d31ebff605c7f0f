// Wide-channel GEMMs and implicit-GEMM convolutions on the bf16 matrix cores (round 5).
//
// What they replace: the 3x3 / 1x1 convolutions of 128 ... 1024 channels and the qkvv / out_proj / fc1 / fc2 / reduction Linears of the
// UNETR++ and SwinUNETR configurations (config/CLI/model/unetrpp.yaml:19-35, swinunetr.yaml:19-30; classes taken from mfai at
// py4cast/models.py:10-20), which until round 4 ran as library calls (im2col + hipBLASLt GEMM + col2im, MIOpen batch norm).
//
//   gemm_nt_kernel   C[m][n] = sum_k A(m,k) B(n,k): both operands contiguous along k.  A = activation rows (Linear forward: x; data
//                    gradient: dy) or the im2col view of an NHWC map (3x3 "same" convolution: k = tap * Cin + ci, rows outside the
//                    image read as zeros through out-of-range buffer offsets -- no im2col buffer exists); B = a prepared bf16 image
//                    of the weight ([N][K]; the data gradient uses the transposed / tap-flipped image, so it is this same kernel).
//                    128 x 128 x 64 tiles, 4 waves of 64 x 64 (2 x 2 v_mfma_f32_32x32x16_bf16 accumulators), LDS tiles XOR-swizzled
//                    for conflict-free ds_read_b128 fragments, register-staged double buffering (loads of tile t+1 in flight during
//                    the products of tile t, written after them: one barrier per k-block), split-K over blockIdx.y for the deep
//                    stages (fp32 slabs, summed in a fixed order by gemm_nt_reduce_kernel).
//                    Epilogue (in the kernel when splits == 1, else in the reduce kernel; staged through LDS so that every store is
//                    a whole 16-byte piece of a row): + bias, GELU (saving the pre-activation) or GELU' (data gradient of fc2),
//                    + residual, bf16 rounding, per-column sums / sums of squares of the rounded output (batch-norm statistics).
//   gemm_tn_kernel   D[i][j] = sum_r P(r,i) Q(r,j): both operands STRIDED along the reduction index (rows / pixels) -- the weight
//                    gradient (P = dy, Q = x or its im2col view).  [64 r][128] LDS tiles with 320-byte rows, fragments by
//                    ds_read_b64_tr_b16 (conflict-free at that stride), split over r (fp32 slabs), bias gradient = column sums of P
//                    accumulated by the loader threads.  gemm_tn_reduce_kernel sums the slabs in a fixed order and writes the
//                    gradient in the canonical torch layout [co][ci][kh][kw].
// Every reduction has a fixed order: reruns are bit-identical.
#include "common.hpp"

namespace p4c {
namespace gemm {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr unsigned int OOB = 0x7fffffffu;
constexpr int ACT_NONE = 0, ACT_GELU_FWD = 1, ACT_GELU_BWD = 2;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ float bf_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void unpack8(u32x4 w, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = bf_lo(w[j]); v[2 * j + 1] = bf_hi(w[j]); }
}
// exact (erf) GELU, as torch.nn.functional.gelu's default
__device__ __forceinline__ float gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// ------------------------------------------------------------------------------------------------ epilogue (shared)
struct Epi {
    const float* bias;      // [N] or NULL
    const bf16* res;        // [M][ldr] residual or NULL
    int64_t ldr;
    const bf16* aux_in;     // ACT_GELU_BWD: pre-activation rows [M][ldaux]
    bf16* aux_out;          // ACT_GELU_FWD: where the pre-activation goes (same ld)
    int64_t ldaux;
    bf16* C;                // [M][ldc]
    int64_t ldc;
    float* stats;           // [row blocks][2][N] column sums / sums of squares of the rounded output, or NULL
    int act;
};

// one 8-wide piece of output row m at columns n .. n+7 (n a multiple of 8, all 8 inside N): v = accumulated products
__device__ __forceinline__ void epi_piece(const Epi& e, float (&v)[8], int64_t m, int n, float (&s1)[8], float (&s2)[8]) {
    if (e.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(e.bias + n), b1 = *reinterpret_cast<const f32x4*>(e.bias + n + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] += b0[j]; v[4 + j] += b1[j]; }
    }
    if (e.act == ACT_GELU_FWD) {
        u32x4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = pack2(v[2 * j], v[2 * j + 1]);
        *reinterpret_cast<u32x4*>(e.aux_out + m * e.ldaux + n) = h;
        float hv[8];
        unpack8(h, hv);       // GELU of the STORED pre-activation: what the backward differentiates
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = gelu(hv[j]);
    } else if (e.act == ACT_GELU_BWD) {
        float hv[8];
        unpack8(*reinterpret_cast<const u32x4*>(e.aux_in + m * e.ldaux + n), hv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= gelu_grad(hv[j]);
    }
    if (e.res) {
        float rv[8];
        unpack8(*reinterpret_cast<const u32x4*>(e.res + m * e.ldr + n), rv);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += rv[j];
    }
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack2(v[2 * j], v[2 * j + 1]);
    *reinterpret_cast<u32x4*>(e.C + m * e.ldc + n) = o;
    if (e.stats) {
        float ov[8];
        unpack8(o, ov);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += ov[j]; s2[j] = __builtin_fmaf(ov[j], ov[j], s2[j]); }
    }
}

// column sums of the 16 row groups (threadIdx.x >> 4) of a 256-thread block, thread's 8 columns = 8 * (threadIdx.x & 15) ..:
// fixed order, written to stats[blk][0 / 1][n0 + col].  red: 16 * 2 * 128 floats of LDS (free at this point).
__device__ __forceinline__ void stats_block_reduce(float* red, const float (&s1)[8], const float (&s2)[8], float* stats, int blk,
                                                   int n0, int N) {
    const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[(rg * 2 + 0) * 128 + ch * 8 + j] = s1[j];
        red[(rg * 2 + 1) * 128 + ch * 8 + j] = s2[j];
    }
    __syncthreads();
    const int which = threadIdx.x >> 7, col = threadIdx.x & 127;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[(g * 2 + which) * 128 + col];
    if (n0 + col < N) stats[((int64_t)blk * 2 + which) * N + n0 + col] = t;
}

// ------------------------------------------------------------------------------------------------ NT kernel
struct NtArgs {
    const bf16* A;          // activations (rows of lda elements; convolution: the NHWC map, lda = its channel count)
    const bf16* B;          // prepared weight image [N][ldb], k contiguous
    int64_t lda, ldb;
    unsigned int a_bytes, b_bytes;
    int M, N, K;
    int H, W, Cin, taps;    // taps == 9: 3x3 "same" convolution over (H, W) maps, K = 9 * Cin;  taps == 1: plain rows
    int tiles_m, tiles_n;
    int nkb, splits, kb_per_split;
    float* partial;         // splits > 1: [split][tile][128][128] fp32
    Epi e;
};

// byte offset of 16-byte chunk c (0..7) of row `row` inside a [128][64] bf16 tile (chunks XOR-swizzled: the 16 rows of one
// ds_read_b128 lane group fall on the 16 different 16-byte slots of the 256-byte bank row)
__device__ __forceinline__ int nt_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

template <bool CONV>
__global__ void __launch_bounds__(256, 2) gemm_nt_kernel(NtArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][A 16 KB | B 16 KB]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = blockIdx.x, tm = tile % a.tiles_m, tn = tile / a.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int kb0 = split * a.kb_per_split;
    int kb1 = kb0 + a.kb_per_split;
    if (kb1 > a.nkb) kb1 = a.nkb;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A, a.a_bytes), rsB = make_rsrc(a.B, a.b_bytes);

    // loader role: chunk c of rows lr + 32 it (it = 0..3) of both tiles
    const int c = tid & 7, lr = tid >> 3;
    unsigned int a_row[4], b_row[4];         // byte offsets of the rows (OOB: row outside the matrix)
    int py[4], px[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int m = m0 + lr + 32 * it, n = n0 + lr + 32 * it;
        a_row[it] = m < a.M ? (unsigned int)((int64_t)m * a.lda * 2) : OOB;
        b_row[it] = n < a.N ? (unsigned int)((int64_t)n * a.ldb * 2) : OOB;
        if (CONV) {
            const int p = m % (a.H * a.W);
            py[it] = p / a.W;
            px[it] = p - py[it] * a.W;
        }
    }
    int k = kb0 * BK + c * 8;                // this thread's k of the current block
    int tap = 0, ci = k;
    if (CONV) { tap = k / a.Cin; ci = k - tap * a.Cin; }

    u32x4 ra[4], rb[4];
    auto issue = [&]() __attribute__((always_inline)) {
        const bool kin = k < a.K;
        if (CONV) {
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int shift = (dy * a.W + dx) * (int)a.lda * 2 + ci * 2;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const bool ok = kin && a_row[it] != OOB && (unsigned)(py[it] + dy) < (unsigned)a.H && (unsigned)(px[it] + dx) < (unsigned)a.W;
                ra[it] = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? a_row[it] + shift : OOB, 0, 0);
            }
        } else {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                ra[it] = __builtin_amdgcn_raw_buffer_load_b128(rsA, (kin && a_row[it] != OOB) ? a_row[it] + k * 2 : OOB, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it)
            rb[it] = __builtin_amdgcn_raw_buffer_load_b128(rsB, (kin && b_row[it] != OOB) ? b_row[it] + k * 2 : OOB, 0, 0);
        k += BK;
        if (CONV) {
            ci += BK;
            while (ci >= a.Cin) { ci -= a.Cin; ++tap; }
        }
    };
    auto stash = [&](int buf) __attribute__((always_inline)) {
        char* ba = smem + buf * 32768;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            *reinterpret_cast<u32x4*>(ba + nt_off(lr + 32 * it, c)) = ra[it];
            *reinterpret_cast<u32x4*>(ba + 16384 + nt_off(lr + 32 * it, c)) = rb[it];
        }
    };

    // compute role: wave (wn, wm) owns n rows 64 wn .. +63 (MFMA A operand, i) x m rows 64 wm .. +63 (B operand, j)
    const int wn = wv & 1, wm = wv >> 1, r = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

    if (kb0 < kb1) {
        issue();
        stash(0);
        __syncthreads();
        for (int kb = kb0; kb < kb1; ++kb) {
            const int cur = (kb - kb0) & 1;
            const bool more = kb + 1 < kb1;
            if (more) issue();
            const char* ba = smem + cur * 32768;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 wf[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wf[i] = *reinterpret_cast<const bf16x8*>(ba + 16384 + nt_off(64 * wn + 32 * i + r, 2 * s + h));
                    xf[i] = *reinterpret_cast<const bf16x8*>(ba + nt_off(64 * wm + 32 * i + r, 2 * s + h));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
            if (more) stash(cur ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: two phases of 64 output rows through an fp32 LDS tile [64][132]
    float* stage = reinterpret_cast<float*>(smem);
    const int ch = tid & 15, rg = tid >> 4;
    const bool col_ok = n0 + ch * 8 < a.N;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    float* slab = a.splits > 1 ? a.partial + ((int64_t)split * gridDim.x + tile) * (BM * BN) : nullptr;
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        if (wm == ph) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        *reinterpret_cast<f32x4*>(stage + (32 * j + r) * 132 + 64 * wn + 32 * i + 8 * q + 4 * h) = v;
                    }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = rg + 16 * it;
            const int64_t m = m0 + 64 * ph + row;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(stage + row * 132 + ch * 8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(stage + row * 132 + ch * 8 + 4);
            if (slab) {
                *reinterpret_cast<f32x4*>(slab + (64 * ph + row) * BN + ch * 8) = v0;
                *reinterpret_cast<f32x4*>(slab + (64 * ph + row) * BN + ch * 8 + 4) = v1;
            } else if (col_ok && m < a.M) {
                float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                epi_piece(a.e, v, m, n0 + ch * 8, s1, s2);
            }
        }
        __syncthreads();
    }
    if (!slab && a.e.stats) stats_block_reduce(stage, s1, s2, a.e.stats, tm, n0, a.N);
}

// sums the split-K slabs of one 32-row band of a tile in split order, then the epilogue.  grid (tiles, 4)
struct NtRedArgs {
    const float* partial;
    int splits, tiles, tiles_m, M, N;
    Epi e;
};
__global__ void __launch_bounds__(256) gemm_nt_reduce_kernel(NtRedArgs a) {
    __shared__ float red[16 * 2 * 128];
    const int tile = blockIdx.x, band = blockIdx.y, tm = tile % a.tiles_m, tn = tile / a.tiles_m;
    const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int n = tn * BN + ch * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = 32 * band + rg + 16 * it;
        const int64_t m = (int64_t)tm * BM + row;
        if (n < a.N && m < a.M) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
            for (int s = 0; s < a.splits; ++s) {
                const float* p = a.partial + ((int64_t)s * a.tiles + tile) * (BM * BN) + row * BN + ch * 8;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] += v0[j]; v[4 + j] += v1[j]; }
            }
            epi_piece(a.e, v, m, n, s1, s2);
        }
    }
    if (a.e.stats) stats_block_reduce(red, s1, s2, a.e.stats, tm * 4 + band, tn * BN, a.N);
}

// ------------------------------------------------------------------------------------------------ TN kernel (weight gradients)
struct TnArgs {
    const bf16* P;          // dy rows [R][ldp]: output index i = its column
    const bf16* Q;          // x rows [R][ldq] (convolution: the NHWC map): output index j = tap * Cin + ci
    int64_t ldp, ldq;
    unsigned int p_bytes, q_bytes;
    int R, Mo, No;
    int H, W, Cin, taps;
    int tiles_i, tiles_j;
    int nrb, splits, rb_per_split;
    float* partial;         // [split][tile][128][128]
    float* bias_partial;    // [split][tiles_i][128] or NULL
};

constexpr int TN_ROW = 320;                  // bytes per LDS row of 128 bf16 (+ 64: the 4 rows of a transposed read hit 4 bank quarters)
constexpr int TN_TILE = 64 * TN_ROW;         // 20 480 B

template <bool CONV, bool BIAS>
__global__ void __launch_bounds__(256, 2) gemm_tn_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][P tile | Q tile]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile = blockIdx.x, ti = tile % a.tiles_i, tj = tile / a.tiles_i;
    const int i0 = ti * 128, j0 = tj * 128;
    const int split = blockIdx.y;
    const int rb0 = split * a.rb_per_split;
    int rb1 = rb0 + a.rb_per_split;
    if (rb1 > a.nrb) rb1 = a.nrb;

    const __amdgpu_buffer_rsrc_t rsP = make_rsrc(a.P, a.p_bytes), rsQ = make_rsrc(a.Q, a.q_bytes);
    // loader role: 16-byte chunk ch (8 columns) of rows lr + 16 it of both tiles
    const int ch = tid & 15, lr = tid >> 4;
    const bool p_ok = i0 + ch * 8 < a.Mo, q_ok = j0 + ch * 8 < a.No;
    int tap = 0, ci = j0 + ch * 8, dy = 0, dx = 0;
    if (CONV) {
        tap = ci / a.Cin;
        ci -= tap * a.Cin;
        dy = tap / 3 - 1;
        dx = tap - (tap / 3) * 3 - 1;
    }
    const unsigned int p_col = (unsigned int)(i0 + ch * 8) * 2;
    const int q_shift = CONV ? ((dy * a.W + dx) * (int)a.ldq + ci) * 2 : (j0 + ch * 8) * 2;
    float bs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bs[j] = 0.f;

    u32x4 rp[4], rq[4];
    auto issue = [&](int rb) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int rr = rb * 64 + lr + 16 * it;
            const bool in = rr < a.R;
            rp[it] = __builtin_amdgcn_raw_buffer_load_b128(rsP, (in && p_ok) ? (unsigned int)((int64_t)rr * a.ldp * 2) + p_col : OOB, 0, 0);
            bool ok = in && q_ok;
            if (CONV) {
                const int p = rr % (a.H * a.W);
                const int y = p / a.W, x = p - y * a.W;
                ok = ok && (unsigned)(y + dy) < (unsigned)a.H && (unsigned)(x + dx) < (unsigned)a.W;
            }
            rq[it] = __builtin_amdgcn_raw_buffer_load_b128(rsQ, ok ? (unsigned int)((int64_t)rr * a.ldq * 2 + q_shift) : OOB, 0, 0);
        }
    };
    auto stash = [&](int buf) __attribute__((always_inline)) {
        char* bp = smem + buf * (2 * TN_TILE);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            *reinterpret_cast<u32x4*>(bp + (lr + 16 * it) * TN_ROW + ch * 16) = rp[it];
            *reinterpret_cast<u32x4*>(bp + TN_TILE + (lr + 16 * it) * TN_ROW + ch * 16) = rq[it];
            if (BIAS) {
                float v[8];
                unpack8(rp[it], v);
#pragma unroll
                for (int j = 0; j < 8; ++j) bs[j] += v[j];
            }
        }
    };

    const int wi = wv & 1, wj = wv >> 1, h = lane >> 5, i16 = lane & 15, tg = (lane >> 4) & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
    // transposed-read address of this lane inside a tile, for k-step 0 / operand block 0
    const int tr_base = (8 * h + (i16 >> 2)) * TN_ROW + (tg * 16 + (i16 & 3) * 4) * 2;

    if (rb0 < rb1) {
        issue(rb0);
        stash(0);
        __syncthreads();
        for (int rb = rb0; rb < rb1; ++rb) {
            const int cur = (rb - rb0) & 1;
            const bool more = rb + 1 < rb1;
            if (more) issue(rb + 1);
            const char* bp = smem + cur * (2 * TN_TILE);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                bf16x8 pf[2], qf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    union { s16x4 v[2]; bf16x8 f; } up, uq;
                    const char* pp = bp + tr_base + 16 * s * TN_ROW + (64 * wi + 32 * i) * 2;
                    const char* qq = bp + TN_TILE + tr_base + 16 * s * TN_ROW + (64 * wj + 32 * i) * 2;
                    up.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pp));
                    up.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(pp + 4 * TN_ROW));
                    uq.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(qq));
                    uq.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(qq + 4 * TN_ROW));
                    pf[i] = up.f;
                    qf[i] = uq.f;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);
            }
            if (more) stash(cur ^ 1);
            __syncthreads();
        }
    }
    // slab [i][j]: accumulator register e of lane (r, h) is element (i = (e & 3) + 8 (e >> 2) + 4 h, j = r): 128-byte row pieces
    float* slab = a.partial + ((int64_t)split * gridDim.x + tile) * (128 * 128);
    const int r = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                slab[(64 * wi + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h) * 128 + 64 * wj + 32 * j + r] = acc[i][j][e];
    if (BIAS && tj == 0) {
        float* red = reinterpret_cast<float*>(smem);      // [16][128]
#pragma unroll
        for (int j = 0; j < 8; ++j) red[lr * 128 + ch * 8 + j] = bs[j];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += red[g * 128 + tid];
            a.bias_partial[((int64_t)split * a.tiles_i + ti) * 128 + tid] = t;
        }
    }
}

// dW[(i * Cin + ci) * taps + tap] = sum_s slab[s][tile(i, j)][i % 128][j % 128], j = tap * Cin + ci;  db[i] = sum_s bias_partial[s][..][i]
struct TnRedArgs {
    const float* partial;
    const float* bias_partial;
    float* dw;
    float* db;
    int splits, tiles, tiles_i, Mo, No, Cin, taps;
};
__global__ void __launch_bounds__(256) gemm_tn_reduce_kernel(TnRedArgs a) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;      // quad index over [Mo][No / 4]
    const int nq = a.No >> 2;
    if (q < (int64_t)a.Mo * nq) {
        const int i = (int)(q / nq), j = (int)(q - (int64_t)i * nq) * 4;
        const int tile = (j >> 7) * a.tiles_i + (i >> 7);
        const float* p = a.partial + (int64_t)tile * (128 * 128) + (i & 127) * 128 + (j & 127);
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < a.splits; ++s) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + (int64_t)s * a.tiles * (128 * 128));
            t += v;
        }
        if (a.taps == 1) {
            *reinterpret_cast<f32x4*>(a.dw + (int64_t)i * a.No + j) = t;
        } else {
            const int tap = j / a.Cin, ci = j - tap * a.Cin;         // Cin % 4 == 0: the quad stays inside one tap
#pragma unroll
            for (int e = 0; e < 4; ++e) a.dw[((int64_t)i * a.Cin + ci + e) * a.taps + tap] = t[e];
        }
    }
    if (a.db && q < a.Mo) {
        const int i = (int)q;
        float t = 0.f;
        for (int s = 0; s < a.splits; ++s) t += a.bias_partial[((int64_t)s * a.tiles_i + (i >> 7)) * 128 + (i & 127)];
        a.db[i] = t;
    }
}

// ------------------------------------------------------------------------------------------------ weight images
// fwd[co][tap][ci] = w[co][ci][tap] and dgrad[ci][tap'][co] = w[co][ci][taps - 1 - tap'] as bf16, from the fp32 master in the torch
// layout [CO][CI][taps]; one workgroup per 32 x 32 (co, ci) block, through LDS so that reads and writes are row pieces.
__global__ void __launch_bounds__(256) gemm_prep_kernel(const float* __restrict__ w, int CO, int CI, int taps, bf16* __restrict__ fwd,
                                                        bf16* __restrict__ dgrad) {
    extern __shared__ float tilew[];          // [32 co][32 ci * taps + 1]
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int rowlen = 32 * taps, ld = rowlen + 1;
    for (int idx = threadIdx.x; idx < 32 * rowlen; idx += 256) {
        const int c = idx / rowlen, e = idx - c * rowlen;            // e = ci_local * taps + tap
        const int ci = ci0 + e / taps;
        tilew[c * ld + e] = (co0 + c < CO && ci < CI) ? w[((int64_t)(co0 + c) * CI + ci0) * taps + e] : 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * rowlen; idx += 256) {
        {   // fwd: (co, tap, ci) with ci fastest
            const int ci = idx & 31, t = (idx >> 5) % taps, c = idx / (32 * taps);
            if (fwd && co0 + c < CO && ci0 + ci < CI)
                fwd[((int64_t)(co0 + c) * taps + t) * CI + ci0 + ci] = __float2bfloat16(tilew[c * ld + ci * taps + t]);
        }
        {   // dgrad: (ci, tap', co) with co fastest
            const int c = idx & 31, t = (idx >> 5) % taps, ci = idx / (32 * taps);
            if (dgrad && co0 + c < CO && ci0 + ci < CI)
                dgrad[((int64_t)(ci0 + ci) * taps + t) * CO + co0 + c] = __float2bfloat16(tilew[c * ld + ci * taps + (taps - 1 - t)]);
        }
    }
}

// batch-norm statistics from the producers' partial sums [nblk][2][C] (fp64 combine, fixed order): mean, rstd, scale = gamma rstd,
// shift = beta - mean scale, and the running statistics (momentum update, unbiased variance) as torch.nn.BatchNorm2d keeps them.
__global__ void __launch_bounds__(256) bnorm_finalize_kernel(const float* __restrict__ partial, int nblk, double count, int C,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                             float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                                             float* __restrict__ shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < nblk; ++b) {
        s1 += (double)partial[((int64_t)b * 2 + 0) * C + c];
        s2 += (double)partial[((int64_t)b * 2 + 1) * C + c];
    }
    const double mu = s1 / count;
    double var = s2 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    mean[c] = (float)mu;
    rstd[c] = rs;
    scale[c] = g * rs;
    shift[c] = b - (float)mu * g * rs;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

int pick_splits(int tiles, int nblocks) {
    // fill ~2 workgroups per CU; at least 4 k-blocks per split so that a slab's traffic stays below its products' time
    const int want = 2 * num_cus();
    int s = (want + tiles - 1) / tiles;
    if (s > nblocks / 4) s = nblocks / 4;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    return s;
}

}  // namespace gemm
}  // namespace p4c

using namespace p4c;
using namespace p4c::gemm;

extern "C" int p4c_gemm_prep_weight(const float* w, int CO, int CI, int taps, void* fwd, void* dgrad, p4c_stream_t stream) {
    P4C_CHECK_ARG(w && (fwd || dgrad), "p4c_gemm_prep_weight: NULL pointer");
    P4C_CHECK_ARG(CO > 0 && CI > 0 && (taps == 1 || taps == 9), "p4c_gemm_prep_weight: CO, CI > 0, taps 1 or 9");
    const int smem = 32 * (32 * taps + 1) * 4;
    hipLaunchKernelGGL(gemm_prep_kernel, dim3((CI + 31) / 32, (CO + 31) / 32), dim3(256), smem, as_stream(stream), w, CO, CI, taps, (bf16*)fwd,
                       (bf16*)dgrad);
    P4C_CHECK_LAUNCH("gemm_prep");
    return P4C_OK;
}

static int nt_plan(int M, int N, int K, int* tiles_m, int* tiles_n, int* nkb, int* splits, int* kbps) {
    *tiles_m = (M + BM - 1) / BM;
    *tiles_n = (N + BN - 1) / BN;
    *nkb = (K + BK - 1) / BK;
    int s = pick_splits(*tiles_m * *tiles_n, *nkb);
    *kbps = (*nkb + s - 1) / s;
    s = (*nkb + *kbps - 1) / *kbps;
    *splits = s;
    return 0;
}

extern "C" size_t p4c_gemm_nt_workspace_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int tm, tn, nkb, s, kbps;
    nt_plan(M, N, K, &tm, &tn, &nkb, &s, &kbps);
    return s > 1 ? (size_t)s * tm * tn * BM * BN * sizeof(float) : 0;
}

extern "C" int p4c_gemm_nt_stat_blocks(int M, int N, int K) {
    int tm, tn, nkb, s, kbps;
    nt_plan(M, N, K, &tm, &tn, &nkb, &s, &kbps);
    return s > 1 ? 4 * tm : tm;
}

// C = epilogue(A x Bimg^T).  taps == 1: A = (M, K) rows with row stride lda.  taps == 9: A = an NHWC map (batch, H, W, Cin) with
// M = batch * H * W pixels and pixel stride lda (>= Cin), K = 9 * Cin: the 3x3 "same" convolution (zero padding).
extern "C" int p4c_gemm_nt(const void* A, int64_t lda, const void* Bimg, int M, int N, int K, int H, int W, int Cin, int taps,
                           const float* bias, const void* res, int64_t ldr, int act, const void* aux_in, void* aux_out, int64_t ldaux,
                           void* C, int64_t ldc, float* stats, void* workspace, p4c_stream_t stream) {
    P4C_CHECK_ARG(A && Bimg && C, "p4c_gemm_nt: NULL pointer");
    P4C_CHECK_ARG(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, "p4c_gemm_nt: M=%d N=%d K=%d (N, K multiples of 8)", M, N, K);
    P4C_CHECK_ARG(taps == 1 || taps == 9, "p4c_gemm_nt: taps must be 1 or 9");
    P4C_CHECK_ARG(lda % 8 == 0 && ldc % 8 == 0 && ldc >= N, "p4c_gemm_nt: row strides must be multiples of 8 (ldc >= N)");
    if (taps == 9) {
        P4C_CHECK_ARG(H > 0 && W > 0 && Cin > 0 && Cin % 8 == 0 && K == 9 * Cin && M % (H * W) == 0 && lda >= Cin,
                      "p4c_gemm_nt: convolution needs K = 9 Cin, Cin a multiple of 8, M a multiple of H W");
    } else {
        P4C_CHECK_ARG(lda >= K, "p4c_gemm_nt: lda < K");
    }
    P4C_CHECK_ARG(act == ACT_NONE || (act == ACT_GELU_FWD && aux_out) || (act == ACT_GELU_BWD && aux_in), "p4c_gemm_nt: activation / aux mismatch");
    P4C_CHECK_ARG(!res || ldr % 8 == 0, "p4c_gemm_nt: residual row stride must be a multiple of 8");
    P4C_CHECK_ARG(act == ACT_NONE || ldaux % 8 == 0, "p4c_gemm_nt: aux row stride must be a multiple of 8");
    const int64_t a_bytes = (int64_t)M * lda * 2, b_bytes = (int64_t)N * K * 2;
    P4C_CHECK_ARG(a_bytes < 0x7fffffffLL && b_bytes < 0x7fffffffLL, "p4c_gemm_nt: operands beyond 2 GiB");
    NtArgs a;
    a.A = (const bf16*)A; a.B = (const bf16*)Bimg; a.lda = lda; a.ldb = K;
    a.a_bytes = (unsigned int)a_bytes; a.b_bytes = (unsigned int)b_bytes;
    a.M = M; a.N = N; a.K = K; a.H = H; a.W = W; a.Cin = Cin; a.taps = taps;
    nt_plan(M, N, K, &a.tiles_m, &a.tiles_n, &a.nkb, &a.splits, &a.kb_per_split);
    P4C_CHECK_ARG(a.splits == 1 || workspace, "p4c_gemm_nt: this shape runs split-K: workspace of p4c_gemm_nt_workspace_bytes required");
    a.partial = (float*)workspace;
    a.e.bias = bias; a.e.res = (const bf16*)res; a.e.ldr = ldr; a.e.aux_in = (const bf16*)aux_in; a.e.aux_out = (bf16*)aux_out;
    a.e.ldaux = ldaux; a.e.C = (bf16*)C; a.e.ldc = ldc; a.e.stats = stats; a.e.act = act;
    const int tiles = a.tiles_m * a.tiles_n, smem = 65536;
    hipStream_t st = as_stream(stream);
    if (taps == 9) {
        P4C_TRY(ensure_dyn_smem((const void*)gemm_nt_kernel<true>, smem));
        hipLaunchKernelGGL((gemm_nt_kernel<true>), dim3(tiles, a.splits), dim3(256), smem, st, a);
    } else {
        P4C_TRY(ensure_dyn_smem((const void*)gemm_nt_kernel<false>, smem));
        hipLaunchKernelGGL((gemm_nt_kernel<false>), dim3(tiles, a.splits), dim3(256), smem, st, a);
    }
    P4C_CHECK_LAUNCH("gemm_nt");
    if (a.splits > 1) {
        NtRedArgs r{(const float*)workspace, a.splits, tiles, a.tiles_m, M, N, a.e};
        hipLaunchKernelGGL(gemm_nt_reduce_kernel, dim3(tiles, 4), dim3(256), 0, st, r);
        P4C_CHECK_LAUNCH("gemm_nt_reduce");
    }
    return P4C_OK;
}

static int tn_plan(int R, int Mo, int No, int* ti, int* tj, int* nrb, int* splits, int* rbps) {
    *ti = (Mo + 127) / 128;
    *tj = (No + 127) / 128;
    *nrb = (R + 63) / 64;
    int s = pick_splits(*ti * *tj, *nrb);
    *rbps = (*nrb + s - 1) / s;
    *splits = (*nrb + *rbps - 1) / *rbps;
    return 0;
}

extern "C" size_t p4c_gemm_tn_workspace_bytes(int R, int Mo, int No) {
    if (R <= 0 || Mo <= 0 || No <= 0) return 0;
    int ti, tj, nrb, s, rbps;
    tn_plan(R, Mo, No, &ti, &tj, &nrb, &s, &rbps);
    return ((size_t)s * ti * tj * 128 * 128 + (size_t)s * ti * 128) * sizeof(float);
}

// dW (Mo, Cin, taps) fp32 = sum over the R rows of dy^T (x) [x or its 3x3 im2col view], db (Mo) = column sums of dy (or NULL)
extern "C" int p4c_gemm_tn(const void* dy, int64_t ldp, const void* x, int64_t ldq, int R, int Mo, int H, int W, int Cin, int taps,
                           float* dw, float* db, void* workspace, p4c_stream_t stream) {
    P4C_CHECK_ARG(dy && x && dw && workspace, "p4c_gemm_tn: NULL pointer");
    P4C_CHECK_ARG(R > 0 && Mo > 0 && Cin > 0 && Mo % 8 == 0 && Cin % 8 == 0 && (taps == 1 || taps == 9), "p4c_gemm_tn: R=%d Mo=%d Cin=%d taps=%d", R, Mo,
                  Cin, taps);
    P4C_CHECK_ARG(ldp % 8 == 0 && ldq % 8 == 0 && ldp >= Mo && ldq >= Cin, "p4c_gemm_tn: row strides must be multiples of 8 covering the rows");
    P4C_CHECK_ARG(taps == 1 || (H > 0 && W > 0 && R % (H * W) == 0), "p4c_gemm_tn: convolution needs R a multiple of H W");
    const int64_t p_bytes = (int64_t)R * ldp * 2, q_bytes = (int64_t)R * ldq * 2;
    P4C_CHECK_ARG(p_bytes < 0x7fffffffLL && q_bytes < 0x7fffffffLL, "p4c_gemm_tn: operands beyond 2 GiB");
    TnArgs a;
    a.P = (const bf16*)dy; a.Q = (const bf16*)x; a.ldp = ldp; a.ldq = ldq; a.p_bytes = (unsigned int)p_bytes; a.q_bytes = (unsigned int)q_bytes;
    a.R = R; a.Mo = Mo; a.No = taps * Cin; a.H = H; a.W = W; a.Cin = Cin; a.taps = taps;
    tn_plan(R, Mo, a.No, &a.tiles_i, &a.tiles_j, &a.nrb, &a.splits, &a.rb_per_split);
    const int tiles = a.tiles_i * a.tiles_j;
    a.partial = (float*)workspace;
    a.bias_partial = db ? (float*)workspace + (size_t)a.splits * tiles * 128 * 128 : nullptr;
    const int smem = 2 * 2 * TN_TILE;
    hipStream_t st = as_stream(stream);
#define P4C_TN_LAUNCH(CV, BS)                                                                         \
    do {                                                                                              \
        P4C_TRY(ensure_dyn_smem((const void*)gemm_tn_kernel<CV, BS>, smem));                          \
        hipLaunchKernelGGL((gemm_tn_kernel<CV, BS>), dim3(tiles, a.splits), dim3(256), smem, st, a);  \
    } while (0)
    if (taps == 9) { if (db) P4C_TN_LAUNCH(true, true); else P4C_TN_LAUNCH(true, false); }
    else { if (db) P4C_TN_LAUNCH(false, true); else P4C_TN_LAUNCH(false, false); }
#undef P4C_TN_LAUNCH
    P4C_CHECK_LAUNCH("gemm_tn");
    TnRedArgs r{a.partial, a.bias_partial, dw, db, a.splits, tiles, a.tiles_i, Mo, a.No, Cin, taps};
    const int64_t quads = (int64_t)Mo * (a.No / 4);
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, r);
    P4C_CHECK_LAUNCH("gemm_tn_reduce");
    return P4C_OK;
}

extern "C" int p4c_bnorm_finalize(const float* partial, int nblk, double count, int C, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, float* mean, float* rstd, float* scale,
                                  float* shift, p4c_stream_t stream) {
    P4C_CHECK_ARG(partial && mean && rstd && scale && shift && nblk > 0 && C > 0 && count > 0, "p4c_bnorm_finalize: bad arguments");
    hipLaunchKernelGGL(bnorm_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, as_stream(stream), partial, nblk, count, C, gamma, beta, eps,
                       momentum, running_mean, running_var, mean, rstd, scale, shift);
    P4C_CHECK_LAUNCH("bnorm_finalize");
    return P4C_OK;
}
