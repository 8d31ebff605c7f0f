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
//                    128 x 128 x 64 tiles, 8 waves (two per SIMD) of 64 x 32 outputs (2 v_mfma_f32_32x32x16_bf16 accumulators),
//                    LDS tiles XOR-swizzled for conflict-free ds_read_b128 fragments.  A 4-stage LDS ring is filled by direct-to-LDS
//                    loads (buffer_load ... lds, the swizzle applied to the SOURCE address, zeros for out-of-range offsets), one 1-KiB
//                    piece issued after each k-step's products, k-blocks t+1 .. t+3 in flight while t is multiplied, counted
//                    s_waitcnt vmcnt + one s_barrier per k-block.  Split-K over blockIdx.y for the deep stages (fp32 slabs, summed in
//                    a fixed order by gemm_nt_reduce_kernel).
//                    Epilogue (in the kernel when splits == 1, else in the reduce kernel; staged through LDS so that every store is
//                    a whole 16-byte piece of a row): + bias, GELU (saving the pre-activation) or GELU' (data gradient of fc2),
//                    + residual, bf16 rounding, per-column sums / sums of squares of the rounded output (batch-norm statistics).
//                    Measured (profiles/r05_gemm_micro.txt): 3x3 conv 128 -> 128 on 2 x 128 x 128 in 20 us (470 TFLOP/s; the library
//                    route it replaces: 260 us); the loop runs at the rate the L2 -> LDS path delivers 32 KiB per k-block and CU
//                    (~0.6 us: 13 TB/s chip-wide), not at the matrix pipe's (0.25 us) -- larger tiles / input-row reuse are the
//                    open item.
//   gemm_tn_kernel   D[i][j] = sum_r P(r,i) Q(r,j): both operands STRIDED along the reduction index (rows / pixels) -- the weight
//                    gradient (P = dy, Q = x or its im2col view).  Same ring; [64 r][128] tiles with 256-byte rows whose 16-byte
//                    chunks are XOR-swizzled by the row so that ds_read_b64_tr_b16 fragments are conflict-free; split over r (fp32
//                    slabs); bias gradient = column sums of P on the matrix pipe (a block of ones as the other operand, spread over
//                    the four j-waves).  gemm_tn_reduce_kernel sums the slabs in a fixed order, turns (tap, ci) -> (ci, tap) through
//                    LDS and writes -- or adds into the parameter's .grad -- the gradient in the torch layout [co][ci][kh][kw].
// Every reduction has a fixed order: reruns are bit-identical.
#include <mutex>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

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

// column sums of the NRG row groups (threadIdx.x >> 4) of a 16 NRG-thread block, thread's 8 columns = 8 * (threadIdx.x & 15) ..:
// fixed order, written to stats[blk][0 / 1][n0 + col].  red: 16 * 2 * 128 floats of LDS (free at this point).
template <int NRG>
__device__ __forceinline__ void stats_block_reduce(float* red, const float (&s1)[8], const float (&s2)[8], float* stats, int blk,
                                                   int n0, int N) {
    const int ch = threadIdx.x & 15, rg = threadIdx.x >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[(rg * 2 + 0) * 128 + ch * 8 + j] = s1[j];
        red[(rg * 2 + 1) * 128 + ch * 8 + j] = s2[j];
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        const int which = threadIdx.x >> 7, col = threadIdx.x & 127;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < NRG; ++g) t += red[(g * 2 + which) * 128 + col];
        if (n0 + col < N) stats[((int64_t)blk * 2 + which) * N + n0 + col] = t;
    }
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

#ifndef P4C_NT_EXP
#define P4C_NT_EXP 0      // timing experiments of diagnostic builds only (results become wrong): 1 no epilogue, 2 no products, 4 no A loads, 8 no B loads
#endif
constexpr int NT_STAGES = 4;                 // LDS ring: 4 x (A 16 KB + B 16 KB); k-blocks t+1 .. t+3 in flight while t is multiplied
constexpr int NT_STAGE_BYTES = 32768;
typedef __attribute__((address_space(3))) char* lds_ptr;

// One direct-to-LDS load (buffer_load_dwordx4 ... lds): 64 lanes x 16 bytes land at lds_addr + 16 lane (lds_addr wave-uniform); an
// out-of-range offset lands as zeros.  Inline asm ON PURPOSE: given the builtin, hipcc (ROCm 7.2) orders every later ds_read behind
// the transfer with s_waitcnt vmcnt(0) -- the whole ring drained before each k-step (the round's first version ran that way: 41 us
// instead of 20 for the stage-0 weight gradient).  The kernels count these loads themselves (s_waitcnt vmcnt(N) + s_barrier before the
// reads).  M0 is saved and restored around the statement (the compiler owns it).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned int voff, unsigned int lds_addr) {
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ unsigned int lds_address(const void* p) { return (unsigned int)(size_t)((lds_ptr)p); }

template <bool CONV>
__global__ void __launch_bounds__(512, 2) gemm_nt_kernel(NtArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [4][A 16 KB | B 16 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Tile order.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2); for the convolution a 128-row M
    // tile is one image row at W = 128 whose two halo rows belong to the neighbouring tiles, so with tile = blockIdx.x every input row
    // is pulled into three L2s (PMC: 3 x the map fetched, r05_pmc_traffic_gemm_nt.json).  Round 6 measured the bijective XCD remap of
    // cdna_hip_programming.md (T1: each XCD a CONTIGUOUS band of tiles, a row shared inside one L2) -- P4C_NT_EXP bit 32 builds it --
    // and it is SLOWER or equal on every bench shape (profiles/r06_ab_runs.txt 4: 128 -> 128 at 128^2 21.0 vs 20.7 us, 256 -> 256 at 64^2
    // 29.3 vs 24.4, 192 -> 96 at 64^2 20.4 vs 17.1; UNETR++ step 138.1 vs 138.2 ms): the loop is not limited by the bytes L2 fetches
    // (they come from the 256 MB Infinity Cache at these sizes) but by the L2 -> LDS delivery per CU, and 32 neighbouring workgroups on
    // one XCD reading the same rows in lock step queue on the same L2 channels.  The round-robin order stays.
    int tile = blockIdx.x;
    if (P4C_NT_EXP & 32) {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (int)(blockIdx.x >> 3);
    }
    const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int split = blockIdx.y;
    const int kb0 = split * a.kb_per_split;
    int kb1 = kb0 + a.kb_per_split;
    if (kb1 > a.nkb) kb1 = a.nkb;
    const int nblk = kb1 - kb0;

    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(a.A, (P4C_NT_EXP & 4) ? 0u : a.a_bytes), rsB = make_rsrc(a.B, (P4C_NT_EXP & 8) ? 0u : a.b_bytes);
    const unsigned int smem_lds = lds_address(smem);

    // loader role (every wave).  Both tiles go to LDS by direct loads (buffer_load ... lds): one wave instruction fills 1 KiB = 8
    // tile rows, lane -> (row 8 q + (lane >> 3), 16-byte slot lane & 7); the slot holds chunk slot ^ ((row >> 1) & 7) of the row
    // (nt_off), so the swizzle is applied to the SOURCE address.  Wave wv issues pieces q = 2 wv + it (it = 0, 1) of both tiles, one
    // piece after each k-step's products (the matrix pipe works on them while the piece is issued); out-of-range offsets (rows beyond
    // the matrix, k beyond K, the convolution's zero padding) land as zeros.
    unsigned int a_row[2], b_row[2];         // byte offsets of the rows (OOB: row outside the matrix)
    unsigned int vmask[2];                   // convolution: bit t = tap t of this row's pixel reads inside the image
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = 16 * wv + 8 * it + (lane >> 3);
        const int m = m0 + row, n = n0 + row;
        a_row[it] = m < a.M ? (unsigned int)((int64_t)m * a.lda * 2) : OOB;
        b_row[it] = n < a.N ? (unsigned int)((int64_t)n * a.ldb * 2) : OOB;
        vmask[it] = 0u;
        if (CONV && m < a.M) {
            const int p = m % (a.H * a.W);
            const int y = p / a.W, x = p - y * a.W;
#pragma unroll
            for (int t = 0; t < 9; ++t)
                if ((unsigned)(y + t / 3 - 1) < (unsigned)a.H && (unsigned)(x + t % 3 - 1) < (unsigned)a.W) vmask[it] |= 1u << t;
        }
    }
    // the chunk a lane fetches: (lane & 7) ^ (4 it + (lane >> 4))
    // k state of the lane's chunk, advanced by 64 per k-block; convolution: (tap, ci) of k and the byte shift of that tap's source
    // pixel + channel -- recomputed only when ci wraps into the next tap (no division, no tap arithmetic in the steady state)
    int kq[2], tap[2], ci[2], shift[2];
    auto tap_shift = [&](int t, int c) __attribute__((always_inline)) {
        const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;
        return ((dy * a.W + dx) * (int)a.lda + c) * 2;
    };
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c = (lane & 7) ^ ((it << 2) | (lane >> 4));
        kq[it] = kb0 * BK + c * 8;
        tap[it] = 0;
        ci[it] = kq[it];
        shift[it] = 0;
        if (CONV) {
            tap[it] = kq[it] / a.Cin;
            ci[it] = kq[it] - tap[it] * a.Cin;
            shift[it] = tap_shift(tap[it], ci[it]);
        }
    }
    // piece 0: A it 0, 1: B it 0, 2: A it 1, 3: B it 1 of the k-block the lane's k state points at
    auto issue_piece = [&](int stage, int piece) __attribute__((always_inline)) {
        const unsigned int base = smem_lds + stage * NT_STAGE_BYTES + wv * 2048;
        const int it = piece >> 1;
        const bool kin = kq[it] < a.K;
        if ((piece & 1) == 0) {
            unsigned int oa;
            if (CONV) {
                const bool ok = kin && ((vmask[it] >> tap[it]) & 1u);
                oa = ok ? a_row[it] + shift[it] : OOB;
            } else {
                oa = (kin && a_row[it] != OOB) ? a_row[it] + kq[it] * 2 : OOB;
            }
            if (!(P4C_NT_EXP & 16)) dma16(rsA, oa, base + it * 1024); else asm volatile("" :: "v"(oa));
        } else {
            const unsigned int ob = (kin && b_row[it] != OOB) ? b_row[it] + kq[it] * 2 : OOB;
            if (!(P4C_NT_EXP & 16)) dma16(rsB, ob, base + 16384 + it * 1024); else asm volatile("" :: "v"(ob));
            kq[it] += BK;             // both pieces of this `it` are out: on to the next k-block
            if (CONV) {
                ci[it] += BK;
                shift[it] += 2 * BK;
                if (ci[it] >= a.Cin) {
                    do { ci[it] -= a.Cin; ++tap[it]; } while (ci[it] >= a.Cin);
                    shift[it] = tap_shift(tap[it], ci[it]);
                }
            }
        }
    };

    // compute role: wave (wn, wm) owns n rows 64 wn .. +63 (MFMA A operand, i) x m rows 32 wm .. +31 (B operand, j)
    const int wn = wv & 1, wm = wv >> 1, r = lane & 31, h = lane >> 5;
    f32x16 acc[2];
    acc[0] = zero16();
    acc[1] = zero16();

#pragma unroll
    for (int p = 0; p < NT_STAGES - 1; ++p)
        if (p < nblk) {
#pragma unroll
            for (int piece = 0; piece < 4; ++piece) issue_piece(p, piece);
        }
    for (int t = 0; t < nblk; ++t) {
        // block t has landed once at most the pieces of the (up to two) younger blocks are outstanding: 4 direct loads per block and wave
        const int younger = nblk - 1 - t;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // every wave's pieces of block t are in LDS; every wave is done reading block t - 1
        asm volatile("" ::: "memory");
        const bool more = t + NT_STAGES - 1 < nblk;
        const int nstage = (t + NT_STAGES - 1) & (NT_STAGES - 1);     // the stage block t - 1 occupied
        const char* ba = smem + (t & (NT_STAGES - 1)) * NT_STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 xf = *reinterpret_cast<const bf16x8*>(ba + nt_off(32 * wm + r, 2 * s + h));
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(ba + 16384 + nt_off(64 * wn + r, 2 * s + h));
            const bf16x8 w1 = *reinterpret_cast<const bf16x8*>(ba + 16384 + nt_off(64 * wn + 32 + r, 2 * s + h));
            if (!(P4C_NT_EXP & 2)) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xf, acc[1], 0, 0, 0);
            } else {
                asm volatile("" :: "v"(w0), "v"(w1), "v"(xf));
            }
            if (more) issue_piece(nstage, s);
        }
    }
    __syncthreads();        // the staging tile below reuses stage 0
    if ((P4C_NT_EXP & 1) && acc[0][0] != 12345.678f) return;

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
        if ((wm >> 1) == ph) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
                    *reinterpret_cast<f32x4*>(stage + (32 * (wm & 1) + r) * 132 + 64 * wn + 32 * i + 8 * q + 4 * h) = v;
                }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = rg + 32 * it;
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
    if (!slab && a.e.stats) stats_block_reduce<32>(stage, s1, s2, a.e.stats, tm, n0, a.N);
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
    if (a.e.stats) stats_block_reduce<16>(red, s1, s2, a.e.stats, tm * 4 + band, tn * BN, a.N);
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
    float* bias_partial;    // [split][tiles_i][4][128] or NULL
};

#ifndef P4C_TN_EXP
#define P4C_TN_EXP 0      // timing experiments of diagnostic builds only (results become wrong): 1 no slab stores, 2 no products, 4 no loads
#endif
constexpr int TN_STAGES = 4;
constexpr int TN_STAGE_BYTES = 32768;        // P tile [64 r][128] bf16 (16 KB) | Q tile (16 KB), 256-byte rows
// 16-byte chunk ch (0..15) of tile row `row` sits in slot ch ^ tn_swz(row): the four rows of a transposed read (ds_read_b64_tr_b16)
// and the two 16-column blocks of a 32-lane half then cover the 256-byte bank row exactly once (conflict-free)
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <bool CONV, bool BIAS>
__global__ void __launch_bounds__(512, 2) gemm_tn_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [4][P tile | Q tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, ti = tile % a.tiles_i, tj = tile / a.tiles_i;
    const int i0 = ti * 128, j0 = tj * 128;
    const int split = blockIdx.y;
    const int rb0 = split * a.rb_per_split;
    int rb1 = rb0 + a.rb_per_split;
    if (rb1 > a.nrb) rb1 = a.nrb;
    const int nblk = rb1 - rb0;

    const __amdgpu_buffer_rsrc_t rsP = make_rsrc(a.P, (P4C_TN_EXP & 4) ? 0u : a.p_bytes), rsQ = make_rsrc(a.Q, (P4C_TN_EXP & 4) ? 0u : a.q_bytes);
    const unsigned int smem_lds = lds_address(smem);
    // loader role (every wave): direct loads to LDS, one wave instruction = 1 KiB = 4 tile rows; lane -> (row 4 q + (lane >> 4), slot
    // lane & 15) fetches chunk slot ^ tn_swz(row).  Wave wv issues pieces q = 2 wv + it of both tiles.
    // Per piece the lane keeps the byte offsets of its row in P and Q and, for a convolution, the pixel (y, x) of that row: all of it
    // advances by 64 rows per k-block without a division or a 64-bit product in the loop.
    int rrow[2], py[2], px[2];
    unsigned int p_off[2], q_off[2];
    int qdy[2], qdx[2];
    bool p_ok[2], q_ok[2];
    const unsigned int p_step = (unsigned int)(64 * a.ldp * 2), q_step = (unsigned int)(64 * a.ldq * 2);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = 8 * wv + 4 * it + (lane >> 4);
        const int chk = (lane & 15) ^ tn_swz(row);
        const int rr = rb0 * 64 + row;
        rrow[it] = rr;
        p_ok[it] = i0 + chk * 8 < a.Mo;
        q_ok[it] = j0 + chk * 8 < a.No;
        p_off[it] = (unsigned int)((int64_t)rr * a.ldp * 2) + (unsigned int)(i0 + chk * 8) * 2;
        qdy[it] = 0;
        qdx[it] = 0;
        py[it] = 0;
        px[it] = 0;
        if (CONV) {
            int col = j0 + chk * 8;
            const int tap = col / a.Cin;
            col -= tap * a.Cin;
            qdy[it] = tap / 3 - 1;
            qdx[it] = tap - (tap / 3) * 3 - 1;
            q_off[it] = (unsigned int)((int64_t)rr * a.ldq * 2 + ((qdy[it] * a.W + qdx[it]) * (int)a.ldq + col) * 2);
            const int p = rr % (a.H * a.W);
            py[it] = p / a.W;
            px[it] = p - py[it] * a.W;
        } else {
            q_off[it] = (unsigned int)((int64_t)rr * a.ldq * 2) + (unsigned int)(j0 + chk * 8) * 2;
        }
    }
    auto issue_piece = [&](int stage, int piece) __attribute__((always_inline)) {
        const unsigned int base = smem_lds + stage * TN_STAGE_BYTES + wv * 2048;
        const int it = piece >> 1;
        const bool in = rrow[it] < a.R;
        if ((piece & 1) == 0) {
            dma16(rsP, (in && p_ok[it]) ? p_off[it] : OOB, base + it * 1024);
        } else {
            bool ok = in && q_ok[it];
            if (CONV) ok = ok && (unsigned)(py[it] + qdy[it]) < (unsigned)a.H && (unsigned)(px[it] + qdx[it]) < (unsigned)a.W;
            dma16(rsQ, ok ? q_off[it] : OOB, base + 16384 + it * 1024);
            rrow[it] += 64;
            p_off[it] += p_step;
            q_off[it] += q_step;
            if (CONV) {
                px[it] += 64;
                while (px[it] >= a.W) { px[it] -= a.W; ++py[it]; }
                while (py[it] >= a.H) py[it] -= a.H;
            }
        }
    };

    // compute role: wave (wi, wj) owns output rows i 64 wi .. +63 (MFMA A operand) x columns j 32 wj .. +31 (B operand)
    const int wi = wv & 1, wj = wv >> 1, h = lane >> 5, i16 = lane & 15, tg = (lane >> 4) & 1;
    const int q4 = i16 >> 2, p4 = i16 & 3;
    // transposed-read byte address inside a tile for k-step 0: rows 8 h + q4 (+ 4 for the second read), the 16 columns of block cb
    auto tr_addr = [&](int cb, int sec) __attribute__((always_inline)) {
        const int chunk = (cb >> 3) + 2 * tg + (p4 >> 1);
        return 256 * (8 * h + q4 + 4 * sec) + 16 * (chunk ^ ((q4 << 2) | ((2 * h + sec) & 3))) + 8 * (p4 & 1);
    };
    int pa[2][2], qa[2];
#pragma unroll
    for (int sec = 0; sec < 2; ++sec) {
        pa[0][sec] = tr_addr(64 * wi, sec);
        pa[1][sec] = tr_addr(64 * wi + 32, sec);
        qa[sec] = 16384 + tr_addr(32 * wj, sec);
    }
    f32x16 acc[2], accb[2];
    acc[0] = zero16(); acc[1] = zero16(); accb[0] = zero16(); accb[1] = zero16();
    const bool do_bias = BIAS && tj == 0;     // wave wj sums P's columns over the rows of k-step s == wj: four partial sums per column
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    auto tr2 = [&](const char* p) __attribute__((always_inline)) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    };
#pragma unroll
    for (int p = 0; p < TN_STAGES - 1; ++p)
        if (p < nblk) {
#pragma unroll
            for (int piece = 0; piece < 4; ++piece) issue_piece(p, piece);
        }
    for (int t = 0; t < nblk; ++t) {
        const int younger = nblk - 1 - t;
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = t + TN_STAGES - 1 < nblk;
        const int nstage = (t + TN_STAGES - 1) & (TN_STAGES - 1);
        const char* bp = smem + (t & (TN_STAGES - 1)) * TN_STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            union { s16x4 v[2]; bf16x8 f; } u0, u1, uq;
            u0.v[0] = tr2(bp + pa[0][0] + 4096 * s);
            u0.v[1] = tr2(bp + pa[0][1] + 4096 * s);
            u1.v[0] = tr2(bp + pa[1][0] + 4096 * s);
            u1.v[1] = tr2(bp + pa[1][1] + 4096 * s);
            uq.v[0] = tr2(bp + qa[0] + 4096 * s);
            uq.v[1] = tr2(bp + qa[1] + 4096 * s);
            if (!(P4C_TN_EXP & 2)) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u0.f, uq.f, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u1.f, uq.f, acc[1], 0, 0, 0);
            } else {
                asm volatile("" :: "v"(u0.f), "v"(u1.f), "v"(uq.f));
            }
            if (do_bias && s == wj) {      // column sums of P on the matrix pipe: times a block of ones
                accb[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u0.f, ones, accb[0], 0, 0, 0);
                accb[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u1.f, ones, accb[1], 0, 0, 0);
            }
            if (more) issue_piece(nstage, s);
        }
    }
    // slab [i][j]: accumulator register e of lane (r, h) is element (i = (e & 3) + 8 (e >> 2) + 4 h, j = r): 128-byte row pieces
    float* slab = a.partial + ((int64_t)split * gridDim.x + tile) * (128 * 128);
    const int r = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (!(P4C_TN_EXP & 1) || acc[i][e] == 12345.678f)
                slab[(64 * wi + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h) * 128 + 32 * wj + r] = acc[i][e];
    if (do_bias && r == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                a.bias_partial[(((int64_t)split * a.tiles_i + ti) * 4 + wj) * 128 + 64 * wi + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h] = accb[i][e];
    }
}

// dW[(i * Cin + ci) * taps + tap] = sum_s slab[s][tile(i, j)][i % 128][j % 128], j = tap * Cin + ci;  db[i] = sum over (s, wj) of
// bias_partial.  One workgroup = output row i x a chunk of <= 512 input channels: the slab pieces are read along j (coalesced, four
// splits in flight, summed in split order), turned through LDS and written along the torch layout's (ci, tap) order (coalesced).
struct TnRedArgs {
    const float* partial;
    const float* bias_partial;
    float* dw;
    float* db;
    int splits, tiles, tiles_i, Mo, No, Cin, taps, cchunk, accumulate;
};
__device__ __forceinline__ void tn_reduce_block(const TnRedArgs& a, const int bx, const int i, float* __restrict__ turn, float* __restrict__ part) {
    const int c0 = bx * a.cchunk;
    const int cn = a.Cin - c0 < a.cchunk ? a.Cin - c0 : a.cchunk;
    const int ne = a.taps * cn;
    const int64_t sstride = (int64_t)a.tiles * (128 * 128);
    // P threads per element (a power of two) when the workgroup has fewer elements than threads: thread p sums the splits
    // s = p (mod P) in order, the P sums are added in the order p = 0 .. P-1 -- a fixed order whatever the launch geometry
    int P = 1;
    while (P < 16 && 2 * P * ne <= 256 && 2 * P <= a.splits) P *= 2;
    for (int e0 = 0; e0 < ne; e0 += 256 / P) {
        const int e = e0 + (int)threadIdx.x / P, p = (int)threadIdx.x % P;
        float t = 0.f;
        if (e < ne) {
            const int tap = e / cn, c = e - tap * cn;
            const int j = tap * a.Cin + c0 + c;
            const float* q = a.partial + ((int64_t)(j >> 7) * a.tiles_i + (i >> 7)) * (128 * 128) + (i & 127) * 128 + (j & 127);
            int s = p;
            for (; s + 3 * P < a.splits; s += 4 * P) {
                const float v0 = q[s * sstride], v1 = q[(s + P) * sstride], v2 = q[(s + 2 * P) * sstride], v3 = q[(s + 3 * P) * sstride];
                t = (((t + v0) + v1) + v2) + v3;
            }
            for (; s < a.splits; s += P) t += q[s * sstride];
        }
        if (P > 1) {
            part[threadIdx.x] = t;
            __syncthreads();
            if (p == 0 && e < ne) {
                for (int k = 1; k < P; ++k) t += part[threadIdx.x + k];
            }
            __syncthreads();
        }
        if (p == 0 && e < ne) {
            const int tap = e / cn, c = e - tap * cn;
            turn[tap * cn + c] = t;
        }
    }
    __syncthreads();
    float* out = a.dw + ((int64_t)i * a.Cin + c0) * a.taps;
    for (int e = threadIdx.x; e < ne; e += 256) {
        const int c = e / a.taps, tap = e - c * a.taps;
        const float v = turn[tap * cn + c];
        out[e] = a.accumulate ? out[e] + v : v;
    }
    if (a.db && bx == 0 && threadIdx.x < 64) {
        // the 4 * splits partial column sums: lane l takes the entries k = l (mod 64) in order, then a fixed shuffle tree
        float t = 0.f;
        for (int k = threadIdx.x; k < 4 * a.splits; k += 64)
            t += a.bias_partial[(((int64_t)(k >> 2) * a.tiles_i + (i >> 7)) * 4 + (k & 3)) * 128 + (i & 127)];
        t = wave_sum(t);
        if (threadIdx.x == 0) a.db[i] = a.accumulate ? a.db[i] + t : t;
    }
}

__global__ void __launch_bounds__(256) gemm_tn_reduce_kernel(TnRedArgs a) {
    __shared__ float turn[9 * 512];
    __shared__ float part[256];
    tn_reduce_block(a, (int)blockIdx.x, (int)blockIdx.y, turn, part);
}

// Round 6: the reductions of one backward pass that ADD into gradient buffers, TN_BATCH per launch (the queue below): UNETR++ issued 804
// of them per optimizer step, 8 us each -- every one a grid of its own for a few hundred KB of slabs.  block -> (job, block of the job's
// own (channel chunk, output row) grid) through a prefix table in the kernel arguments; a job's arithmetic is tn_reduce_block's, so a
// batched and a separate reduction give the same bits.
constexpr int TN_BATCH = 32;
struct TnBatchArgs {
    TnRedArgs job[TN_BATCH];
    int first[TN_BATCH + 1];
    int n;
};
static_assert(sizeof(TnBatchArgs) <= 4000, "the job table travels in the kernel arguments");

__global__ void __launch_bounds__(256) gemm_tn_reduce_batch_kernel(TnBatchArgs b) {
    __shared__ float turn[9 * 512];
    __shared__ float part[256];
    int jb = 0;
    while (jb + 1 < b.n && (int)blockIdx.x >= b.first[jb + 1]) ++jb;
    const TnRedArgs& a = b.job[jb];
    const int local = (int)blockIdx.x - b.first[jb];
    const int gx = (a.Cin + a.cchunk - 1) / a.cchunk;
    tn_reduce_block(a, local % gx, local / gx, turn, part);
}

// ------------------------------------------------------------------------------------------------ weight images
// fwd[co][tap][ci] = w[co][ci][tap] and dgrad[ci][tap'][co] = w[co][ci][taps - 1 - tap'] as bf16, from the fp32 master in the torch
// layout [CO][CI][taps]; one workgroup per 32 x 32 (co, ci) block, through LDS so that reads and writes are row pieces.
// rowscale (CO floats or NULL): the images hold rowscale[co] * w[co] (fp32 product, one rounding) -- a per-output-channel scale of the
// layer (UNETR++'s layer scale gamma) folded into its weight; bias_out[co] = rowscale[co] * bias[co] comes with it (block (0, y)).
__device__ __forceinline__ void prep_block(const float* __restrict__ w, int CO, int CI, int taps, bf16* __restrict__ fwd,
                                           bf16* __restrict__ dgrad, const float* __restrict__ rowscale, const float* __restrict__ bias,
                                           float* __restrict__ bias_out, const int bx, const int by, float* __restrict__ tilew) {
    const int co0 = by * 32, ci0 = bx * 32;
    const int rowlen = 32 * taps, ld = rowlen + 1;
    for (int idx = threadIdx.x; idx < 32 * rowlen; idx += 256) {
        const int c = idx / rowlen, e = idx - c * rowlen;            // e = ci_local * taps + tap
        const int ci = ci0 + e / taps;
        float v = (co0 + c < CO && ci < CI) ? w[((int64_t)(co0 + c) * CI + ci0) * taps + e] : 0.f;
        if (rowscale && co0 + c < CO) v *= rowscale[co0 + c];
        tilew[c * ld + e] = v;
    }
    if (bias_out && bx == 0 && threadIdx.x < 32 && co0 + threadIdx.x < CO)
        bias_out[co0 + threadIdx.x] = (rowscale ? rowscale[co0 + threadIdx.x] : 1.f) * bias[co0 + threadIdx.x];
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

__global__ void __launch_bounds__(256) gemm_prep_kernel(const float* __restrict__ w, int CO, int CI, int taps, bf16* __restrict__ fwd,
                                                        bf16* __restrict__ dgrad, const float* __restrict__ rowscale,
                                                        const float* __restrict__ bias, float* __restrict__ bias_out) {
    extern __shared__ float tilew[];          // [32 co][32 ci * taps + 1]
    prep_block(w, CO, CI, taps, fwd, dgrad, rowscale, bias, bias_out, (int)blockIdx.x, (int)blockIdx.y, tilew);
}

// Round 6: the images of PREP_BATCH weights per launch (UNETR++ prepares 154 per optimizer step, SwinUNETR 54 -- a launch of ~12 us each
// for a few hundred KB): block -> (job, block of the job's own (ci, co) grid) through a prefix table in the kernel arguments.
constexpr int PREP_BATCH = 24;
struct WPrepJob {
    const float* w;
    const float* rowscale;
    const float* bias;
    float* bias_out;
    bf16* fwd;
    bf16* dgrad;
    int CO, CI, taps;
};
struct PrepBatchArgs {
    WPrepJob job[PREP_BATCH];
    int first[PREP_BATCH + 1];
    int n;
};
static_assert(sizeof(PrepBatchArgs) <= 4000, "the job table travels in the kernel arguments");

__global__ void __launch_bounds__(256) gemm_prep_batch_kernel(PrepBatchArgs b) {
    extern __shared__ float tilew[];
    int jb = 0;
    while (jb + 1 < b.n && (int)blockIdx.x >= b.first[jb + 1]) ++jb;
    const WPrepJob& J = b.job[jb];
    const int local = (int)blockIdx.x - b.first[jb];
    const int gx = (J.CI + 31) / 32;
    prep_block(J.w, J.CO, J.CI, J.taps, J.fwd, J.dgrad, J.rowscale, J.bias, J.bias_out, local % gx, local / gx, tilew);
}

// batch-norm statistics from the producers' partial sums [nblk][2][C] (fp64 combine, fixed order): mean, rstd, scale = gamma rstd,
// shift = beta - mean scale, and the running statistics (momentum update, unbiased variance) as torch.nn.BatchNorm2d keeps them.
__global__ void __launch_bounds__(256) bnorm_finalize_kernel(const float* __restrict__ partial, int nblk, double count, int C,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                             float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                                             float* __restrict__ shift, long long* __restrict__ batches_tracked) {
    // (batches_tracked: torch.nn.BatchNorm2d.num_batches_tracked, incremented here instead of by a launch of its own)
    if (batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1;
    // a workgroup owns 32 channels; its 8 thread rows take the partial blocks b = row (mod 8), then the rows are added in order
    __shared__ double red[2][8][32];
    const int cl = threadIdx.x & 31, row = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        int b = row;
        for (; b + 24 < nblk; b += 32) {          // four blocks' loads in flight (256 blocks at the first stage: 8 rounds, not 32)
            const float a0 = partial[((int64_t)b * 2 + 0) * C + c], q0 = partial[((int64_t)b * 2 + 1) * C + c];
            const float a1 = partial[((int64_t)(b + 8) * 2 + 0) * C + c], q1 = partial[((int64_t)(b + 8) * 2 + 1) * C + c];
            const float a2 = partial[((int64_t)(b + 16) * 2 + 0) * C + c], q2 = partial[((int64_t)(b + 16) * 2 + 1) * C + c];
            const float a3 = partial[((int64_t)(b + 24) * 2 + 0) * C + c], q3 = partial[((int64_t)(b + 24) * 2 + 1) * C + c];
            s1 = (((s1 + (double)a0) + (double)a1) + (double)a2) + (double)a3;
            s2 = (((s2 + (double)q0) + (double)q1) + (double)q2) + (double)q3;
        }
        for (; b < nblk; b += 8) {
            s1 += (double)partial[((int64_t)b * 2 + 0) * C + c];
            s2 += (double)partial[((int64_t)b * 2 + 1) * C + c];
        }
    }
    red[0][row][cl] = s1;
    red[1][row][cl] = s2;
    __syncthreads();
    if (row != 0 || c >= C) return;
#pragma unroll
    for (int r = 1; r < 8; ++r) { s1 += red[0][r][cl]; s2 += red[1][r][cl]; }
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
    // one workgroup per CU (the kernels' LDS rings fill it); at least 4 k-blocks per split so that a slab's traffic stays below
    // its products' time
    const int want = num_cus();
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

extern "C" int p4c_gemm_prep_weight_scaled(const float* w, const float* rowscale, const float* bias, float* bias_out, int CO, int CI,
                                           int taps, void* fwd, void* dgrad, p4c_stream_t stream) {
    P4C_CHECK_ARG(w && (fwd || dgrad), "p4c_gemm_prep_weight: NULL pointer");
    P4C_CHECK_ARG(CO > 0 && CI > 0 && (taps == 1 || taps == 9), "p4c_gemm_prep_weight: CO, CI > 0, taps 1 or 9");
    P4C_CHECK_ARG((bias == nullptr) == (bias_out == nullptr), "p4c_gemm_prep_weight_scaled: bias and bias_out come together");
    const int smem = 32 * (32 * taps + 1) * 4;
    hipLaunchKernelGGL(gemm_prep_kernel, dim3((CI + 31) / 32, (CO + 31) / 32), dim3(256), smem, as_stream(stream), w, CO, CI, taps, (bf16*)fwd,
                       (bf16*)dgrad, rowscale, bias, bias_out);
    P4C_CHECK_LAUNCH("gemm_prep");
    return P4C_OK;
}

extern "C" int p4c_gemm_prep_weight(const float* w, int CO, int CI, int taps, void* fwd, void* dgrad, p4c_stream_t stream) {
    return p4c_gemm_prep_weight_scaled(w, nullptr, nullptr, nullptr, CO, CI, taps, fwd, dgrad, stream);
}

// n jobs given as parallel arrays (rowscale / bias / bias_out entries may be NULL; or the three arrays themselves): the same images as
// n calls of p4c_gemm_prep_weight_scaled, PREP_BATCH jobs per launch
extern "C" int p4c_gemm_prep_weight_batch(int n, const float* const* w, const float* const* rowscale, const float* const* bias,
                                          float* const* bias_out, const int* CO, const int* CI, const int* taps, void* const* fwd,
                                          void* const* dgrad, p4c_stream_t stream) {
    P4C_CHECK_ARG(n > 0 && w && CO && CI && taps && fwd && dgrad, "p4c_gemm_prep_weight_batch: NULL pointer or no job");
    hipStream_t st = as_stream(stream);
    PrepBatchArgs b;
    b.n = 0;
    b.first[0] = 0;
    int smem = 0;
    auto launch = [&]() -> int {
        hipLaunchKernelGGL(gemm_prep_batch_kernel, dim3(b.first[b.n]), dim3(256), smem, st, b);
        P4C_CHECK_LAUNCH("gemm_prep_batch");
        b.n = 0;
        smem = 0;
        return P4C_OK;
    };
    for (int i = 0; i < n; ++i) {
        P4C_CHECK_ARG(w[i] && (fwd[i] || dgrad[i]) && CO[i] > 0 && CI[i] > 0 && (taps[i] == 1 || taps[i] == 9), "p4c_gemm_prep_weight_batch: job %d: bad arguments", i);
        const float* rs = rowscale ? rowscale[i] : nullptr;
        const float* bi = bias ? bias[i] : nullptr;
        float* bo = bias_out ? bias_out[i] : nullptr;
        P4C_CHECK_ARG((bi == nullptr) == (bo == nullptr), "p4c_gemm_prep_weight_batch: job %d: bias and bias_out come together", i);
        if (b.n == PREP_BATCH) P4C_TRY(launch());
        b.job[b.n] = WPrepJob{w[i], rs, bi, bo, (bf16*)fwd[i], (bf16*)dgrad[i], CO[i], CI[i], taps[i]};
        b.first[b.n + 1] = b.first[b.n] + ((CI[i] + 31) / 32) * ((CO[i] + 31) / 32);
        const int need = 32 * (32 * taps[i] + 1) * 4;
        if (need > smem) smem = need;
        ++b.n;
    }
    if (b.n) P4C_TRY(launch());
    return P4C_OK;
}

// Backward of a layer whose weight and bias were scaled per output channel (z = x W^T + b, y = gamma (.) z): from the RAW gradients
// dW_raw = dy^T x and db_raw = sum dy (p4c_gemm_tn on the unscaled dy):
//   dW = gamma (.) dW_raw,   db = gamma (.) db_raw,   dgamma[c] = <dW_raw[c,:], W[c,:]> + b[c] db_raw[c]     (= sum_r dy[r,c] z[r,c])
// one workgroup per output channel, fixed-order sums; accumulate: add into dw / db / dgamma (the parameters' .grad) instead of writing.
namespace p4c { namespace gemm {
__global__ void __launch_bounds__(256) scale_fold_bwd_kernel(const float* __restrict__ dw_raw, const float* __restrict__ db_raw,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             const float* __restrict__ gamma, int K, float* __restrict__ dw,
                                                             float* __restrict__ db, float* __restrict__ dgamma, int accumulate) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    const float g = gamma[c];
    float dot = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float r = dw_raw[(int64_t)c * K + k];
        dot += r * w[(int64_t)c * K + k];
        const float v = g * r;
        dw[(int64_t)c * K + k] = accumulate ? dw[(int64_t)c * K + k] + v : v;
    }
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = (red[0] + red[1]) + (red[2] + red[3]);
        const float dbr = db_raw ? db_raw[c] : 0.f;
        if (b) t += b[c] * dbr;
        dgamma[c] = accumulate ? dgamma[c] + t : t;
        if (db) db[c] = accumulate ? db[c] + g * dbr : g * dbr;
    }
}
}}  // namespace p4c::gemm

extern "C" int p4c_gemm_scale_fold_bwd(const float* dw_raw, const float* db_raw, const float* w, const float* b, const float* gamma, int CO,
                                       int K, float* dw, float* db, float* dgamma, int accumulate, p4c_stream_t stream) {
    P4C_CHECK_ARG(dw_raw && w && gamma && dw && dgamma, "p4c_gemm_scale_fold_bwd: NULL pointer");
    P4C_CHECK_ARG((b == nullptr) == (db_raw == nullptr) && (b == nullptr) == (db == nullptr),
                  "p4c_gemm_scale_fold_bwd: b, db_raw and db come together");
    P4C_CHECK_ARG(CO > 0 && K > 0, "p4c_gemm_scale_fold_bwd: CO, K > 0");
    hipLaunchKernelGGL(p4c::gemm::scale_fold_bwd_kernel, dim3(CO), dim3(256), 0, as_stream(stream), dw_raw, db_raw, w, b, gamma, K, dw, db,
                       dgamma, accumulate);
    P4C_CHECK_LAUNCH("gemm_scale_fold_bwd");
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
    const int tiles = a.tiles_m * a.tiles_n, smem = NT_STAGES * NT_STAGE_BYTES;
    hipStream_t st = as_stream(stream);
    if (taps == 9) {
        P4C_TRY(ensure_dyn_smem((const void*)gemm_nt_kernel<true>, smem));
        hipLaunchKernelGGL((gemm_nt_kernel<true>), dim3(tiles, a.splits), dim3(512), smem, st, a);
    } else {
        P4C_TRY(ensure_dyn_smem((const void*)gemm_nt_kernel<false>, smem));
        hipLaunchKernelGGL((gemm_nt_kernel<false>), dim3(tiles, a.splits), dim3(512), smem, st, a);
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
    return ((size_t)s * ti * tj * 128 * 128 + (size_t)s * ti * 4 * 128) * sizeof(float);
}

// ---- the queue of accumulating reductions (kernels.hpp: tn_reduce_*; flushed by p4c_grad_reduce_flush at the end of a backward pass)
namespace {
struct TnPending {
    TnRedArgs job;
    hipStream_t stream;
};
std::mutex g_tn_mu;
std::vector<TnPending> g_tn_pending;

int tn_launch_batch(const TnBatchArgs& b, hipStream_t st) {
    hipLaunchKernelGGL(gemm_tn_reduce_batch_kernel, dim3(b.first[b.n]), dim3(256), 0, st, b);
    P4C_CHECK_LAUNCH("gemm_tn_reduce_batch");
    return P4C_OK;
}
}  // namespace

namespace p4c {
int tn_reduce_pending() {
    std::lock_guard<std::mutex> lk(g_tn_mu);
    return (int)g_tn_pending.size();
}
void tn_reduce_drop() {
    std::lock_guard<std::mutex> lk(g_tn_mu);
    g_tn_pending.clear();
}
// the queued jobs in submission order, TN_BATCH per launch; a job that adds into a buffer a job of the launch being assembled already adds
// into (the same weight in the next AR step of a rollout) starts a new launch: additions into one element happen in submission order
int tn_reduce_flush(hipStream_t st) {
    std::vector<TnPending> jobs;
    {
        std::lock_guard<std::mutex> lk(g_tn_mu);
        jobs.swap(g_tn_pending);
    }
    if (jobs.empty()) return P4C_OK;
    TnBatchArgs b;
    b.n = 0;
    b.first[0] = 0;
    for (const TnPending& pj : jobs) {
        P4C_CHECK_ARG(pj.stream == st, "p4c_grad_reduce_flush: a queued weight-gradient reduction was produced on another stream");
        bool clash = false;
        for (int q = 0; q < b.n && !clash; ++q)
            clash = pj.job.dw == b.job[q].dw || (pj.job.db && pj.job.db == b.job[q].db);
        const int blocks = ((pj.job.Cin + pj.job.cchunk - 1) / pj.job.cchunk) * pj.job.Mo;
        if (b.n == TN_BATCH || clash || (b.n && (int64_t)b.first[b.n] + blocks > (1 << 20))) {
            P4C_TRY(tn_launch_batch(b, st));
            b.n = 0;
        }
        b.job[b.n] = pj.job;
        b.first[b.n + 1] = b.first[b.n] + blocks;
        ++b.n;
    }
    if (b.n) P4C_TRY(tn_launch_batch(b, st));
    return P4C_OK;
}
}  // namespace p4c

// dW (Mo, Cin, taps) fp32 = sum over the R rows of dy^T (x) [x or its 3x3 im2col view], db (Mo) = column sums of dy (or NULL)
extern "C" int p4c_gemm_tn(const void* dy, int64_t ldp, const void* x, int64_t ldq, int R, int Mo, int H, int W, int Cin, int taps,
                           float* dw, float* db, int accumulate, void* workspace, p4c_stream_t stream) {
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
    const int smem = TN_STAGES * TN_STAGE_BYTES;
    hipStream_t st = as_stream(stream);
#define P4C_TN_LAUNCH(CV, BS)                                                                         \
    do {                                                                                              \
        P4C_TRY(ensure_dyn_smem((const void*)gemm_tn_kernel<CV, BS>, smem));                          \
        hipLaunchKernelGGL((gemm_tn_kernel<CV, BS>), dim3(tiles, a.splits), dim3(512), smem, st, a);  \
    } while (0)
    if (taps == 9) { if (db) P4C_TN_LAUNCH(true, true); else P4C_TN_LAUNCH(true, false); }
    else { if (db) P4C_TN_LAUNCH(false, true); else P4C_TN_LAUNCH(false, false); }
#undef P4C_TN_LAUNCH
    P4C_CHECK_LAUNCH("gemm_tn");
    // input channels per workgroup of the reduction: as many as keep >= ~1024 workgroups (a workgroup's splits are summed serially)
    int cchunk = 512;
    while (cchunk > 32 && (int64_t)((Cin + cchunk - 1) / cchunk) * Mo < 1024) cchunk >>= 1;
    if (cchunk > Cin) cchunk = Cin;
    TnRedArgs r{a.partial, a.bias_partial, dw, db, a.splits, tiles, a.tiles_i, Mo, a.No, Cin, taps, cchunk, accumulate ? 1 : 0};
    if (accumulate && grad_reduce_deferring()) {
        // (the caller keeps the workspace alive until p4c_grad_reduce_flush: py4cast_amd.ops_nodeproj.GradQueue)
        std::lock_guard<std::mutex> lk(g_tn_mu);
        g_tn_pending.push_back(TnPending{r, st});
        return P4C_OK;
    }
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((Cin + cchunk - 1) / cchunk, Mo), dim3(256), 0, st, r);
    P4C_CHECK_LAUNCH("gemm_tn_reduce");
    return P4C_OK;
}

extern "C" int p4c_bnorm_finalize(const float* partial, int nblk, double count, int C, const float* gamma, const float* beta, float eps,
                                  float momentum, float* running_mean, float* running_var, float* mean, float* rstd, float* scale,
                                  float* shift, int64_t* num_batches_tracked, p4c_stream_t stream) {
    P4C_CHECK_ARG(partial && mean && rstd && scale && shift && nblk > 0 && C > 0 && count > 0, "p4c_bnorm_finalize: bad arguments");
    hipLaunchKernelGGL(bnorm_finalize_kernel, dim3((C + 31) / 32), dim3(256), 0, as_stream(stream), partial, nblk, count, C, gamma, beta, eps,
                       momentum, running_mean, running_var, mean, rstd, scale, shift, reinterpret_cast<long long*>(num_batches_tracked));
    P4C_CHECK_LAUNCH("bnorm_finalize");
    return P4C_OK;
}
