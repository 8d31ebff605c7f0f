// Fused row MLP of the mesh-GNN models:  y = LayerNorm(W2 SiLU(W1 x + b1 [+ a[ia] + b[ib]]) + b2) [+ res]   on R rows, hidden = out = 64.
// Every MLP of GraphLam / HiLAM is this shape (make_mlp: Linear - SiLU - Linear - LayerNorm, hidden_layers 1, hidden_dims 64:
// config/CLI/model/graphlam.yaml:21-22; the networks come from mfai, py4cast/models.py:10-20); on edges the first Linear is
// distributed over cat[e, x_s[src], x_r[dst]] (see graph.hip), which is the optional gathered addend here.  The library path runs it
// as 2 GEMMs + SiLU + LayerNorm + add (and 6 more passes backward), each a full stream of 0.5-2 M rows through HBM; here a row is
// read once and written once per direction:
//   * orientation "weights x rows": A = W (M = output feature), B = x^T (N = row): a lane loads 16 bytes of ITS row straight from
//     HBM as the B operand (no LDS staging of activations); the accumulator has the features of a row on the registers of one lane
//     pair, so SiLU, LayerNorm statistics (per-lane sums + one exchange between lane halves) and the residual are register math;
//   * the first product's accumulator, rounded to bf16, IS the B operand of the second (cdna_hip_programming.md "An accumulator tile
//     as the next MFMA's operand"); weights are re-laid once per workgroup into LDS operand images (the second layer's in the permuted
//     k order that idiom needs);
//   * backward recomputes the forward from x (nothing saved but x), chains dz -> dh -> dpre -> dx through the same idiom with the
//     transposed weight images, and forms dW1, dW2, db1, db2 from [row][feature] LDS images of (x, h, dz, dpre) with transposed reads
//     (ds_read_b64_tr_b16) -- the reduction index of a weight gradient is the row; 64 x K + 64 x 64 accumulators stay in registers
//     over a persistent wave's rows; all parameter gradients are reduced in a fixed order (no atomics).
#include "kernels.hpp"

namespace p4c {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr int HID = 64;
constexpr int PROW = HID * 2 + 16;   // LDS row stride of a [row][64 features] bf16 image (padded: transposed reads spread over banks)

struct MlpArgs {
    const bf16* x;          // (R, K) rows, K = 16 * KS (zero-padded features)
    const float* w1;        // first Linear weight [64][>= Kreal], row stride ldw1 (may be a column slice of a wider matrix)
    int ldw1, Kreal;
    const float* b1;        // [64] or NULL
    const float* w2;        // [Oreal][64] contiguous
    const float* b2;        // [Oreal] or NULL
    int Oreal;              // real output features (<= 64; rows beyond are zero)
    const float* gamma;     // LayerNorm weight / bias [64]; NULL = no LayerNorm
    const float* beta;
    float eps;
    const bf16* ga;         // gathered addends to the pre-activation: ga[ia[r]] + gb[ib[r]] (rows of 64) or NULL
    const int32_t* ia;
    const bf16* gb;
    const int32_t* ib;
    const bf16* res;        // residual rows (R, 64) or NULL
    bf16* out;              // y (R, 64) or NULL
    bf16* out_res;          // y + res (R, 64) or NULL
    // backward
    const bf16* dy;         // gradient of out or NULL
    const bf16* dy_res;     // gradient of out_res or NULL
    bf16* dx;               // (R, K) or NULL
    bf16* dpre;             // gradient of the pre-activation (R, 64): the gradient of ga / gb rows before their segment sums; or NULL
    float* partial;         // per-workgroup parameter-gradient partials
    const void* prepared;   // constants + weight operand images as laid out in LDS (p4c_row_mlp_prepare), or NULL
    int dx_add_dyr;         // backward, K = 64: dx += dy_res (res IS x: both gradients of the one tensor in one store)
    int64_t R;
};

__device__ __forceinline__ int rowmap(int i) { return (i & 3) + 8 * (i >> 2); }

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)0.f;
    return r;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ bf16x8 acc_op(const float* x, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)x[8 * s + j];
    return r;
}
__device__ __forceinline__ f32x4 load4(const bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4(bf16* p, float a, float b, float c, float d) {
    bf16x4 o;
    o[0] = (__bf16)a; o[1] = (__bf16)b; o[2] = (__bf16)c; o[3] = (__bf16)d;
    *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ void lds_order() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}
// natural-order transposed operand of k-step ks (rows 16 ks .. +15) from a [row][feature] image with row stride `rs` bytes:
// lane (feature 32 tile + (lane & 31), h) receives rows 16 ks + 8 h + j, j = 0..7
__device__ __forceinline__ bf16x8 read_tr_nat(const char* img, int rs, int ks, int tile, int lane) {
    const int i = lane & 15, tg = (lane >> 4) & 1, h = lane >> 5;
    const char* p = img + (16 * ks + 8 * h + (i >> 2)) * rs + (32 * tile + tg * 16 + (i & 3) * 4) * 2;
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * rs));
    return u.v;
}

// sum over the 8 lanes that share (lane & 7), in every lane
__device__ __forceinline__ float lane8_sum(float v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ float silu_sig(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-x * LOG2E)); }

// ---- LDS weight operand images: image[(tile * S + s) * 64 + lane] = 8 bf16 (16 B)
// natural k order:  element j = M[row 32 tile + (lane & 31)][col 16 s + 8 h + j]
// permuted k order: element j = M[row 32 tile + (lane & 31)][col 16 s + 8 (j >> 2) + 4 h + (j & 3)]   (matches acc_op)
// M(row, col) = transposed ? W[col][row] : W[row][col], zero outside (rows_real, cols_real)
__device__ __forceinline__ void build_image(bf16* img, int tiles, int S, const float* W, int ld, int rows_real, int cols_real,
                                            bool transposed, bool permuted, int tid, int nthreads) {
    const int total = tiles * S * 64 * 8;
    for (int idx = tid; idx < total; idx += nthreads) {
        const int j = idx & 7, lane = (idx >> 3) & 63, ts = idx >> 9;
        const int s = ts % S, tile = ts / S;
        const int h = lane >> 5;
        const int row = 32 * tile + (lane & 31);
        const int col = permuted ? 16 * s + 8 * (j >> 2) + 4 * h + (j & 3) : 16 * s + 8 * h + j;
        float v = 0.f;
        if (row < rows_real && col < cols_real) v = transposed ? W[(int64_t)col * ld + row] : W[(int64_t)row * ld + col];
        img[idx] = (bf16)v;
    }
}
__device__ __forceinline__ bf16x8 wop(const bf16* img, int S, int tile, int s, int lane) {
    return *reinterpret_cast<const bf16x8*>(img + ((tile * S + s) * 64 + lane) * 8);
}

// constants: [b1 | b2 | gamma | beta] x 64 floats
__device__ __forceinline__ void build_consts(float* lc, const MlpArgs& a, int tid, int nthreads) {
    for (int i = tid; i < 4 * HID; i += nthreads) {
        const int which = i >> 6, c = i & 63;
        float v = 0.f;
        if (which == 0 && a.b1) v = a.b1[c];
        if (which == 1 && a.b2 && c < a.Oreal) v = a.b2[c];
        if (which == 2) v = a.gamma ? a.gamma[c] : 1.f;
        if (which == 3 && a.beta) v = a.beta[c];
        lc[i] = v;
    }
}


// constants + weight images into LDS: a 16-byte-per-thread copy of the prepared blob, or built from the raw fp32 parameters.
// The copy is split into its loads (all in flight at once: ONE memory round trip -- as a `dst[i] = src[i]` loop over a run-time
// trip count it was 5 / 9 dependent round trips, ~4 / ~7 us of a small-grid launch that computes for ~4) and its LDS stores, so that
// a kernel can issue its first tile's row loads in between.  Workgroups are 256 threads.
template <int KS, bool BWD>
struct StagedParams {
    static constexpr int NKT = (16 * KS + 31) / 32;
    static constexpr int bytes = 4 * HID * 4 + (BWD ? (2 * KS + 8 + 8 + 4 * NKT) : (2 * KS + 8)) * 1024;
    static constexpr int N16 = bytes / 16, IT = (N16 + 255) / 256;
    uint4 v[IT];
    __device__ __forceinline__ void load(const MlpArgs& a) {
        // (without a prepared blob the same loads read the first weight -- valid memory, results unused: one instruction stream)
        const uint4* src = reinterpret_cast<const uint4*>(a.prepared ? a.prepared : (const void*)a.w2);
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int i = a.prepared ? threadIdx.x + 256 * it : 0;
            v[it] = src[i < N16 ? i : N16 - 1];
        }
    }
    __device__ __forceinline__ void store(char* smem, const MlpArgs& a) const {
        if (a.prepared) {
            uint4* dst = reinterpret_cast<uint4*>(smem);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int i = threadIdx.x + 256 * it;
                if (i < N16) dst[i] = v[it];
            }
            return;
        }
        float* lc = reinterpret_cast<float*>(smem);
        bf16* w1img = reinterpret_cast<bf16*>(smem + 4 * HID * 4);
        bf16* w2img = w1img + 2 * KS * 512;
        build_consts(lc, a, threadIdx.x, blockDim.x);
        build_image(w1img, 2, KS, a.w1, a.ldw1, HID, a.Kreal, false, false, threadIdx.x, blockDim.x);
        build_image(w2img, 2, 4, a.w2, HID, a.Oreal, HID, false, true, threadIdx.x, blockDim.x);
        if (BWD) {
            bf16* w2timg = w2img + 8 * 512;
            bf16* w1timg = w2timg + 8 * 512;
            build_image(w2timg, 2, 4, a.w2, HID, HID, a.Oreal, true, true, threadIdx.x, blockDim.x);
            build_image(w1timg, NKT, 4, a.w1, a.ldw1, a.Kreal, HID, true, true, threadIdx.x, blockDim.x);
        }
    }
};
template <int KS, bool BWD>
__device__ __forceinline__ void stage_parameters(char* smem, const MlpArgs& a) {
    StagedParams<KS, BWD> sp;
    sp.load(a);
    sp.store(smem, a);
}

// the same images written to global memory once per parameter version (every launch of the fused kernels then copies them)
template <int KS>
__global__ void __launch_bounds__(256) row_mlp_prepare_kernel(MlpArgs a, char* blob) {
    constexpr int NKT = (16 * KS + 31) / 32;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
    float* lc = reinterpret_cast<float*>(blob);
    bf16* w1img = reinterpret_cast<bf16*>(blob + 4 * HID * 4);
    bf16* w2img = w1img + 2 * KS * 512;
    bf16* w2timg = w2img + 8 * 512;
    bf16* w1timg = w2timg + 8 * 512;
    build_consts(lc, a, tid, n);
    build_image(w1img, 2, KS, a.w1, a.ldw1, HID, a.Kreal, false, false, tid, n);
    build_image(w2img, 2, 4, a.w2, HID, a.Oreal, HID, false, true, tid, n);
    build_image(w2timg, 2, 4, a.w2, HID, HID, a.Oreal, true, true, tid, n);
    build_image(w1timg, NKT, 4, a.w1, a.ldw1, a.Kreal, HID, true, true, tid, n);
}

// LayerNorm statistics of the lane's row (the 64 features live on this lane and its partner in the other half)
__device__ __forceinline__ void row_stats(const float* z, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += z[i];
    s += __shfl_xor(s, 32, 64);
    mean = s * (1.f / 64.f);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) q += (z[i] - mean) * (z[i] - mean);
    q += __shfl_xor(q, 32, 64);
    rstd = rsqrtf(q * (1.f / 64.f) + eps);
}

// ---------------------------------------------------------------------------------------------- forward
// Rows enter and leave through a per-wave LDS staging tile: HBM sees 16 bytes per lane with 8 lanes per 128-byte row (whole cache
// lines per instruction), the MFMA / accumulator layout (a lane = one row, 8-byte pieces of it) is taken from LDS.  Reading the
// pieces straight from HBM touches every line of a row in 8 separate instructions and runs at a quarter of this rate (L1 bound).
constexpr int SROW = HID * 2 + 16;   // staged row stride
template <int KS> constexpr int stage_bytes() { return 32 * ((16 * KS * 2 + 16) > SROW ? (16 * KS * 2 + 16) : SROW); }
template <int KS> constexpr int fwd_lds_bytes() { return 4 * HID * 4 + (2 * KS + 8) * 1024 + 4 * stage_bytes<KS>(); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int KS>
__global__ void __launch_bounds__(256, 2) row_mlp_fwd_kernel(MlpArgs a) {
    constexpr int K = 16 * KS, XV = K / 8, XROW = K * 2 + 16, X_IT = (32 * XV + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lc = reinterpret_cast<float*>(smem);
    bf16* w1img = reinterpret_cast<bf16*>(smem + 4 * HID * 4);
    bf16* w2img = w1img + 2 * KS * 512;
    char* stg = smem + 4 * HID * 4 + (2 * KS + 8) * 1024 + (threadIdx.x >> 6) * stage_bytes<KS>();
    stage_parameters<KS, false>(smem, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
    const int rsub = lane >> 3, chunk = lane & 7;            // row layout: rows 8 it + rsub, 16-byte chunk
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (a.R + 31) / 32;
    const bool has_ga = a.ga != nullptr, has_gb = a.gb != nullptr, has_res = a.out_res != nullptr, has_ln = a.gamma != nullptr;

    for (int64_t tIdx = wave; tIdx < ntiles; tIdx += nwaves) {
        asm volatile("" ::: "memory");   // keep the loop-invariant LDS reads (weights, constants) inside the loop: hoisted, they spill
        const int64_t row0 = tIdx * 32;
        const int64_t last = a.R - 1;
        // ---- all HBM loads of the tile, row layout (addresses of rows past the end are clamped: their results are never stored)
        u32x4 vx[X_IT], va[4], vb[4], vr[4];
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int v = lane + 64 * it, rr = v / XV, c = v - rr * XV;
            int64_t gr = row0 + (rr < 32 ? rr : 31);
            gr = gr < last ? gr : last;
            vx[it] = reinterpret_cast<const u32x4*>(a.x)[gr * XV + c];
        }
        int64_t rl = row0 + r;
        rl = rl < last ? rl : last;
        if (has_ga) {
            const int ia = a.ia ? a.ia[rl] : (int)rl;
#pragma unroll
            for (int it = 0; it < 4; ++it) va[it] = reinterpret_cast<const u32x4*>(a.ga)[(int64_t)__shfl(ia, 8 * it + rsub, 64) * 8 + chunk];
        }
        if (has_gb) {
            const int ib = a.ib ? a.ib[rl] : (int)rl;
#pragma unroll
            for (int it = 0; it < 4; ++it) vb[it] = reinterpret_cast<const u32x4*>(a.gb)[(int64_t)__shfl(ib, 8 * it + rsub, 64) * 8 + chunk];
        }
        if (has_res) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                int64_t gr = row0 + 8 * it + rsub;
                gr = gr < last ? gr : last;
                vr[it] = reinterpret_cast<const u32x4*>(a.res)[gr * 8 + chunk];
            }
        }
        // ---- x through the staging tile -> B operands -> pre = W1 x
        bf16x8 xop[KS];
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int v = lane + 64 * it, rr = v / XV, c = v - rr * XV;
            if (rr < 32) *reinterpret_cast<u32x4*>(stg + rr * XROW + c * 16) = vx[it];
        }
        lds_order();
#pragma unroll
        for (int s = 0; s < KS; ++s) xop[s] = *reinterpret_cast<const bf16x8*>(stg + r * XROW + (16 * s + 8 * h) * 2);
        lds_order();
        f32x16 acc[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            acc[m] = zero16();
#pragma unroll
            for (int s = 0; s < KS; ++s) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wop(w1img, KS, m, s, lane), xop[s], acc[m], 0, 0, 0);
        }
        float pre[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(lc + 32 * (q >> 2) + 8 * (q & 3) + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) pre[4 * q + e] = acc[q >> 2][4 * (q & 3) + e] + v[e];
        }
        // ---- gathered addends through the staging tile (one source after the other)
        auto add_staged = [&](const u32x4 (&v)[4], float* dstv) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < 4; ++it) *reinterpret_cast<u32x4*>(stg + (8 * it + rsub) * SROW + chunk * 16) = v[it];
            lds_order();
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bf16x4 pc = *reinterpret_cast<const bf16x4*>(stg + r * SROW + (32 * (q >> 2) + 8 * (q & 3) + 4 * h) * 2);
#pragma unroll
                for (int e = 0; e < 4; ++e) dstv[4 * q + e] += (float)pc[e];
            }
            lds_order();
        };
        if (has_ga) add_staged(va, pre);
        if (has_gb) add_staged(vb, pre);
        // ---- h = silu(pre) -> z = W2 h + b2 -> LayerNorm
        float hv[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) hv[i] = pre[i] * silu_sig(pre[i]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            acc[m] = zero16();
#pragma unroll
            for (int sp = 0; sp < 4; ++sp)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wop(w2img, 4, m, sp, lane), acc_op(hv + 16 * (sp >> 1), sp & 1), acc[m], 0, 0, 0);
        }
        float y[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(lc + 64 + 32 * (q >> 2) + 8 * (q & 3) + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[4 * q + e] = acc[q >> 2][4 * (q & 3) + e] + v[e];
        }
        if (has_ln) {
            float mean, rstd;
            row_stats(y, a.eps, mean, rstd);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c0 = 32 * (q >> 2) + 8 * (q & 3) + 4 * h;
                const f32x4 gm = *reinterpret_cast<const f32x4*>(lc + 128 + c0), bt = *reinterpret_cast<const f32x4*>(lc + 192 + c0);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[4 * q + e] = (y[4 * q + e] - mean) * rstd * gm[e] + bt[e];
            }
        }
        // ---- outputs through the staging tile: accumulator-layout pieces in, whole rows out
        auto store_staged = [&](bf16* dst) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                store4(reinterpret_cast<bf16*>(stg + r * SROW + (32 * (q >> 2) + 8 * (q & 3) + 4 * h) * 2), y[4 * q], y[4 * q + 1], y[4 * q + 2],
                       y[4 * q + 3]);
            lds_order();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = 8 * it + rsub;
                const u32x4 v = *reinterpret_cast<const u32x4*>(stg + rr * SROW + chunk * 16);
                if (row0 + rr < a.R) reinterpret_cast<u32x4*>(dst)[(row0 + rr) * 8 + chunk] = v;
            }
            lds_order();
        };
        if (a.out) store_staged(a.out);
        if (has_res) {
            float zero[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) zero[i] = 0.f;
            (void)zero;
            add_staged(vr, y);
            store_staged(a.out_res);
        }
    }
}

// ---------------------------------------------------------------------------------------------- backward
#ifndef P4C_MLP_EXP
#define P4C_MLP_EXP 0   // diagnostic builds: bit 0 no weight-gradient phase, 1 no column sums, 2 no HBM stores, 4 no HBM loads after the first tile
#endif
template <int KS> constexpr int nkt() { return (16 * KS + 31) / 32; }
template <int KS> constexpr int xrow_bytes() { return 16 * KS * 2 + 16; }
// per-wave LDS images of one 32-row tile: x | h | dz | dpre | d*xhat | d
template <int KS> constexpr int wave_img_bytes() { return 32 * (xrow_bytes<KS>() + 5 * PROW); }
template <int KS> constexpr int bwd_weights_bytes() { return (2 * KS + 8 + 8 + 4 * nkt<KS>()) * 1024; }
template <int KS> constexpr int bwd_lds_bytes() { return 4 * HID * 4 + bwd_weights_bytes<KS>() + 4 * wave_img_bytes<KS>(); }
// floats of one workgroup's partial: dW1 [64][K] | dW2 [64][64] | db1 | db2 | dgamma | dbeta
template <int KS> constexpr int partial_floats() { return HID * 16 * KS + HID * HID + 4 * HID; }
constexpr int PER_WAVE_MAX_G = 16;   // workgroups up to which every wave leaves its own partial slot (see the end of row_mlp_bwd_kernel)

// A lane of the accumulator layout owns features 8q + 4h .. +3 (q = 0..7) of its row: eight 8-byte pieces, interleaved with those
// of its partner lane (same row, other half h).  Loaded as such, a wave instruction touches 32 rows x 16 bytes; instead each lane
// loads the four 16-byte chunks 16i + 8h .. +7 (32 contiguous bytes per row and instruction, half the memory instructions) and
// the partners trade the halves that belong to the other one: v_permlane32_swap exchanges lanes 32..63 of its first operand with
// lanes 0..31 of its second, which is exactly that trade.
typedef unsigned int mlp_u32x2 __attribute__((ext_vector_type(2)));
struct RowChunks { u32x4 c[4]; };
__device__ __forceinline__ void load_row_chunks(const bf16* __restrict__ rowp, int h, bool live, RowChunks& out) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(rowp + 16 * i + 8 * h);
        out.c[i] = live ? v : u32x4{0u, 0u, 0u, 0u};
    }
}
// the trade, done where the values are consumed (next to the load it would wait for the memory and undo the prefetch)
__device__ __forceinline__ void trade_pieces(const RowChunks& in, bf16x4 (&out)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const mlp_u32x2 s0 = __builtin_amdgcn_permlane32_swap(in.c[i][0], in.c[i][2], false, false);
        const mlp_u32x2 s1 = __builtin_amdgcn_permlane32_swap(in.c[i][1], in.c[i][3], false, false);
        out[2 * i] = __builtin_bit_cast(bf16x4, mlp_u32x2{s0.x, s1.x});       // features 16 i + 4 h ..
        out[2 * i + 1] = __builtin_bit_cast(bf16x4, mlp_u32x2{s0.y, s1.y});   // features 16 i + 8 + 4 h ..
    }
}

// what a lane reads from HBM for one tile: 16-byte pieces of its row of x, 8-byte pieces of the upstream gradients
template <int KS>
struct TileIn {
    bf16x8 x[KS];
    RowChunks dy, dyr;
};
struct TileGather {
    RowChunks a, b;
};

// One wave = one 32-row tile per iteration; the NEXT tile's rows, gradients and gathered rows are in flight while the current one
// is computed (one wave per SIMD here: 64 x (K + 64) gradient accumulators + the working set need the whole register file, so
// memory latency is hidden by this software prefetch, not by occupancy).
template <int KS, bool GATHER>
__global__ void __launch_bounds__(256, 1) row_mlp_bwd_kernel(MlpArgs a) {
    constexpr int K = 16 * KS, NKT = nkt<KS>(), XROW = xrow_bytes<KS>();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lc = reinterpret_cast<float*>(smem);
    bf16* w1img = reinterpret_cast<bf16*>(smem + 4 * HID * 4);
    bf16* w2img = w1img + 2 * KS * 512;
    bf16* w2timg = w2img + 8 * 512;        // A[m = hidden][k = output], permuted k
    bf16* w1timg = w2timg + 8 * 512;       // A[m = input feature][k = hidden], permuted k
    char* wbase = smem + 4 * HID * 4 + bwd_weights_bytes<KS>() + (threadIdx.x >> 6) * wave_img_bytes<KS>();
    char* imgX = wbase;
    char* imgH = imgX + 32 * XROW;
    char* imgDZ = imgH + 32 * PROW;
    char* imgDP = imgDZ + 32 * PROW;
    char* imgDG = imgDP + 32 * PROW;
    char* imgDD = imgDG + 32 * PROW;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (a.R + 31) / 32;
    const bool has_dy = a.dy != nullptr, has_dyr = a.dy_res != nullptr, has_ln = a.gamma != nullptr;

    f32x16 dw1[2][NKT], dw2[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int n = 0; n < NKT; ++n) dw1[m][n] = zero16();
        dw2[m][0] = zero16();
        dw2[m][1] = zero16();
    }
    float db1[8], db2[8], dgam[8], dbet[8];   // column sums of the dpre / dz / d*xhat / d images: features 8 (lane & 7) + j
#pragma unroll
    for (int i = 0; i < 8; ++i) db1[i] = db2[i] = dgam[i] = dbet[i] = 0.f;

    auto load_in = [&](TileIn<KS>& in, int64_t tile) __attribute__((always_inline)) {
        const int64_t row = tile * 32 + r;
        const bool live = tile < ntiles && row < a.R;
        const int64_t rc = live ? row : 0;          // clamped address, result discarded: every path issues the same loads
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.x + rc * K + 16 * s + 8 * h);
            in.x[s] = live ? v : zero8();
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) in.dy.c[i] = in.dyr.c[i] = u32x4{0u, 0u, 0u, 0u};
        if (has_dy) load_row_chunks(a.dy + rc * HID, h, live, in.dy);
        if (has_dyr) load_row_chunks(a.dy_res + rc * HID, h, live, in.dyr);
    };
    auto load_idx = [&](int64_t tile, int& ja, int& jb) __attribute__((always_inline)) {
        const int64_t row = tile * 32 + r;
        const bool live = tile < ntiles && row < a.R;
        ja = jb = 0;
        if (GATHER) {
            if (a.ga) ja = a.ia ? a.ia[live ? row : 0] : (int)(live ? row : 0);
            if (a.gb) jb = a.ib ? a.ib[live ? row : 0] : (int)(live ? row : 0);
        }
    };
    auto load_gather = [&](TileGather& g, int ja, int jb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g.a.c[i] = g.b.c[i] = u32x4{0u, 0u, 0u, 0u};
        if (a.ga) load_row_chunks(a.ga + (int64_t)ja * HID, h, true, g.a);
        if (a.gb) load_row_chunks(a.gb + (int64_t)jb * HID, h, true, g.b);
    };

    // Prologue: the first tile's index and row loads go out BEFORE the parameter images are staged, the gathers (which need the
    // indices) between the staging's loads and its LDS stores: index -> gather and parameters -> LDS overlap instead of following
    // each other (most launches of a hierarchical GNN are one tile per wave: the prologue IS the kernel there).
    TileIn<KS> cur, nxt;
    TileGather gcur, gnxt;
    int ja_n = 0, jb_n = 0;
    {
        int ja, jb;
        load_idx(wave, ja, jb);
        load_in(cur, wave);
        StagedParams<KS, true> sp;
        sp.load(a);
        if (GATHER) load_gather(gcur, ja, jb);
        load_idx(wave + nwaves, ja_n, jb_n);
        sp.store(smem, a);
    }
    __syncthreads();

    if (P4C_MLP_EXP & 16) { nxt = cur; gnxt = gcur; }
    for (int64_t tIdx = wave; tIdx < ntiles; tIdx += nwaves) {
        asm volatile("" ::: "memory");   // keep the loop-invariant LDS reads (weights, constants) inside the loop: hoisted, they spill
        const int64_t row = tIdx * 32 + r;
        const bool live = row < a.R;
        // ---- next tile's loads first: they stay in flight during this tile's compute
        if (!(P4C_MLP_EXP & 16)) {
            if (GATHER) load_gather(gnxt, ja_n, jb_n);
            load_in(nxt, tIdx + nwaves);
            load_idx(tIdx + 2 * nwaves, ja_n, jb_n);
        }

        // ---- forward recompute: pre = W1 x + b1 (+ gathers), h = silu(pre), z = W2 h + b2
        float pre[32], dz[32];
        {
            f32x16 acc[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m] = zero16();
#pragma unroll
                for (int s = 0; s < KS; ++s) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wop(w1img, KS, m, s, lane), cur.x[s], acc[m], 0, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) *reinterpret_cast<bf16x8*>(imgX + r * XROW + (16 * s + 8 * h) * 2) = cur.x[s];
            bf16x4 pa[8], pb[8];
            if (GATHER) {
                trade_pieces(gcur.a, pa);
                trade_pieces(gcur.b, pb);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int m = q >> 2, g = q & 3, c0 = 32 * m + 8 * g + 4 * h;
                f32x4 v = *reinterpret_cast<const f32x4*>(lc + c0);
                if (GATHER) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)pa[q][e] + (float)pb[q][e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) pre[m * 16 + 4 * g + e] = acc[m][4 * g + e] + v[e];
            }
        }
        float zz[32];
        {
            float hv[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                // silu'(x) = s + h (1 - s) with s = sigmoid(x), h = x s: taken here, where s exists, so that the exp / rcp pair
                // (quarter-rate instructions, the largest single vector cost of a tile) runs once per element, not twice
                const float sg = silu_sig(pre[i]);
                hv[i] = pre[i] * sg;
                pre[i] = __builtin_fmaf(hv[i], 1.f - sg, sg);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                store4(reinterpret_cast<bf16*>(imgH + r * PROW + (32 * (q >> 2) + 8 * (q & 3) + 4 * h) * 2), hv[4 * q], hv[4 * q + 1],
                       hv[4 * q + 2], hv[4 * q + 3]);
            f32x16 acc[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m] = zero16();
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wop(w2img, 4, m, sp, lane), acc_op(hv + 16 * (sp >> 1), sp & 1), acc[m], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(lc + 64 + 32 * (q >> 2) + 8 * (q & 3) + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) zz[4 * q + e] = acc[q >> 2][4 * (q & 3) + e] + v[e];
            }
        }
        // ---- upstream gradient d (of y), LayerNorm backward -> dz; d*xhat and d go to images (their column sums are dgamma, dbeta)
        {
            float mean = 0.f, rstd = 1.f;
            if (has_ln) row_stats(zz, a.eps, mean, rstd);
            float m1 = 0.f, m2 = 0.f;
            bf16x4 pdy[8], pdyr[8];
            trade_pieces(cur.dy, pdy);
            trade_pieces(cur.dyr, pdyr);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int c0 = 32 * (q >> 2) + 8 * (q & 3) + 4 * h;
                float d[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = (float)pdy[q][e] + (float)pdyr[q][e];
                if (has_ln) {
                    const f32x4 gm = *reinterpret_cast<const f32x4*>(lc + 128 + c0);
                    float dg[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 4 * q + e;
                        const float xh = (zz[i] - mean) * rstd;
                        dg[e] = d[e] * xh;
                        const float gq = d[e] * gm[e];
                        m1 += gq;
                        m2 += gq * xh;
                        dz[i] = gq;
                        zz[i] = xh;
                    }
                    store4(reinterpret_cast<bf16*>(imgDG + r * PROW + c0 * 2), dg[0], dg[1], dg[2], dg[3]);
                    store4(reinterpret_cast<bf16*>(imgDD + r * PROW + c0 * 2), d[0], d[1], d[2], d[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dz[4 * q + e] = d[e];
                }
            }
            if (has_ln) {
                m1 += __shfl_xor(m1, 32, 64);
                m2 += __shfl_xor(m2, 32, 64);
                m1 *= (1.f / 64.f);
                m2 *= (1.f / 64.f);
#pragma unroll
                for (int i = 0; i < 32; ++i) dz[i] = rstd * (dz[i] - m1 - zz[i] * m2);
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            store4(reinterpret_cast<bf16*>(imgDZ + r * PROW + (32 * (q >> 2) + 8 * (q & 3) + 4 * h) * 2), dz[4 * q], dz[4 * q + 1], dz[4 * q + 2],
                   dz[4 * q + 3]);
        // ---- dh = W2^T dz, dpre = dh * silu'(pre)
        {
            f32x16 acc[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc[m] = zero16();
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wop(w2timg, 4, m, sp, lane), acc_op(dz + 16 * (sp >> 1), sp & 1), acc[m], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) pre[i] *= acc[i >> 4][i & 15];   // dpre = dh * silu'(pre)
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int c0 = 32 * (q >> 2) + 8 * (q & 3) + 4 * h, i = 4 * q;
            store4(reinterpret_cast<bf16*>(imgDP + r * PROW + c0 * 2), pre[i], pre[i + 1], pre[i + 2], pre[i + 3]);
        }
        if (a.dpre) {   // whole rows from the image just written (16 bytes per lane: full cache lines per store instruction)
            lds_order();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = 8 * it + (lane >> 3);
                const u32x4 v = *reinterpret_cast<const u32x4*>(imgDP + rr * PROW + (lane & 7) * 16);
                if (tIdx * 32 + rr < a.R && !(P4C_MLP_EXP & 4)) reinterpret_cast<u32x4*>(a.dpre)[(tIdx * 32 + rr) * 8 + (lane & 7)] = v;
            }
        }
        // ---- dx = W1^T dpre
        if (a.dx) {
            f32x16 acc[NKT];
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                acc[kt] = zero16();
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
                    acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wop(w1timg, 4, kt, sp, lane), acc_op(pre + 16 * (sp >> 1), sp & 1), acc[kt], 0, 0, 0);
            }
            if (KS == 4 && a.dx_add_dyr) {   // (accumulator element 4 g + e of tile kt <-> feature 32 kt + 8 g + 4 h + e = piece 4 kt + g)
                bf16x4 pr[8];
                trade_pieces(cur.dyr, pr);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[kt][j] += (float)pr[(4 * kt + (j >> 2)) & 7][j & 3];
            }
            // the reverse trade of load_row_chunks: 16-byte stores of the chunk 16 i + 8 h .. +7
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                const int kt = i >> 1, g0 = 2 * (i & 1);
                bf16x4 pa, pb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pa[e] = (__bf16)acc[kt][4 * g0 + e];
                    pb[e] = (__bf16)acc[kt][4 * g0 + 4 + e];
                }
                const mlp_u32x2 ua = __builtin_bit_cast(mlp_u32x2, pa), ub = __builtin_bit_cast(mlp_u32x2, pb);
                const mlp_u32x2 s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
                const mlp_u32x2 s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
                if (live && !(P4C_MLP_EXP & 4)) *reinterpret_cast<u32x4*>(a.dx + row * K + 16 * i + 8 * h) = u32x4{s0.x, s1.x, s0.y, s1.y};
            }
        }
        lds_order();
        // ---- weight gradients: reduction over the tile's 32 rows (2 k-steps), operands by transposed reads of the images
#pragma unroll
        for (int ks = 0; ks < ((P4C_MLP_EXP & 1) ? 0 : 2); ++ks) {
            bf16x8 adz[2], adp[2], bh[2], bx[NKT];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                adz[m] = read_tr_nat(imgDZ, PROW, ks, m, lane);
                adp[m] = read_tr_nat(imgDP, PROW, ks, m, lane);
                bh[m] = read_tr_nat(imgH, PROW, ks, m, lane);
            }
#pragma unroll
            for (int n = 0; n < NKT; ++n) {
                // K = 16 (mod 32): the upper half-tile re-reads the lower one (valid memory), its columns are discarded at the end
                const bool upper_missing = (32 * n + 16 >= K) && (((lane >> 4) & 1) == 1);
                bx[n] = read_tr_nat(imgX - (upper_missing ? 32 : 0), XROW, ks, n, lane);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int n = 0; n < 2; ++n) dw2[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adz[m], bh[n], dw2[m][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NKT; ++n) dw1[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(adp[m], bx[n], dw1[m][n], 0, 0, 0);
            }
        }
        // ---- bias / LayerNorm-parameter gradients: column sums of the images (lane = 16-byte chunk lane & 7 of rows (lane >> 3) + 8 it)
#pragma unroll
        for (int it = 0; it < ((P4C_MLP_EXP & 2) ? 0 : 4); ++it) {
            const int off = ((lane >> 3) + 8 * it) * PROW + (lane & 7) * 16;
            const bf16x8 vz = *reinterpret_cast<const bf16x8*>(imgDZ + off);
            const bf16x8 vp = *reinterpret_cast<const bf16x8*>(imgDP + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                db2[j] += (float)vz[j];
                db1[j] += (float)vp[j];
            }
            if (has_ln) {
                const bf16x8 vg = *reinterpret_cast<const bf16x8*>(imgDG + off);
                const bf16x8 vd = *reinterpret_cast<const bf16x8*>(imgDD + off);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    dgam[j] += (float)vg[j];
                    dbet[j] += (float)vd[j];
                }
            }
        }
        lds_order();
        cur = nxt;     // (a two-tile unrolled loop with swapped buffers avoids these copies but spills: 169 registers)
        if (GATHER) gcur = gnxt;
    }

    // ---- parameter-gradient partials.  Small grids (<= PER_WAVE_MAX_G workgroups: the upper mesh levels, where the launch is its
    // own latency): every wave that had a tile stores its sums straight to its OWN global slot (waves 0 .. min(tiles, 4 G) - 1:
    // contiguous) -- no barrier, no pass through LDS.  Larger grids: the four waves of a workgroup add in wave order through LDS
    // (fixed order) and leave one partial per workgroup (4 x less partial traffic where it counts).
    constexpr int PF = partial_floats<KS>();
    static_assert(4 * PF * 4 <= bwd_lds_bytes<KS>(), "a reduction slot per wave must fit the kernel's LDS");
    constexpr int OFF_W2 = HID * K, OFF_B1 = OFF_W2 + HID * HID, OFF_B2 = OFF_B1 + HID, OFF_G = OFF_B2 + HID, OFF_BT = OFF_G + HID;
    const bool per_wave = (int)gridDim.x <= PER_WAVE_MAX_G;
    const bool contributed = wave < ntiles;   // wave 0 of a workgroup always has a tile
    auto write_sums = [&](float* red) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int o = 32 * m + rowmap(i) + 4 * h;
#pragma unroll
                for (int n = 0; n < NKT; ++n) {
                    const int k = 32 * n + r;
                    if (k < K) red[o * K + k] = dw1[m][n][i];
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) red[OFF_W2 + o * HID + 32 * n + r] = dw2[m][n][i];
            }
        if (lane < 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = lane * 8 + j;
                red[OFF_B1 + c] = db1[j];
                red[OFF_B2 + c] = db2[j];
                red[OFF_G + c] = dgam[j];
                red[OFF_BT + c] = dbet[j];
            }
    };
    if (contributed) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            db1[j] = lane8_sum(db1[j]);
            db2[j] = lane8_sum(db2[j]);
            dgam[j] = lane8_sum(dgam[j]);
            dbet[j] = lane8_sum(dbet[j]);
        }
    }
    if (per_wave) {
        if (contributed) write_sums(a.partial + wave * (int64_t)PF);
        return;
    }
    __syncthreads();           // weights and images are dead from here: the whole LDS allocation is free
    if (contributed) write_sums(reinterpret_cast<float*>(smem) + wv * PF);
    __syncthreads();
    const int64_t left = ntiles - (int64_t)blockIdx.x * 4;
    const int nact = left >= 4 ? 4 : (int)left;
    const f32x4* slot = reinterpret_cast<const f32x4*>(smem);
    f32x4* dst = reinterpret_cast<f32x4*>(a.partial + (int64_t)blockIdx.x * PF);
    static_assert(PF % 4 == 0, "the partial is summed in 16-byte pieces");
    constexpr int PF4 = PF / 4, RIT = (PF4 + 255) / 256;
#pragma unroll
    for (int it = 0; it < RIT; ++it) {
        const int i = threadIdx.x + 256 * it;
        if (i < PF4) {
            f32x4 t = slot[i];
            for (int w = 1; w < nact; ++w) t += slot[w * PF4 + i];
            dst[i] = t;
        }
    }
}

// out[j] = sum_s partial[s][j] in a fixed order (same scheme as rows.hip)
__global__ void __launch_bounds__(256) mlp_param_reduce_kernel(const float* __restrict__ partial, int slots, int n, float* __restrict__ out) {
    __shared__ float red[8][33];
    const int jj = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + jj;
    float s0 = 0.f, s1 = 0.f;
    if (j < n) {
        int s = sg;
        for (; s + 8 < slots; s += 16) {
            s0 += partial[(int64_t)s * n + j];
            s1 += partial[(int64_t)(s + 8) * n + j];
        }
        for (; s < slots; s += 8) s0 += partial[(int64_t)s * n + j];
    }
    red[sg][jj] = s0 + s1;
    __syncthreads();
    if (sg == 0 && j < n) {
        float t = red[0][jj];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][jj];
        out[j] = t;
    }
}

// the same sums ADDED into the parameters' gradient buffers: a job for the reduction queue of nodeproj.hip (grad_reduce_submit: at once,
// or batched with the other jobs of the backward while p4c_grad_reduce_defer is on)
struct GradSinks {
    float* dw1; int ld_dw1, k_real;
    float* dw2; int o_real;
    float* db1; float* db2; float* dgamma; float* dbeta;
};

int mlp_grid(int64_t R, int per_cu) {
    const int64_t tiles = (R + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// partial slots a backward launch over R rows leaves (what its reduction job sums)
int mlp_bwd_slots(int64_t R) {
    const int G = mlp_grid(R, 1);
    if (G > PER_WAVE_MAX_G) return G;
    const int64_t tiles = (R + 31) / 32;
    return (int)(tiles < 4 * (int64_t)G ? tiles : 4 * (int64_t)G);
}

int check_args(const char* name, const MlpArgs& a, int K) {
    P4C_CHECK_ARG(a.R > 0, "%s: R must be positive", name);
    P4C_CHECK_ARG(K % 16 == 0 && K >= 16 && K <= 80, "%s: K = %d input features (multiples of 16 up to 80; pad)", name, K);
    P4C_CHECK_ARG(a.x && a.w1 && a.w2, "%s: NULL pointer", name);
    P4C_CHECK_ARG(a.Kreal > 0 && a.Kreal <= K && a.ldw1 >= a.Kreal, "%s: bad first-layer weight shape", name);
    P4C_CHECK_ARG(a.Oreal > 0 && a.Oreal <= HID, "%s: bad output feature count %d", name, a.Oreal);
    P4C_CHECK_ARG((a.ga != nullptr || a.ia == nullptr) && (a.gb != nullptr || a.ib == nullptr), "%s: an index list needs its rows", name);
    P4C_CHECK_ARG(a.R < (int64_t)1 << 31, "%s: too many rows", name);
    P4C_CHECK_ARG((a.gamma == nullptr) == (a.beta == nullptr), "%s: gamma and beta go together", name);
    return P4C_OK;
}

template <int KS>
int launch_fwd(const MlpArgs& a, hipStream_t s) {
    constexpr int smem = fwd_lds_bytes<KS>();
    hipLaunchKernelGGL(row_mlp_fwd_kernel<KS>, dim3(mlp_grid(a.R, 4)), dim3(256), smem, s, a);
    P4C_CHECK_LAUNCH("row_mlp_fwd");
    return P4C_OK;
}
template <int KS>
int launch_bwd(const MlpArgs& a, float* grads, const GradSinks* sinks, hipStream_t s) {
    constexpr int smem = bwd_lds_bytes<KS>();
    static_assert(4 * wave_img_bytes<KS>() >= partial_floats<KS>() * 4, "reduction buffer must fit the waves' images");
    P4C_TRY(ensure_dyn_smem((const void*)row_mlp_bwd_kernel<KS, false>, smem));
    P4C_TRY(ensure_dyn_smem((const void*)row_mlp_bwd_kernel<KS, true>, smem));
    const int G = mlp_grid(a.R, 1);
    if (a.ga || a.gb)
        hipLaunchKernelGGL((row_mlp_bwd_kernel<KS, true>), dim3(G), dim3(256), smem, s, a);
    else
        hipLaunchKernelGGL((row_mlp_bwd_kernel<KS, false>), dim3(G), dim3(256), smem, s, a);
    P4C_CHECK_LAUNCH("row_mlp_bwd");
    const int n = partial_floats<KS>();
    if (sinks) {
        GradReduceJob job{};
        job.partial = a.partial; job.slots = mlp_bwd_slots(a.R); job.n = n; job.kind = GRAD_JOB_MLP; job.K = 16 * KS;
        job.p[0] = sinks->dw1; job.p[1] = sinks->dw2; job.p[2] = sinks->db1; job.p[3] = sinks->db2; job.p[4] = sinks->dgamma; job.p[5] = sinks->dbeta;
        job.ld[0] = sinks->ld_dw1; job.k_real = sinks->k_real; job.o_real = sinks->o_real;
        return grad_reduce_submit(job, s);
    }
    hipLaunchKernelGGL(mlp_param_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, s, a.partial, mlp_bwd_slots(a.R), n, grads);
    P4C_CHECK_LAUNCH("mlp_param_reduce");
    return P4C_OK;
}

}  // namespace
}  // namespace p4c

using namespace p4c;

static MlpArgs to_args(const p4c_row_mlp_desc* d) {
    MlpArgs a{};
    a.x = (const bf16*)d->x; a.w1 = d->w1; a.ldw1 = d->ldw1; a.Kreal = d->k_real; a.b1 = d->b1;
    a.w2 = d->w2; a.b2 = d->b2; a.Oreal = d->o_real; a.gamma = d->gamma; a.beta = d->beta; a.eps = d->eps;
    a.ga = (const bf16*)d->gather_a; a.ia = d->index_a; a.gb = (const bf16*)d->gather_b; a.ib = d->index_b;
    a.res = (const bf16*)d->res; a.out = (bf16*)d->out; a.out_res = (bf16*)d->out_res;
    a.dy = (const bf16*)d->dy; a.dy_res = (const bf16*)d->dy_res; a.dx = (bf16*)d->dx; a.dpre = (bf16*)d->dpre;
    a.partial = nullptr; a.R = d->rows; a.prepared = d->prepared; a.dx_add_dyr = d->dx_plus_dy_res;
    return a;
}

extern "C" int p4c_row_mlp_fwd(const p4c_row_mlp_desc* d, p4c_stream_t stream) {
    P4C_CHECK_ARG(d != nullptr, "p4c_row_mlp_fwd: NULL descriptor");
    MlpArgs a = to_args(d);
    int rc = check_args("p4c_row_mlp_fwd", a, d->k);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_ARG(a.out || a.out_res, "p4c_row_mlp_fwd: no output");
    P4C_CHECK_ARG(!a.out_res || a.res, "p4c_row_mlp_fwd: out_res needs res");
    hipStream_t s = as_stream(stream);
    switch (d->k / 16) {
        case 1: return launch_fwd<1>(a, s);
        case 2: return launch_fwd<2>(a, s);
        case 3: return launch_fwd<3>(a, s);
        case 4: return launch_fwd<4>(a, s);
        default: return launch_fwd<5>(a, s);
    }
}

extern "C" size_t p4c_row_mlp_bwd_workspace_bytes(int64_t rows, int k) {
    if (rows <= 0 || k <= 0) return 0;
    return (size_t)mlp_bwd_slots(rows) * (HID * (size_t)k + HID * HID + 4 * HID) * sizeof(float);
}

static int row_mlp_bwd_common(const char* name, const p4c_row_mlp_desc* d, float* grads, const GradSinks* sinks, void* workspace,
                              p4c_stream_t stream) {
    MlpArgs a = to_args(d);
    int rc = check_args(name, a, d->k);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_ARG(a.dy || a.dy_res, "%s: no upstream gradient", name);
    P4C_CHECK_ARG(!a.dx_add_dyr || (d->k == 64 && a.dx && a.dy_res), "%s: dx_plus_dy_res needs k = 64, dx and dy_res", name);
    a.partial = reinterpret_cast<float*>(workspace);
    hipStream_t s = as_stream(stream);
    switch (d->k / 16) {
        case 1: return launch_bwd<1>(a, grads, sinks, s);
        case 2: return launch_bwd<2>(a, grads, sinks, s);
        case 3: return launch_bwd<3>(a, grads, sinks, s);
        case 4: return launch_bwd<4>(a, grads, sinks, s);
        default: return launch_bwd<5>(a, grads, sinks, s);
    }
}

extern "C" int p4c_row_mlp_bwd(const p4c_row_mlp_desc* d, float* grads, void* workspace, p4c_stream_t stream) {
    P4C_CHECK_ARG(d != nullptr && grads != nullptr && workspace != nullptr, "p4c_row_mlp_bwd: NULL pointer");
    return row_mlp_bwd_common("p4c_row_mlp_bwd", d, grads, nullptr, workspace, stream);
}

extern "C" int p4c_row_mlp_bwd_accumulate(const p4c_row_mlp_desc* d, const p4c_row_mlp_grad_sinks* sinks, void* workspace,
                                          p4c_stream_t stream) {
    P4C_CHECK_ARG(d != nullptr && sinks != nullptr && workspace != nullptr, "p4c_row_mlp_bwd_accumulate: NULL pointer");
    P4C_CHECK_ARG(sinks->dw1 == nullptr || sinks->ld_dw1 >= d->k_real, "p4c_row_mlp_bwd_accumulate: bad dw1 row stride");
    GradSinks g{sinks->dw1, sinks->ld_dw1, d->k_real, sinks->dw2, d->o_real, sinks->db1, sinks->db2, sinks->dgamma, sinks->dbeta};
    return row_mlp_bwd_common("p4c_row_mlp_bwd_accumulate", d, nullptr, &g, workspace, stream);
}

extern "C" size_t p4c_row_mlp_prepared_bytes(int k) {
    if (k <= 0 || k % 16) return 0;
    const int KS = k / 16, NKT = (k + 31) / 32;
    return (size_t)4 * HID * 4 + (size_t)(2 * KS + 8 + 8 + 4 * NKT) * 1024;
}

extern "C" int p4c_row_mlp_prepare(const p4c_row_mlp_desc* d, void* prepared, p4c_stream_t stream) {
    P4C_CHECK_ARG(d != nullptr && prepared != nullptr, "p4c_row_mlp_prepare: NULL pointer");
    MlpArgs a = to_args(d);
    a.prepared = nullptr;
    P4C_CHECK_ARG(d->k % 16 == 0 && d->k >= 16 && d->k <= 80, "p4c_row_mlp_prepare: K = %d input features (multiples of 16 up to 80)", d->k);
    P4C_CHECK_ARG(a.w1 && a.w2 && a.Kreal > 0 && a.Kreal <= d->k && a.ldw1 >= a.Kreal && a.Oreal > 0 && a.Oreal <= HID,
                  "p4c_row_mlp_prepare: bad weight shapes");
    hipStream_t s = as_stream(stream);
    char* blob = reinterpret_cast<char*>(prepared);
    switch (d->k / 16) {
        case 1: hipLaunchKernelGGL(row_mlp_prepare_kernel<1>, dim3(32), dim3(256), 0, s, a, blob); break;
        case 2: hipLaunchKernelGGL(row_mlp_prepare_kernel<2>, dim3(32), dim3(256), 0, s, a, blob); break;
        case 3: hipLaunchKernelGGL(row_mlp_prepare_kernel<3>, dim3(32), dim3(256), 0, s, a, blob); break;
        case 4: hipLaunchKernelGGL(row_mlp_prepare_kernel<4>, dim3(32), dim3(256), 0, s, a, blob); break;
        default: hipLaunchKernelGGL(row_mlp_prepare_kernel<5>, dim3(32), dim3(256), 0, s, a, blob); break;
    }
    P4C_CHECK_LAUNCH("row_mlp_prepare");
    return P4C_OK;
}
