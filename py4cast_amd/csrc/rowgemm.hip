// Linear layers on token rows (SwinUNetR's qkv / proj / MLP projections and patch merging, py4cast/models.py:10-20 -> mfai): y = x W^T + b
// with R = 10^4..10^5 rows and K, N of a few dozen to a few hundred features.  The library runs these shapes as stream-K GEMMs with a
// floor of ~20 us per call and 50-110 us for the weight gradient (a reduction over all rows), against 3-30 MB of traffic per call:
// every one of them is a stream of short rows, HBM-bound at a few microseconds.
//   * row_gemm_kernel       : Y = X M^T (+ bias), M = W (forward) or W^T (data gradient: X = dY).  The fp32 master weight is laid out
//     once per workgroup as the MFMA A-operand image in LDS (zero-padded to 32 x 16 tiles, converted to bf16 on the way: no cast
//     launch); a wave takes 32 rows at a time, its B operands are 16-byte loads of the rows straight from HBM, the accumulator
//     layout gives every lane 4 consecutive output features of its row (8-byte stores).
//   * row_gemm_wgrad_kernel : dW[n][k] = sum_r dY[r][n] X[r][k] and db[n] = sum_r dY[r][n] (as one more column: a ones column
//     appended to the X tile).  The reduction index is the ROW, so both operands are transposed LDS reads (ds_read_b64_tr_b16) of the
//     [row][feature] tiles; blockIdx.y takes 64 output features, persistent waves keep the 64 x K accumulator in registers,
//     per-wave partials are summed in a fixed order by row_gemm_reduce_kernel (no atomics: bit-identical reruns).
#include "common.hpp"

namespace p4c {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __hip_bfloat16 bf16;

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)0.f;
    return r;
}

struct RowGemmArgs {
    const bf16* x;      // (R, K) rows, row stride ldx elements
    int64_t ldx;
    const float* w;     // fp32 master weight; M(n, k) = transposed ? w[k * ldw + n] : w[n * ldw + k]
    int ldw, transposed;
    const float* bias;  // [N] or NULL
    bf16* y;            // (R, N) rows, row stride ldy
    int64_t ldy;
    int64_t R;
    int K, N;
    // fused epilogue (round 5): act 1 = the bf16-rounded pre-activation goes to aux (row stride ldaux), GELU (erf) of it on;
    // act 2 = times GELU'(aux) (data gradient through the GELU); then + res (row stride ldres; may alias y: accumulate)
    const bf16* res;
    int64_t ldres;
    bf16* aux;
    int64_t ldaux;
    int act;
};

__device__ __forceinline__ float rg_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float rg_gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// LDS: image[(tile * S + s) * 64 + lane] = 8 bf16 = M[32 tile + (lane & 31)][16 s + 8 (lane >> 5) + j], zero outside (N, K); then
// the bias as floats [32 * tiles]
template <int S>
__global__ void __launch_bounds__(256, (S <= 8 ? 2 : 1)) row_gemm_kernel(RowGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles = (a.N + 31) / 32;
    __bf16* img = reinterpret_cast<__bf16*>(smem);
    float* lb = reinterpret_cast<float*>(smem + (size_t)tiles * S * 64 * 16);
    {
        // one 16-byte operand piece (8 consecutive k of one n) per thread and trip: two float4 loads of the master weight when its
        // rows allow it (K is a multiple of 8, so a piece is all inside or all outside the matrix), eight strided ones when transposed
        const int pieces = tiles * S * 64;
        const bool vec = !a.transposed && (a.ldw & 3) == 0 && ((reinterpret_cast<uintptr_t>(a.w) & 15) == 0);
        for (int idx = threadIdx.x; idx < pieces; idx += blockDim.x) {
            const int ln = idx & 63, ts = idx >> 6;
            const int s = ts % S, tile = ts / S;
            const int n = 32 * tile + (ln & 31), k0 = 16 * s + 8 * (ln >> 5);
            bf16x8 o = zero8();
            if (n < a.N && k0 < a.K) {
                if (vec) {
                    const float4 lo = *reinterpret_cast<const float4*>(a.w + (int64_t)n * a.ldw + k0);
                    const float4 hi = *reinterpret_cast<const float4*>(a.w + (int64_t)n * a.ldw + k0 + 4);
                    o[0] = (__bf16)lo.x; o[1] = (__bf16)lo.y; o[2] = (__bf16)lo.z; o[3] = (__bf16)lo.w;
                    o[4] = (__bf16)hi.x; o[5] = (__bf16)hi.y; o[6] = (__bf16)hi.z; o[7] = (__bf16)hi.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        o[j] = (__bf16)(a.transposed ? a.w[(int64_t)(k0 + j) * a.ldw + n] : a.w[(int64_t)n * a.ldw + k0 + j]);
                }
            }
            *reinterpret_cast<bf16x8*>(img + (int64_t)idx * 8) = o;
        }
        for (int i = threadIdx.x; i < 32 * tiles; i += blockDim.x) lb[i] = (a.bias && i < a.N) ? a.bias[i] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (a.R + 31) / 32;

    auto load = [&](bf16x8 (&xs)[S], int64_t t) __attribute__((always_inline)) {
        const int64_t row = t * 32 + r;
        const bool live = t < ntiles && row < a.R;
        const bf16* p = a.x + (live ? row : 0) * a.ldx;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int k = 16 * s + 8 * h;
            // (clamped address, discarded result: every lane issues the same loads)
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + (k < a.K ? k : 0));
            xs[s] = (live && k < a.K) ? v : zero8();
        }
    };
    bf16x8 cur[S], nxt[S];
    load(cur, wave);
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        load(nxt, t + nwaves);                       // in flight during this tile's products
        const int64_t row = t * 32 + r;
        const bool live = row < a.R;
        bf16* yrow = a.y + (live ? row : 0) * a.ldy;
        for (int c0 = 0; c0 < tiles; c0 += 4) {      // 128 output features at a time: 64 accumulator registers
            f32x16 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = zero16();
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (c0 + u < tiles)
                        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            *reinterpret_cast<const bf16x8*>(img + (((c0 + u) * S + s) * 64 + lane) * 8), cur[s], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (c0 + u >= tiles) continue;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = 32 * (c0 + u) + 8 * q + 4 * h;      // accumulator element 4 q + e <-> feature n + e of the lane's row
                    if (live && n < a.N) {
                        const float4 bv = *reinterpret_cast<const float4*>(lb + n);
                        float v[4] = {acc[u][4 * q] + bv.x, acc[u][4 * q + 1] + bv.y, acc[u][4 * q + 2] + bv.z, acc[u][4 * q + 3] + bv.w};
                        if (a.act == 1) {
                            bf16x4 hq;
#pragma unroll
                            for (int e = 0; e < 4; ++e) hq[e] = (__bf16)v[e];
                            *reinterpret_cast<bf16x4*>(a.aux + row * a.ldaux + n) = hq;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = rg_gelu((float)hq[e]);      // of the STORED pre-activation
                        } else if (a.act == 2) {
                            const bf16x4 hq = *reinterpret_cast<const bf16x4*>(a.aux + row * a.ldaux + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= rg_gelu_grad((float)hq[e]);
                        }
                        if (a.res) {
                            const bf16x4 rq = *reinterpret_cast<const bf16x4*>(a.res + row * a.ldres + n);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += (float)rq[e];
                        }
                        bf16x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                        *reinterpret_cast<bf16x4*>(yrow + n) = o;
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) cur[s] = nxt[s];
    }
}

// ------------------------------------------------------------------------------------------------ weight (+ bias) gradient
constexpr int WROWS = 64;   // rows per wave tile
constexpr int WAVE_SLOTS_FROM = 4;   // input tiles (32 columns) from which every wave writes its own partial

struct RowWgradArgs {
    const bf16* dy;     // (R, N), row stride ldy
    int64_t ldy;
    const bf16* x;      // (R, K), row stride ldx
    int64_t ldx;
    float* partial;     // [gridDim.x][4 waves (NT >= WAVE_SLOTS_FROM) | 1][gridDim.y][64][32 * NT]
    int64_t R;
    int N, K, ones;     // ones: 1 = append the ones column (index K) whose products are the bias gradient
};

template <int NT>
__global__ void __launch_bounds__(256, 1) row_gemm_wgrad_kernel(RowWgradArgs a) {
    constexpr int KP = 32 * NT;
    constexpr int DROW = 64 * 2 + 16, XROW = KP * 2 + 16;          // padded LDS row strides in bytes
    constexpr int WAVE_LDS = WROWS * (DROW + XROW);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* imgD = smem + wv * WAVE_LDS;
    char* imgX = imgD + WROWS * DROW;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (a.R + WROWS - 1) / WROWS;
    const int n0 = 64 * blockIdx.y;                                // this workgroup's 64 output features
    const int nv = (a.N - n0 < 64 ? a.N - n0 : 64) / 8;            // live 16-byte vectors per dY row (N is a multiple of 8)
    const int xv = a.K / 8;                                        // ... per X row

    // the padding columns of both images stay zero for the whole kernel: clear everything once
    for (int i = lane; i < WAVE_LDS / 16; i += 64) reinterpret_cast<u32x4*>(imgD)[i] = u32x4{0, 0, 0, 0};
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = zero16();

    const int i16 = lane & 15, tg = (lane >> 4) & 1, h = lane >> 5;
    const unsigned short one_bf16 = 0x3F80;

    for (int64_t t = wave; t < ntiles; t += nwaves) {
        const int64_t row0 = t * WROWS;
        for (int v = lane; v < WROWS * nv; v += 64) {
            const int rr = v / nv, c = v - rr * nv;
            const u32x4 d = (row0 + rr < a.R) ? *reinterpret_cast<const u32x4*>(a.dy + (row0 + rr) * a.ldy + n0 + 8 * c) : u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(imgD + rr * DROW + c * 16) = d;
        }
        for (int v = lane; v < WROWS * xv; v += 64) {
            const int rr = v / xv, c = v - rr * xv;
            const u32x4 d = (row0 + rr < a.R) ? *reinterpret_cast<const u32x4*>(a.x + (row0 + rr) * a.ldx + 8 * c) : u32x4{0, 0, 0, 0};
            *reinterpret_cast<u32x4*>(imgX + rr * XROW + c * 16) = d;
        }
        if (a.ones) *reinterpret_cast<unsigned short*>(imgX + lane * XROW + a.K * 2) = (row0 + lane < a.R) ? one_bf16 : (unsigned short)0;
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < WROWS / 16; ++ks) {
            // operand of k-step ks (rows 16 ks .. +15): lane (feature 32 tile + (lane & 31), h) receives rows 8 h + j, j = 0..7
            const int rbase = 16 * ks + 8 * h + (i16 >> 2);
            bf16x8 av[2], bv[NT];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const char* p = imgD + rbase * DROW + (32 * m + tg * 16 + (i16 & 3) * 4) * 2;
                union { s16x4 s[2]; bf16x8 v; } u;
                u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
                u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * DROW));
                av[m] = u.v;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const char* p = imgX + rbase * XROW + (32 * n + tg * 16 + (i16 & 3) * 4) * 2;
                union { s16x4 s[2]; bf16x8 v; } u;
                u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
                u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * XROW));
                bv[n] = u.v;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }

    const int r = lane & 31;
    if constexpr (NT >= WAVE_SLOTS_FROM) {
        // wide inputs: every wave hands its accumulators to its own partial slot (for a fixed accumulator element the 32 lanes of a
        // half-wave write 128 contiguous bytes).  Adding the four waves through LDS first cost four serial turns of 64 x KP scattered
        // LDS updates -- ~20 us of a 69 us launch at KP = 224 -- against 4x the partial traffic this way
        float* dst = a.partial + (((int64_t)blockIdx.x * 4 + wv) * gridDim.y + blockIdx.y) * (64 * KP);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int i = 0; i < 16; ++i) dst[(32 * m + (i & 3) + 8 * (i >> 2) + 4 * h) * KP + 32 * n + r] = acc[m][n][i];
    } else {
        // narrow inputs: the four waves add in wave order through LDS (cheap at <= 96 columns), one partial per workgroup
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);                 // [64][KP]
        for (int turn = 0; turn < 4; ++turn) {
            if (wv == turn) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int o = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * h, k = 32 * n + r;
                            if (turn == 0) red[o * KP + k] = acc[m][n][i];
                            else red[o * KP + k] += acc[m][n][i];
                        }
            }
            __syncthreads();
        }
        float* dst = a.partial + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * (64 * KP);
        for (int i = threadIdx.x; i < 64 * KP; i += blockDim.x) dst[i] = red[i];
    }
}

// out[j] = sum_s partial[s][j], j < n, in a fixed order: a block owns 32 outputs, its 8 thread rows take the slots s = sg (mod 8)
// with four independent partial sums each, then the 8 rows are added in order through LDS.
__global__ void __launch_bounds__(256) row_gemm_reduce_kernel(const float* __restrict__ partial, int slots, int n, float* __restrict__ out) {
    __shared__ float red[8][33];
    const int jj = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + jj;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < n) {
        int s = sg;
        for (; s + 24 < slots; s += 32) {
            s0 += partial[(int64_t)s * n + j];
            s1 += partial[(int64_t)(s + 8) * n + j];
            s2 += partial[(int64_t)(s + 16) * n + j];
            s3 += partial[(int64_t)(s + 24) * n + j];
        }
        for (; s < slots; s += 8) s0 += partial[(int64_t)s * n + j];
    }
    red[sg][jj] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sg == 0 && j < n) {
        float t = red[0][jj];
#pragma unroll
        for (int i = 1; i < 8; ++i) t += red[i][jj];
        out[j] = t;
    }
}

constexpr int MAX_S = 32;             // K <= 512 features
constexpr int FWD_LDS_LIMIT = 144 * 1024;

int fwd_steps(int K) {   // k-steps of 16 the instantiation for K uses (the image is zero-padded to it)
    const int s = (K + 15) / 16;
    return s <= 4 ? s : s <= 6 ? 6 : s <= 8 ? 8 : s <= 12 ? 12 : s <= 16 ? 16 : s <= 24 ? 24 : 32;
}

int fwd_lds_bytes(int K, int N) {
    const int S = fwd_steps(K), tiles = (N + 31) / 32;
    return tiles * S * 64 * 16 + tiles * 32 * 4;
}

int fwd_grid(int64_t R) {
    const int64_t tiles = (R + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * 2;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int wgrad_nt(int K, int with_bias) { return (K + (with_bias ? 1 : 0) + 31) / 32; }

int wgrad_slots_per_wg(int NT) { return NT >= WAVE_SLOTS_FROM ? 4 : 1; }

int wgrad_grid(int64_t R, int chunks, int NT) {
    const int64_t tiles = (R + WROWS - 1) / WROWS;
    // narrow inputs: at least two tiles per wave (a workgroup's fixed costs -- clearing its LDS tiles, the four-turn reduction, its
    // partial -- are those of ~1.5 tiles); wide ones (per-wave partials, no reduction turns): one
    int64_t blocks = NT >= WAVE_SLOTS_FROM ? (tiles + 3) / 4 : (tiles + 7) / 8;
    int64_t cap = num_cus() / chunks;          // one workgroup per CU (the LDS tiles of four waves fill it)
    if (cap < 1) cap = 1;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int S>
int launch_fwd(const RowGemmArgs& a, hipStream_t st) {
    const int smem = fwd_lds_bytes(a.K, a.N);
    P4C_TRY(ensure_dyn_smem((const void*)row_gemm_kernel<S>, smem));
    hipLaunchKernelGGL((row_gemm_kernel<S>), dim3(fwd_grid(a.R)), dim3(256), smem, st, a);
    P4C_CHECK_LAUNCH("row_gemm");
    return P4C_OK;
}

template <int NT>
int launch_wgrad(const RowWgradArgs& a, int G, int chunks, hipStream_t st) {
    constexpr int smem_tiles = 4 * WROWS * ((64 * 2 + 16) + (32 * NT * 2 + 16));
    constexpr int smem_red = 64 * 32 * NT * 4;
    constexpr int smem = smem_tiles > smem_red ? smem_tiles : smem_red;
    P4C_TRY(ensure_dyn_smem((const void*)row_gemm_wgrad_kernel<NT>, smem));
    hipLaunchKernelGGL((row_gemm_wgrad_kernel<NT>), dim3(G, chunks), dim3(256), smem, st, a);
    P4C_CHECK_LAUNCH("row_gemm_wgrad");
    return P4C_OK;
}

}  // namespace
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_row_gemm_supported(int K, int N) {
    return (K > 0 && N > 0 && K % 8 == 0 && N % 4 == 0 && (K + 15) / 16 <= MAX_S && fwd_lds_bytes(K, N) <= FWD_LDS_LIMIT) ? 1 : 0;
}

extern "C" int p4c_row_gemm(const void* x, int64_t ldx, const float* w, int ldw, int transposed, const float* bias, void* y, int64_t ldy,
                            int64_t R, int K, int N, const void* res, int64_t ldres, int act, void* aux, int64_t ldaux, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && w && y, "p4c_row_gemm: NULL pointer");
    P4C_CHECK_ARG(R >= 0 && p4c_row_gemm_supported(K, N), "p4c_row_gemm: unsupported sizes R=%lld K=%d N=%d (K multiple of 8 up to 512, N multiple of 4, "
                  "weight image within %d KiB of LDS)", (long long)R, K, N, FWD_LDS_LIMIT / 1024);
    P4C_CHECK_ARG(ldx >= K && ldy >= N && ldx % 8 == 0 && ldy % 4 == 0, "p4c_row_gemm: row strides must cover the rows (ldx multiple of 8, ldy of 4)");
    P4C_CHECK_ARG(act >= 0 && act <= 2 && (act == 0 || (aux && ldaux >= N && ldaux % 4 == 0)), "p4c_row_gemm: act 1 / 2 need the aux rows (ldaux multiple of 4)");
    P4C_CHECK_ARG(!res || (ldres >= N && ldres % 4 == 0), "p4c_row_gemm: residual row stride must be a multiple of 4 covering the rows");
    if (R == 0) return P4C_OK;
    RowGemmArgs a{(const bf16*)x, ldx, w, ldw, transposed, bias, (bf16*)y, ldy, R, K, N, (const bf16*)res, ldres, (bf16*)aux, ldaux, act};
    hipStream_t st = as_stream(stream);
    switch (fwd_steps(K)) {
        case 1: return launch_fwd<1>(a, st);
        case 2: return launch_fwd<2>(a, st);
        case 3: return launch_fwd<3>(a, st);
        case 4: return launch_fwd<4>(a, st);
        case 6: return launch_fwd<6>(a, st);
        case 8: return launch_fwd<8>(a, st);
        case 12: return launch_fwd<12>(a, st);
        case 16: return launch_fwd<16>(a, st);
        case 24: return launch_fwd<24>(a, st);
        default: return launch_fwd<32>(a, st);
    }
}

extern "C" int p4c_row_gemm_wgrad_supported(int N, int K, int with_bias) {
    return (N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0 && wgrad_nt(K, with_bias) <= 7 && (N + 63) / 64 <= 8) ? 1 : 0;
}

extern "C" size_t p4c_row_gemm_wgrad_workspace_bytes(int64_t R, int N, int K, int with_bias) {
    if (R <= 0 || !p4c_row_gemm_wgrad_supported(N, K, with_bias)) return 0;
    const int chunks = (N + 63) / 64;
    const int NT = wgrad_nt(K, with_bias);
    return (size_t)wgrad_grid(R, chunks, NT) * wgrad_slots_per_wg(NT) * chunks * 64 * 32 * NT * sizeof(float);
}

extern "C" int p4c_row_gemm_wgrad(const void* dy, int64_t ldy, const void* x, int64_t ldx, float* out, void* workspace, int64_t R, int N,
                                  int K, int with_bias, p4c_stream_t stream) {
    P4C_CHECK_ARG(dy && x && out && workspace, "p4c_row_gemm_wgrad: NULL pointer");
    P4C_CHECK_ARG(R > 0 && p4c_row_gemm_wgrad_supported(N, K, with_bias), "p4c_row_gemm_wgrad: unsupported sizes R=%lld N=%d K=%d "
                  "(multiples of 8, K (+1 with bias) <= 224, N <= 512)", (long long)R, N, K);
    P4C_CHECK_ARG(ldy >= N && ldx >= K && ldy % 8 == 0 && ldx % 8 == 0, "p4c_row_gemm_wgrad: row strides must be multiples of 8 covering the rows");
    const int chunks = (N + 63) / 64, NT = wgrad_nt(K, with_bias);
    const int G = wgrad_grid(R, chunks, NT);
    RowWgradArgs a{(const bf16*)dy, ldy, (const bf16*)x, ldx, (float*)workspace, R, N, K, with_bias ? 1 : 0};
    hipStream_t st = as_stream(stream);
    int rc;
    switch (NT) {
        case 1: rc = launch_wgrad<1>(a, G, chunks, st); break;
        case 2: rc = launch_wgrad<2>(a, G, chunks, st); break;
        case 3: rc = launch_wgrad<3>(a, G, chunks, st); break;
        case 4: rc = launch_wgrad<4>(a, G, chunks, st); break;
        case 5: rc = launch_wgrad<5>(a, G, chunks, st); break;
        case 6: rc = launch_wgrad<6>(a, G, chunks, st); break;
        default: rc = launch_wgrad<7>(a, G, chunks, st); break;
    }
    if (rc != P4C_OK) return rc;
    const int n = chunks * 64 * 32 * NT;
    hipLaunchKernelGGL(row_gemm_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, st, (const float*)workspace, wgrad_slots_per_wg(NT) * G, n, out);
    P4C_CHECK_LAUNCH("row_gemm_reduce");
    return P4C_OK;
}
