// K3 losses: WeightedLoss / ScaledLoss (py4cast/losses.py:103-210) and the fused
// "AR update + loss" training-path kernels (lightning.py:599-633 + 811-816 + losses.py:130-169).
//
// HBM-bound: every kernel streams (B,T,N,F) rows once.  Lanes of a wave are spread over the
// features of one (or several) grid points, so per-feature weights live in registers, the
// per-grid-point interior mask is one load per segment, and ScaledLoss's per-feature sums are
// lane-local.  Reductions are two-stage and deterministic (block partials in a caller-provided
// workspace + a tiny finalize kernel); no float atomics.
//
// Compiled with -ffp-contract=off (see rollout.hip).
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace p4c {

__host__ __device__ static inline int pow2_ge64(int v) {
    int p = 1;
    while (p < v && p < 64) p <<= 1;
    return p;
}

constexpr int LOSS_MAX_ITERS = 4;     // F <= 256
constexpr int LOSS_MAX_BLOCKS = 512;  // partial blocks per (b,t)

struct MaskArg {
    const void* ptr;
    int mode;
};

// returns mask value m and rewrites tgt (nan_to_num) for P4C_MASK_FROM_NAN
__device__ __forceinline__ float load_mask(const MaskArg& ma, int64_t idx, float& tgt) {
    switch (ma.mode) {
        case P4C_MASK_FROM_NAN: {
            const bool isn = tgt != tgt;
            if (isn) tgt = 0.0f;
            return isn ? 0.0f : 1.0f;
        }
        case P4C_MASK_F32: return ((const float*)ma.ptr)[idx];
        case P4C_MASK_U8: return ((const unsigned char*)ma.ptr)[idx] ? 1.0f : 0.0f;
        default: return 1.0f;
    }
}

__device__ __forceinline__ float loss_elem(float pred, float tgt, float m, int kind) {
    const float pm = pred * m, tm = tgt * m;  // losses.py:144 / 195
    const float d = pm - tm;
    return kind == P4C_LOSS_MSE ? d * d : fabsf(d);
}

// d loss_elem / d pred
__device__ __forceinline__ float loss_elem_grad(float pred, float tgt, float m, int kind) {
    const float d = pred * m - tgt * m;
    if (kind == P4C_LOSS_MSE) return 2.0f * d * m;
    return (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f)) * m;  // torch L1Loss backward: sign(d)
}

// ------------------------------------------------------------------ mask union count
__global__ void __launch_bounds__(256)
    mask_all_zero_count_kernel(MaskArg ma, int64_t bs, int64_t ts, int B, int T, int64_t N, int F, int FP, int iters,
                               int32_t* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const unsigned long long segmask = (FP == 64) ? ~0ull : (((1ull << FP) - 1ull) << (pp * FP));
    int local = 0;
    for (int64_t base = (int64_t)wave * PP; base < N; base += (int64_t)nwaves * PP) {
        const int64_t n = base + pp;
        const bool live = n < N;
        bool any_set = false;
        for (int b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t)
                for (int it = 0; it < iters; ++it) {
                    const int f = c0 + it * FP;
                    bool set = false;
                    if (live && f < F) {
                        const int64_t idx = (int64_t)b * bs + (int64_t)t * ts + n * F + f;
                        float tg = (ma.mode == P4C_MASK_FROM_NAN) ? ((const float*)ma.ptr)[idx] : 0.0f;
                        set = load_mask(ma, idx, tg) != 0.0f;
                    }
                    const unsigned long long bal = __ballot(set);
                    any_set = any_set || ((bal & segmask) != 0ull);
                }
        if (live && c0 == 0 && !any_set) local += 1;
    }
    // integer atomics: order independent, exact
    const int tot = (int)wave_sum((float)local);  // local <= a few thousand: exact in fp32
    if (lane == 0 && tot) atomicAdd(count, tot);
}

// ------------------------------------------------------------------ weighted loss, reduced
// grid: (nblk, B*T).  partial[bt*nblk + blk]
__global__ void __launch_bounds__(256)
    weighted_loss_partial_kernel(const float* __restrict__ pred, int64_t pred_bs, int64_t pred_ts,
                                 const float* __restrict__ target, int64_t tgt_bs, int64_t tgt_ts, MaskArg ma,
                                 int64_t mask_bs, int64_t mask_ts, const float* __restrict__ weights,
                                 const float* __restrict__ interior, int kind, float* __restrict__ partial, int T,
                                 int64_t N, int F, int FP, int iters) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bt = blockIdx.y, b = bt / T, t = bt - b * T;
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const float* p = pred + (int64_t)b * pred_bs + (int64_t)t * pred_ts;
    const float* g = target + (int64_t)b * tgt_bs + (int64_t)t * tgt_ts;
    const int64_t mbase = (int64_t)b * mask_bs + (int64_t)t * mask_ts;
    float w[LOSS_MAX_ITERS];
#pragma unroll
    for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
        const int f = c0 + it * FP;
        w[it] = (it < iters && f < F) ? weights[f] : 0.0f;
    }
    float acc = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < N; n += stride) {
        const float im = interior[n];
        float s = 0.0f;
#pragma unroll
        for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
            const int f = c0 + it * FP;
            if (it < iters && f < F) {
                const int64_t e = n * F + f;
                float tg = g[e];
                const float m = load_mask(ma, mbase + e, tg);
                s += loss_elem(p[e], tg, m, kind) * w[it];
            }
        }
        acc += s * im;
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)bt * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[bt*out_stride] = sum(partials) / (num_interior - masked); one wave per (b,t), fixed summation order
__global__ void __launch_bounds__(64) weighted_loss_final_kernel(const float* __restrict__ partial, int nblk, float num_interior,
                                                                 const int32_t* __restrict__ masked_count,
                                                                 float* __restrict__ out, int64_t out_stride, int nbt) {
    const int bt = blockIdx.x;
    if (bt >= nbt) return;
    float s = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += 64) s += partial[(int64_t)bt * nblk + i];
    s = wave_sum(s);
    if (threadIdx.x == 0) {
        const float denom = num_interior - (masked_count ? (float)(*masked_count) : 0.0f);
        out[(int64_t)bt * out_stride] = s / denom;
    }
}

// ------------------------------------------------------------------ weighted loss map (no spatial reduce)
__global__ void __launch_bounds__(256)
    weighted_loss_map_kernel(const float* __restrict__ pred, int64_t pred_bs, int64_t pred_ts,
                             const float* __restrict__ target, int64_t tgt_bs, int64_t tgt_ts, MaskArg ma,
                             int64_t mask_bs, int64_t mask_ts, const float* __restrict__ weights, int kind,
                             float* __restrict__ out_map, int T, int64_t N, int F, int FP, int iters) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bt = blockIdx.y, b = bt / T, t = bt - b * T;
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const float* p = pred + (int64_t)b * pred_bs + (int64_t)t * pred_ts;
    const float* g = target + (int64_t)b * tgt_bs + (int64_t)t * tgt_ts;
    const int64_t mbase = (int64_t)b * mask_bs + (int64_t)t * mask_ts;
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    // every lane of a wave runs the same number of iterations (shuffles need all lanes)
    for (int64_t nb = ((int64_t)blockIdx.x * 4 + wv) * PP; nb < N; nb += stride) {
        const int64_t n = nb + pp;
        float s = 0.0f;
        if (n < N) {
            for (int it = 0; it < iters; ++it) {
                const int f = c0 + it * FP;
                if (f < F) {
                    const int64_t e = n * F + f;
                    float tg = g[e];
                    const float m = load_mask(ma, mbase + e, tg);
                    s += loss_elem(p[e], tg, m, kind) * weights[f];
                }
            }
        }
        s = seg_sum(s, FP);
        if (n < N && c0 == 0) out_map[(int64_t)bt * N + n] = s;
    }
}

// ------------------------------------------------------------------ weighted loss backward
__global__ void __launch_bounds__(256)
    weighted_loss_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ pred, int64_t pred_bs,
                             int64_t pred_ts, const float* __restrict__ target, int64_t tgt_bs, int64_t tgt_ts,
                             MaskArg ma, int64_t mask_bs, int64_t mask_ts, const float* __restrict__ weights,
                             const float* __restrict__ interior, float num_interior,
                             const int32_t* __restrict__ masked_count, int kind, float* __restrict__ dpred,
                             int64_t dpred_bs, int64_t dpred_ts, int T, int64_t N, int F, int FP, int iters) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bt = blockIdx.y, b = bt / T, t = bt - b * T;
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const float* p = pred + (int64_t)b * pred_bs + (int64_t)t * pred_ts;
    const float* g = target + (int64_t)b * tgt_bs + (int64_t)t * tgt_ts;
    float* dp = dpred + (int64_t)b * dpred_bs + (int64_t)t * dpred_ts;
    const int64_t mbase = (int64_t)b * mask_bs + (int64_t)t * mask_ts;
    const float denom = num_interior - (masked_count ? (float)(*masked_count) : 0.0f);
    const float scale = gout[bt] / denom;
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < N; n += stride) {
        const float sc = scale * interior[n];
        for (int it = 0; it < iters; ++it) {
            const int f = c0 + it * FP;
            if (f < F) {
                const int64_t e = n * F + f;
                float tg = g[e];
                const float m = load_mask(ma, mbase + e, tg);
                dp[e] = sc * weights[f] * loss_elem_grad(p[e], tg, m, kind);
            }
        }
    }
}

// ------------------------------------------------------------------ scaled loss
// partial[(bt*nblk + blk)*F + f]
__global__ void __launch_bounds__(256)
    scaled_loss_partial_kernel(const float* __restrict__ pred, int64_t pred_bs, int64_t pred_ts,
                               const float* __restrict__ target, int64_t tgt_bs, int64_t tgt_ts, MaskArg ma,
                               int64_t mask_bs, int64_t mask_ts, const float* __restrict__ interior, int kind,
                               float* __restrict__ partial, int T, int64_t N, int F, int FP, int iters) {
    __shared__ float red[4][64 * LOSS_MAX_ITERS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bt = blockIdx.y, b = bt / T, t = bt - b * T;
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const float* p = pred + (int64_t)b * pred_bs + (int64_t)t * pred_ts;
    const float* g = target + (int64_t)b * tgt_bs + (int64_t)t * tgt_ts;
    const int64_t mbase = (int64_t)b * mask_bs + (int64_t)t * mask_ts;
    float acc[LOSS_MAX_ITERS];
#pragma unroll
    for (int it = 0; it < LOSS_MAX_ITERS; ++it) acc[it] = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < N; n += stride) {
        const float im = interior[n];
#pragma unroll
        for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
            const int f = c0 + it * FP;
            if (it < iters && f < F) {
                const int64_t e = n * F + f;
                float tg = g[e];
                const float m = load_mask(ma, mbase + e, tg);
                acc[it] += loss_elem(p[e], tg, m, kind) * im;  // losses.py:200-201
            }
        }
    }
#pragma unroll
    for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
        const float v = cross_seg_sum(acc[it], FP);
        if (pp == 0) red[wv][it * 64 + c0] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < iters * FP; i += blockDim.x) {
        const int it = i / FP, c = i - it * FP;
        const int f = c + it * FP;
        if (f < F) {
            const int k = it * 64 + c;
            partial[((int64_t)bt * gridDim.x + blockIdx.x) * F + f] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
        }
    }
}

__global__ void scaled_loss_final_kernel(const float* __restrict__ partial, int nblk, float num_interior,
                                         const int32_t* __restrict__ masked_count, const float* __restrict__ std,
                                         int kind, float* __restrict__ out, int nbt, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbt * F) return;
    const int bt = i / F, f = i - bt * F;
    float s = 0.0f;
    for (int k = 0; k < nblk; ++k) s += partial[((int64_t)bt * nblk + k) * F + f];
    const float denom = num_interior - (masked_count ? (float)(*masked_count) : 0.0f);
    float v = s / denom;
    if (kind == P4C_LOSS_MSE) v = sqrtf(v);  // losses.py:205-206
    out[i] = v * std[f];                     // losses.py:208-210
}

// ------------------------------------------------------------------ anomaly-correlation sums (validation metric)
// MetricACC.update (py4cast/metrics.py:387-433): per (b,t,f), over the spatial points,
//   num = mean((p-c)*(t-c)*m),  pp = mean(((p-c)*m)^2),  tt = mean(((t-c)*m)^2)      c = climate mean of feature f
// partial[((k*nbt + bt)*nblk + blk)*F + f], k = 0 (num), 1 (pp), 2 (tt); same two-stage fixed-order reduction as the
// losses, one pass over prediction and target.
__global__ void __launch_bounds__(256)
    acc_partial_kernel(const float* __restrict__ pred, int64_t pred_bs, int64_t pred_ts, const float* __restrict__ target,
                       int64_t tgt_bs, int64_t tgt_ts, MaskArg ma, int64_t mask_bs, int64_t mask_ts,
                       const float* __restrict__ clim, float* __restrict__ partial, int T, int64_t N, int F, int FP,
                       int iters) {
    __shared__ float red[3][4][64 * LOSS_MAX_ITERS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bt = blockIdx.y, b = bt / T, t = bt - b * T;
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const float* p = pred + (int64_t)b * pred_bs + (int64_t)t * pred_ts;
    const float* g = target + (int64_t)b * tgt_bs + (int64_t)t * tgt_ts;
    const int64_t mbase = (int64_t)b * mask_bs + (int64_t)t * mask_ts;
    float a0[LOSS_MAX_ITERS], a1[LOSS_MAX_ITERS], a2[LOSS_MAX_ITERS];
#pragma unroll
    for (int it = 0; it < LOSS_MAX_ITERS; ++it) a0[it] = a1[it] = a2[it] = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < N; n += stride) {
#pragma unroll
        for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
            const int f = c0 + it * FP;
            if (it < iters && f < F) {
                const int64_t e = n * F + f;
                float tg = g[e];
                const float m = load_mask(ma, mbase + e, tg);
                const float c = clim[f];
                const float dp = p[e] - c, dt = tg - c;
                a0[it] += dp * dt * m;             // metrics.py:414-418
                const float pm = dp * m, tm = dt * m;
                a1[it] += pm * pm;                 // metrics.py:419-421
                a2[it] += tm * tm;                 // metrics.py:421-423
            }
        }
    }
#pragma unroll
    for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
        const float v0 = cross_seg_sum(a0[it], FP), v1 = cross_seg_sum(a1[it], FP), v2 = cross_seg_sum(a2[it], FP);
        if (pp == 0) { red[0][wv][it * 64 + c0] = v0; red[1][wv][it * 64 + c0] = v1; red[2][wv][it * 64 + c0] = v2; }
    }
    __syncthreads();
    const int nbt = gridDim.y;
    for (int i = threadIdx.x; i < 3 * iters * FP; i += blockDim.x) {
        const int k = i / (iters * FP), j = i - k * iters * FP;
        const int it = j / FP, c = j - it * FP;
        const int f = c + it * FP;
        if (f < F) {
            const int q = it * 64 + c;
            partial[(((int64_t)k * nbt + bt) * gridDim.x + blockIdx.x) * F + f] =
                (red[k][0][q] + red[k][1][q]) + (red[k][2][q] + red[k][3][q]);
        }
    }
}

// 16-byte vectorised form (F % 4 == 0, F <= 64, no explicit mask tensor): a lane owns 4 features of a grid point
__global__ void __launch_bounds__(256)
    acc_partial_v4_kernel(const float* __restrict__ pred, int64_t pred_bs, int64_t pred_ts, const float* __restrict__ target,
                          int64_t tgt_bs, int64_t tgt_ts, int from_nan, const float* __restrict__ clim,
                          float* __restrict__ partial, int T, int64_t N, int F, int FP4) {
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    __shared__ float red[3][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int bt = blockIdx.y, b = bt / T, t = bt - b * T;
    const int PP = 64 / FP4;
    const int pp = lane / FP4, q = lane % FP4;
    const bool act = 4 * q < F;
    const float* p = pred + (int64_t)b * pred_bs + (int64_t)t * pred_ts;
    const float* g = target + (int64_t)b * tgt_bs + (int64_t)t * tgt_ts;
    v4f_ c = {0, 0, 0, 0};
    if (act) c = *reinterpret_cast<const v4f_*>(clim + 4 * q);
    v4f_ a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < N; n += stride) {
        if (!act) continue;
        const v4f_ pv = *reinterpret_cast<const v4f_*>(p + n * F + 4 * q);
        v4f_ tv = *reinterpret_cast<const v4f_*>(g + n * F + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float tg = tv[j], m = 1.0f;
            if (from_nan) { m = (tg != tg) ? 0.0f : 1.0f; tg = (tg != tg) ? 0.0f : tg; }
            const float dp = pv[j] - c[j], dt = tg - c[j];
            a0[j] += dp * dt * m;
            const float pm = dp * m, tm = dt * m;
            a1[j] += pm * pm;
            a2[j] += tm * tm;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v0 = cross_seg_sum(a0[j], FP4), v1 = cross_seg_sum(a1[j], FP4), v2 = cross_seg_sum(a2[j], FP4);
        if (pp == 0 && act) { red[0][wv][4 * q + j] = v0; red[1][wv][4 * q + j] = v1; red[2][wv][4 * q + j] = v2; }
    }
    __syncthreads();
    const int nbt = gridDim.y;
    for (int i = threadIdx.x; i < 3 * F; i += blockDim.x) {
        const int k = i / F, f = i - k * F;
        partial[(((int64_t)k * nbt + bt) * gridDim.x + blockIdx.x) * F + f] =
            (red[k][0][f] + red[k][1][f]) + (red[k][2][f] + red[k][3][f]);
    }
}

__global__ void acc_final_kernel(const float* __restrict__ partial, int nblk, float n_points, float* __restrict__ out,
                                 int nbt, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * nbt * F) return;
    const int k = i / (nbt * F), j = i - k * nbt * F;
    const int bt = j / F, f = j - bt * F;
    float s = 0.0f;
    for (int q = 0; q < nblk; ++q) s += partial[(((int64_t)k * nbt + bt) * nblk + q) * F + f];
    out[i] = s / n_points;   // torch .mean(dim=spatial)
}

// ------------------------------------------------------------------ NaN-aware moments (dataset statistics)
// compute_mean_std_min_max / compute_time_step_stats (py4cast/datasets/compute_dataset_stats.py:11-127) need, per
// (sample, feature) over all grid points and time steps: the sum and the sum of squares of the non-NaN values, their
// count, and min / max with NaN ignored.  value = x[idx] (or x_next[idx] - x[idx] for the time-step differences).
// partial[((k*B + b)*nblk + blk)*F + f], k = 0 sum, 1 sum of squares, 2 count, 3 min, 4 max
__global__ void __launch_bounds__(256)
    nan_moments_partial_kernel(const float* __restrict__ x, const float* __restrict__ x_next, int64_t bs, int64_t R,
                               float* __restrict__ partial, int F, int FP, int iters) {
    __shared__ float red[5][4][64 * LOSS_MAX_ITERS];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const float* p = x + (int64_t)b * bs;
    const float* q = x_next ? x_next + (int64_t)b * bs : nullptr;
    float a0[LOSS_MAX_ITERS], a1[LOSS_MAX_ITERS], a2[LOSS_MAX_ITERS], mn[LOSS_MAX_ITERS], mx[LOSS_MAX_ITERS];
#pragma unroll
    for (int it = 0; it < LOSS_MAX_ITERS; ++it) { a0[it] = a1[it] = a2[it] = 0.0f; mn[it] = INFINITY; mx[it] = -INFINITY; }
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < R; n += stride) {
#pragma unroll
        for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
            const int f = c0 + it * FP;
            if (it < iters && f < F) {
                const int64_t e = n * F + f;
                const float v = q ? q[e] - p[e] : p[e];
                if (v == v) {
                    a0[it] += v;
                    a1[it] += v * v;
                    a2[it] += 1.0f;
                    mn[it] = fminf(mn[it], v);
                    mx[it] = fmaxf(mx[it], v);
                }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
        const float v0 = cross_seg_sum(a0[it], FP), v1 = cross_seg_sum(a1[it], FP), v2 = cross_seg_sum(a2[it], FP);
        float lo = mn[it], hi = mx[it];
        for (int off = FP; off < 64; off <<= 1) {   // lanes with the same feature are FP apart
            lo = fminf(lo, __shfl_xor(lo, off));
            hi = fmaxf(hi, __shfl_xor(hi, off));
        }
        if (pp == 0) {
            red[0][wv][it * 64 + c0] = v0; red[1][wv][it * 64 + c0] = v1; red[2][wv][it * 64 + c0] = v2;
            red[3][wv][it * 64 + c0] = lo; red[4][wv][it * 64 + c0] = hi;
        }
    }
    __syncthreads();
    const int B = gridDim.y;
    for (int i = threadIdx.x; i < 5 * iters * FP; i += blockDim.x) {
        const int k = i / (iters * FP), j = i - k * iters * FP;
        const int it = j / FP, c = j - it * FP;
        const int f = c + it * FP;
        if (f < F) {
            const int z = it * 64 + c;
            float v;
            if (k < 3) v = (red[k][0][z] + red[k][1][z]) + (red[k][2][z] + red[k][3][z]);
            else if (k == 3) v = fminf(fminf(red[3][0][z], red[3][1][z]), fminf(red[3][2][z], red[3][3][z]));
            else v = fmaxf(fmaxf(red[4][0][z], red[4][1][z]), fmaxf(red[4][2][z], red[4][3][z]));
            partial[(((int64_t)k * B + b) * gridDim.x + blockIdx.x) * F + f] = v;
        }
    }
}

__global__ void nan_moments_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out, int B, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 5 * B * F) return;
    const int k = i / (B * F), j = i - k * B * F;
    const int b = j / F, f = j - b * F;
    const float* src = partial + (((int64_t)k * B + b) * nblk) * F + f;
    float s = k == 3 ? INFINITY : (k == 4 ? -INFINITY : 0.0f);
    for (int q = 0; q < nblk; ++q) {
        const float v = src[(int64_t)q * F];
        s = k < 3 ? s + v : (k == 3 ? fminf(s, v) : fmaxf(s, v));
    }
    out[i] = s;
}

// ------------------------------------------------------------------ fused AR update + loss, 16-byte vectorised
// Same arithmetic (and operation order) as the scalar kernels below, for F % 4 == 0, F <= 64 and fp32 y/dy rows
// whose stride is a multiple of 4: a lane owns 4 consecutive features of a grid point, so every access is a
// 16-byte load/store and a wave covers 64/FP4 grid points per iteration.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int norm4u __attribute__((ext_vector_type(4)));

// optional extra output of the fused step: the NEXT AR step's network input in the padded layout of p4c_build_x
// (new state | statics | next step's forcing | zero padding), so that the state just computed is not read again
struct NextX {
    void* x;
    int c_pad;
    const float* statics;
    int64_t statics_bs;
    int Fs;
    const float* forcing;
    int64_t forcing_bs;
    int Ff;
    // optional: d loss_elem / d pred of every element as bf16 rows (N, F) per sample -- what the backward otherwise recomputes
    // from the new state and the target (480 bytes per grid point read again; 120 written here and read there instead)
    void* lgrad;
    int64_t lgrad_bs;
};
typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <typename TY, bool NEXT>
__global__ void __launch_bounds__(256)
    ar_update_loss_fwd_v4_kernel(const float* __restrict__ prev, int64_t prev_bs, const TY* __restrict__ y, int y_cs,
                                 const float* __restrict__ target, int64_t tgt_bs, const float* __restrict__ std,
                                 const float* __restrict__ mean, const float* __restrict__ border_mask,
                                 const float* __restrict__ interior_mask, float* __restrict__ new_state, int64_t new_bs,
                                 const float* __restrict__ weights, int kind, int mask_mode, float* __restrict__ partial,
                                 int64_t N, int F, float keep_prev, int FP4, NextX nx) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int PP = 64 / FP4;
    const int pp = lane / FP4, q = lane % FP4;
    const bool act = 4 * q < F;
    const bool from_nan = mask_mode == P4C_MASK_FROM_NAN;
    v4f w = {0, 0, 0, 0}, sd = {1, 1, 1, 1}, mn = {0, 0, 0, 0};
    if (act) {
        w = *reinterpret_cast<const v4f*>(weights + 4 * q);
        if (std) {
            sd = *reinterpret_cast<const v4f*>(std + 4 * q);
            mn = *reinterpret_cast<const v4f*>(mean + 4 * q);
        }
    }
    float acc = 0.0f;
    // tail quad of the next-step input owned by this lane: channels F + 4q .. +3 (statics, forcing, zero padding)
    int tkind = 4;  // 0: 16 bytes from one source, 1: last 1..3 forcing values, 2: zeros, 4: none
    int tvalid = 0;
    const float* tsrc = nullptr;
    int64_t tstride = 0;
    TY* xn = nullptr;
    bf16* lgr = nullptr;
    if (NEXT && nx.lgrad) lgr = reinterpret_cast<bf16*>(nx.lgrad) + (int64_t)b * nx.lgrad_bs;
    if (NEXT && nx.x) {
        xn = reinterpret_cast<TY*>(nx.x) + (int64_t)b * N * nx.c_pad;
        const int c0 = F + 4 * q, o_forc = F + nx.Fs, c_in = F + nx.Fs + nx.Ff;
        if (c0 < nx.c_pad) {
            if (c0 >= c_in) tkind = 2;
            else if (c0 + 3 < o_forc) { tkind = 0; tsrc = nx.statics + (int64_t)b * nx.statics_bs + (c0 - F); tstride = nx.Fs; }
            else {
                tsrc = nx.forcing + (int64_t)b * nx.forcing_bs + (c0 - o_forc);
                tstride = nx.Ff;
                tvalid = c_in - c0 < 4 ? c_in - c0 : 4;
                tkind = tvalid == 4 ? 0 : 1;
            }
        }
    }
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < N; n += stride) {
        if (NEXT && xn && tkind != 4) {
            v4f tq = {0, 0, 0, 0};
            if (tkind == 0) tq = *reinterpret_cast<const v4f_a4*>(tsrc + n * tstride);
            else if (tkind == 1) {
                const float* pf = tsrc + n * tstride;
                tq[0] = pf[0];
                if (tvalid > 1) tq[1] = pf[1];
                if (tvalid > 2) tq[2] = pf[2];
            }
            store4f(xn + n * nx.c_pad + F + 4 * q, tq);
        }
        if (!act) continue;
        const float im = interior_mask[n];
        const float bm = border_mask ? border_mask[n] : 0.0f;
        const int64_t e = n * F + 4 * q;
        const v4f yv = load4f(y + ((int64_t)b * N + n) * y_cs + 4 * q);
        v4f pv = {0, 0, 0, 0};
        if (prev) pv = *reinterpret_cast<const v4f*>(prev + (int64_t)b * prev_bs + e);
        v4f tg = *reinterpret_cast<const v4f*>(target + (int64_t)b * tgt_bs + e);
        v4f o, lg;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float p0 = pv[j], t0 = tg[j], m = 1.0f;
            if (from_nan) {
                p0 = nan_to_zero(p0);
                m = (t0 != t0) ? 0.0f : 1.0f;
                t0 = nan_to_zero(t0);
            }
            float pr;
            if (std) {
                pr = p0 * keep_prev + yv[j] * sd[j];
                pr = pr + mn[j];
            } else {
                pr = p0 * keep_prev + yv[j];
            }
            if (border_mask) pr = bm * t0 + im * pr;
            o[j] = pr;
            s += loss_elem(pr, t0, m, kind) * w[j];
            if (NEXT) lg[j] = loss_elem_grad(pr, t0, m, kind);
        }
        *reinterpret_cast<v4f*>(new_state + (int64_t)b * new_bs + e) = o;
        if (NEXT && xn) store4f(xn + n * nx.c_pad + 4 * q, o);
        if (NEXT && lgr) store4f(lgr + e, lg);
        acc += s * im;
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// SAVED: `new_state` holds bf16 rows of d loss_elem / d pred written by the forward (NextX::lgrad; stride new_bs), `target` is unused
template <typename TY, bool SAVED = false>
__global__ void __launch_bounds__(256)
    ar_update_loss_bwd_v4_kernel(const float* __restrict__ g_next, int64_t g_next_bs, const TY* __restrict__ g_next2,
                                 int g2_cs, const float* __restrict__ gloss, int64_t gloss_stride,
                                 const float* __restrict__ new_state, int64_t new_bs, const float* __restrict__ target,
                                 int64_t tgt_bs, const float* __restrict__ std, const float* __restrict__ interior_mask,
                                 int force_border, const float* __restrict__ weights, float num_interior,
                                 const int32_t* __restrict__ masked_count, int kind, int mask_mode, TY* __restrict__ dy,
                                 int y_cs, float* dprev, int64_t dprev_bs, int64_t N, int F, float keep_prev, int FP4) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int PP = 64 / FP4;
    const int pp = lane / FP4, q = lane % FP4;
    const bool act = 4 * q < F, wr = 4 * q < y_cs;
    const bool from_nan = mask_mode == P4C_MASK_FROM_NAN;
    const float denom = num_interior - (masked_count ? (float)(*masked_count) : 0.0f);
    const float scale = gloss ? gloss[(int64_t)b * gloss_stride] / denom : 0.0f;
    v4f w = {0, 0, 0, 0}, sd = {1, 1, 1, 1};
    if (act) {
        w = *reinterpret_cast<const v4f*>(weights + 4 * q);
        if (std) sd = *reinterpret_cast<const v4f*>(std + 4 * q);
    }
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < N; n += stride) {
        if (!wr) continue;
        v4f gy = {0, 0, 0, 0};
        if (act) {
            const float im = interior_mask[n];
            const float sc = scale * im;
            const float blend = force_border ? im : 1.0f;
            const int64_t e = n * F + 4 * q;
            v4f ns = {0, 0, 0, 0}, tg = {0, 0, 0, 0}, lgv = {0, 0, 0, 0};
            if (SAVED) {
                lgv = load4f(reinterpret_cast<const bf16*>(new_state) + (int64_t)b * new_bs + e);
            } else {
                ns = *reinterpret_cast<const v4f*>(new_state + (int64_t)b * new_bs + e);
                tg = *reinterpret_cast<const v4f*>(target + (int64_t)b * tgt_bs + e);
            }
            v4f g1 = {0, 0, 0, 0}, g2 = {0, 0, 0, 0};
            if (g_next) g1 = *reinterpret_cast<const v4f*>(g_next + (int64_t)b * g_next_bs + e);
            if (g_next2) g2 = load4f(g_next2 + ((int64_t)b * N + n) * g2_cs + 4 * q);
            v4f gp;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t0 = tg[j], m = 1.0f;
                if (from_nan) {
                    m = (t0 != t0) ? 0.0f : 1.0f;
                    t0 = nan_to_zero(t0);
                }
                float g = sc * w[j] * (SAVED ? lgv[j] : loss_elem_grad(ns[j], t0, m, kind));
                if (g_next) g += g1[j];
                if (g_next2) g += g2[j];
                gp[j] = g * blend;
                gy[j] = std ? gp[j] * sd[j] : gp[j];
            }
            if (dprev) *reinterpret_cast<v4f*>(dprev + (int64_t)b * dprev_bs + e) = gp * keep_prev;
        }
        store4f(dy + ((int64_t)b * N + n) * y_cs + 4 * q, gy);
    }
}

// ------------------------------------------------------------------ fused AR update + loss for ANY feature count ("flat" kernels)
// The 16-byte kernels above need F % 4 == 0 (a lane owns 4 features of ONE grid point).  The reference's shipped Titan configuration
// has F = 21 (config/CLI/dataset/titan.yaml:38-76), which left that shape to the scalar kernels: one grid point per wave instruction,
// 21 of 64 lanes busy -- 299 us per backward launch at 2 x 512 x 640 (1.3 TB/s), 10 % of the training step.  Here the (N, F) fp32
// arrays (previous state, target, new state, upstream gradients, saved loss gradients) are streamed FLAT, 16 bytes per lane whatever
// F is (a lane's 4 consecutive elements may straddle two grid points), and the ROW tensors of the network -- y / dy / dx with 64
// channels per grid point, the next network input with c_pad -- pass through LDS tiles of 64 grid points, loaded / stored as whole rows
// with 16-byte slots.  Same arithmetic per element as the scalar kernels (bit-identical new state, dy, dprev); the loss is the same
// sum in another order.  Needs N * F % 4 == 0 per sample, 16-byte aligned rows, no NaN masks.
constexpr int FLAT_P = 64;   // grid points per tile

// (pl, f) of flat element e of a tile, e < 4096, F <= 64: exact multiply-shift division
__device__ __forceinline__ void flat_pf(int e, int F, unsigned rcp, int& pl, int& f) {
    pl = (int)(((unsigned)e * rcp) >> 20);
    f = e - pl * F;
}

// FRONT (bf16 rows only): the y tile is not read from memory but FORMED -- the network's 1x1 output convolution on relu(norm(a)) runs on
// the matrix cores at the head of every tile (the front end of out_conv_update_loss_fwd_kernel below: the same MFMA chain per element
// as the convolution's own launch, one rounding to bf16), so y is never written and read back for feature counts off the 16-byte grid
// either (the shipped Titan configuration: F = 21).  A wave = 32 grid points x 32 output features; the tile's 8-byte slots are XOR-ed
// with the grid-point index like that kernel's.
struct FlatFront {
    const bf16* a;            // (B,N,64) bf16: raw output of the network's last block
    const float* a_scale;     // (B,64) each
    const float* a_shift;
    const float* wout;        // (cout,64) fp32
    int cout;
};
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

template <typename TY, bool FRONT = false>
__global__ void __launch_bounds__(256)
    ar_update_loss_fwd_flat_kernel(const float* __restrict__ prev, int64_t prev_bs, const TY* __restrict__ y, int y_cs,
                                   const float* __restrict__ target, int64_t tgt_bs, const float* __restrict__ std,
                                   const float* __restrict__ mean, const float* __restrict__ border_mask,
                                   const float* __restrict__ interior_mask, float* __restrict__ new_state, int64_t new_bs,
                                   const float* __restrict__ weights, int kind, float* __restrict__ partial, int64_t N, int F,
                                   float keep_prev, NextX nx, FlatFront fa) {
    static_assert(!FRONT || std::is_same<TY, bf16>::value, "the fused front end forms bf16 rows");
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    // LDS: y tile [64][y_cs] | x tile [64][c_pad] (next input, if any) | per-feature constants 3 x 64 floats | masks 2 x 64 floats
    //      | FRONT: A fragments of the 1x1 weight 2 x 4 x 64 x 16 bytes | the sample's scale, shift 2 x 64 floats
    TY* ytile = reinterpret_cast<TY*>(fsm);
    TY* xtile = ytile + FLAT_P * y_cs;
    float* cst = reinterpret_cast<float*>(xtile + (nx.x ? FLAT_P * nx.c_pad : 0));
    float* msk = cst + 3 * 64;
    bf16x8_t* wimg = reinterpret_cast<bf16x8_t*>(msk + 2 * 64);
    float* lsc = reinterpret_cast<float*>(wimg + 2 * 4 * 64);
    float* lsh = lsc + 64;
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y;
    if (FRONT) {
        for (int t = tid; t < 2 * 4 * 64; t += 256) {   // (T, ks, lane) = W[32 T + (l & 31)][16 ks + 8 (l >> 5) .. + 7]
            const int l = t & 63, ks = (t >> 6) & 3, T = t >> 8;
            const int co = 32 * T + (l & 31), k0 = 16 * ks + 8 * (l >> 5);
            v4f lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
            if (co < fa.cout) {
                lo = *reinterpret_cast<const v4f*>(fa.wout + (int64_t)co * 64 + k0);
                hi = *reinterpret_cast<const v4f*>(fa.wout + (int64_t)co * 64 + k0 + 4);
            }
            bf16x8_t w8;
#pragma unroll
            for (int j = 0; j < 4; ++j) { w8[j] = (__bf16)lo[j]; w8[4 + j] = (__bf16)hi[j]; }
            wimg[t] = w8;
        }
        if (tid < 64) {
            lsc[tid] = fa.a_scale[b * 64 + tid];
            lsh[tid] = fa.a_shift[b * 64 + tid];
        }
    }
    const unsigned rcp = ((1u << 20) + F - 1) / F;
    if (tid < 64) {
        const bool ok = tid < F;
        cst[tid] = ok ? weights[tid] : 0.f;
        cst[64 + tid] = (ok && std) ? std[tid] : 1.f;
        cst[128 + tid] = (ok && mean) ? mean[tid] : 0.f;
    }
    const float* prevb = prev ? prev + (int64_t)b * prev_bs : nullptr;
    const float* tgtb = target + (int64_t)b * tgt_bs;
    float* newb = new_state + (int64_t)b * new_bs;
    const TY* yb = y + (int64_t)b * N * y_cs;
    TY* xn = nx.x ? reinterpret_cast<TY*>(nx.x) + (int64_t)b * N * nx.c_pad : nullptr;
    bf16* lgr = nx.lgrad ? reinterpret_cast<bf16*>(nx.lgrad) + (int64_t)b * nx.lgrad_bs : nullptr;
    const float* stat = xn ? nx.statics + (int64_t)b * nx.statics_bs : nullptr;
    const float* forc = xn ? nx.forcing + (int64_t)b * nx.forcing_bs : nullptr;
    const int yslots = y_cs * (int)sizeof(TY) / 16, xslots = xn ? nx.c_pad * (int)sizeof(TY) / 16 : 0;
    const int64_t ntiles = (N + FLAT_P - 1) / FLAT_P;
    // statics / forcing rows as flat 16-byte streams (whole tiles start on 16-byte boundaries; <= 64 channels each for the
    // multiply-shift division); otherwise a lane per channel
    const bool tail_flat = xn && nx.Fs > 0 && nx.Ff > 0 && nx.Fs <= 64 && nx.Ff <= 64 && nx.statics_bs % 4 == 0 && nx.forcing_bs % 4 == 0 &&
                           (reinterpret_cast<uintptr_t>(nx.statics) & 15) == 0 && (reinterpret_cast<uintptr_t>(nx.forcing) & 15) == 0 &&
                           (N * nx.Fs) % 4 == 0 && (N * nx.Ff) % 4 == 0;
    const unsigned rcp_s = tail_flat ? ((1u << 20) + nx.Fs - 1) / nx.Fs : 0u, rcp_f = tail_flat ? ((1u << 20) + nx.Ff - 1) / nx.Ff : 0u;
    if (tail_flat)   // the zero padding beyond the forcing channels: laid once, never written again
        for (int i = tid; i < FLAT_P * (nx.c_pad - F - nx.Fs - nx.Ff); i += 256) {
            const int w = nx.c_pad - F - nx.Fs - nx.Ff, pl = i / w;
            xtile[pl * nx.c_pad + F + nx.Fs + nx.Ff + (i - pl * w)] = from_f32<TY>(0.f);
        }
    float acc = 0.f;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t n0 = t * FLAT_P;
        const int np = (int)((N - n0) < FLAT_P ? (N - n0) : FLAT_P);
        __syncthreads();   // (the previous tile's rows are out; the constants are in)
        // ---- stage: y rows (whole rows, 16-byte slots), the masks of the tile's grid points
        if (FRONT) {
            // y^T = W relu(norm(a))^T for this wave's 32 grid points x 32 features (the first trip's barrier above covers wimg / lsc)
            const int r = lane & 31, h = lane >> 5, pb = wv & 1, T = wv >> 1;
            if (32 * T < fa.cout) {
                const int64_t pn = n0 + 32 * pb + r;
                const bf16* ab = fa.a + (int64_t)b * N * 64;
                u32x4_t cur[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    cur[ks] = u32x4_t{0u, 0u, 0u, 0u};
                    if (pn < N) cur[ks] = *reinterpret_cast<const u32x4_t*>(ab + pn * 64 + 16 * ks + 8 * h);
                }
                f32x16_t yy;
#pragma unroll
                for (int i = 0; i < 16; ++i) yy[i] = 0.f;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const v4f s0 = *reinterpret_cast<const v4f*>(lsc + 16 * ks + 8 * h), s1 = *reinterpret_cast<const v4f*>(lsc + 16 * ks + 8 * h + 4);
                    const v4f t0 = *reinterpret_cast<const v4f*>(lsh + 16 * ks + 8 * h), t1 = *reinterpret_cast<const v4f*>(lsh + 16 * ks + 8 * h + 4);
                    const float scv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
                    const float shv[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                    u32x4_t o;
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) {
                        // relu(a * scale + shift), fp32, one rounding to bf16: what the 1x1 convolution's loader stages (conv_rows.hip: xform2<2>)
                        const unsigned int wd = cur[ks][k2];
                        const float lo = __builtin_fmaf(__builtin_bit_cast(float, wd << 16), scv[2 * k2], shv[2 * k2]);
                        const float hi = __builtin_fmaf(__builtin_bit_cast(float, wd & 0xffff0000u), scv[2 * k2 + 1], shv[2 * k2 + 1]);
                        const s16x2_t z = {0, 0};
                        o[k2] = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, pack_bf16(lo, hi)), z));
                    }
                    yy = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wimg[(T * 4 + ks) * 64 + lane], __builtin_bit_cast(bf16x8_t, o), yy, 0, 0, 0);
                }
                // C[feature][point]: lane = point r (+ half h), register quad gq -> features 32 T + 8 gq + 4 h .. + 3
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    u32x2_t o;
                    o[0] = pack_bf16(yy[4 * gq], yy[4 * gq + 1]);
                    o[1] = pack_bf16(yy[4 * gq + 2], yy[4 * gq + 3]);
                    *reinterpret_cast<u32x2_t*>(reinterpret_cast<bf16*>(ytile) + (32 * pb + r) * 64 + 4 * ((8 * T + 2 * gq + h) ^ (r & 15))) = o;
                }
            }
        } else {
            for (int sl = tid; sl < np * yslots; sl += 256)
                reinterpret_cast<norm4u*>(ytile)[sl] = reinterpret_cast<const norm4u*>(yb + n0 * y_cs)[sl];
        }
        if (tid < 64) {
            const bool ok = tid < np;
            msk[tid] = ok ? interior_mask[n0 + tid] : 0.f;
            msk[64 + tid] = (ok && border_mask) ? border_mask[n0 + tid] : 0.f;
        }
        // ---- the next input's channels F .. c_pad - 1: statics | next forcing | zeros (lightning.py:760-765)
        if (xn && tail_flat) {
            // statics and forcing streamed FLAT like the state (16 bytes per lane; the zero padding was laid once, above)
            const int nst = np * nx.Fs, nfo = np * nx.Ff;
            for (int e0 = 4 * tid; e0 < nst; e0 += 4 * 256) {
                const v4f v = *reinterpret_cast<const v4f*>(stat + n0 * nx.Fs + e0);
                int pl, f;
                flat_pf(e0, nx.Fs, rcp_s, pl, f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (e0 + j < nst) xtile[pl * nx.c_pad + F + f] = from_f32<TY>(v[j]);
                    if (++f == nx.Fs) { f = 0; ++pl; }
                }
            }
            for (int e0 = 4 * tid; e0 < nfo; e0 += 4 * 256) {
                const v4f v = *reinterpret_cast<const v4f*>(forc + n0 * nx.Ff + e0);
                int pl, f;
                flat_pf(e0, nx.Ff, rcp_f, pl, f);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (e0 + j < nfo) xtile[pl * nx.c_pad + F + nx.Fs + f] = from_f32<TY>(v[j]);
                    if (++f == nx.Ff) { f = 0; ++pl; }
                }
            }
        } else if (xn) {
            const int ntail = nx.c_pad - F;
            for (int pl = tid >> 6; pl < np; pl += 4)          // a wave per grid point, a lane per channel: no divisions
                for (int c = tid & 63; c < ntail; c += 64) {
                    float v = 0.f;
                    if (c < nx.Fs) v = stat[(n0 + pl) * nx.Fs + c];
                    else if (c < nx.Fs + nx.Ff) v = forc[(n0 + pl) * nx.Ff + (c - nx.Fs)];
                    xtile[pl * nx.c_pad + F + c] = from_f32<TY>(v);
                }
        }
        __syncthreads();
        // ---- flat phase: 4 consecutive elements of the (np, F) block per lane and trip
        const int nel = np * F;   // a multiple of 4 (host: N * F % 4 == 0; tiles of 64 grid points)
        for (int e0 = 4 * tid; e0 < nel; e0 += 4 * 256) {
            const int64_t E = n0 * F + e0;
            v4f pv = {0, 0, 0, 0};
            if (prevb) pv = *reinterpret_cast<const v4f*>(prevb + E);
            const v4f tg = *reinterpret_cast<const v4f*>(tgtb + E);
            int pl, f;
            flat_pf(e0, F, rcp, pl, f);
            v4f o, lg;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float yv = to_f32<TY>(FRONT ? ytile[pl * 64 + 4 * ((f >> 2) ^ (pl & 15)) + (f & 3)] : ytile[pl * y_cs + f]);
                const float im = msk[pl], bm = msk[64 + pl];
                const float p0 = pv[j], t0 = tg[j];
                float pr;
                if (std) {
                    pr = p0 * keep_prev + yv * cst[64 + f];
                    pr = pr + cst[128 + f];
                } else {
                    pr = p0 * keep_prev + yv;
                }
                if (border_mask) pr = bm * t0 + im * pr;
                o[j] = pr;
                acc += (loss_elem(pr, t0, 1.0f, kind) * cst[f]) * im;
                lg[j] = loss_elem_grad(pr, t0, 1.0f, kind);
                if (xn) xtile[pl * nx.c_pad + f] = from_f32<TY>(pr);
                if (++f == F) { f = 0; ++pl; }
            }
            *reinterpret_cast<v4f*>(newb + E) = o;
            if (lgr) store4f(lgr + E, lg);
        }
        if (xn) {
            __syncthreads();
            for (int sl = tid; sl < np * xslots; sl += 256)
                reinterpret_cast<norm4u*>(xn + n0 * nx.c_pad)[sl] = reinterpret_cast<const norm4u*>(xtile)[sl];
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (tid == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// SAVED: `new_state` holds the bf16 rows of d loss_elem / d pred the forward saved (stride new_bs), `target` is unused
template <typename TY, bool SAVED>
__global__ void __launch_bounds__(256)
    ar_update_loss_bwd_flat_kernel(const float* __restrict__ g_next, int64_t g_next_bs, const TY* __restrict__ g_next2, int g2_cs,
                                   const float* __restrict__ gloss, int64_t gloss_stride, const float* __restrict__ new_state,
                                   int64_t new_bs, const float* __restrict__ target, int64_t tgt_bs, const float* __restrict__ std,
                                   const float* __restrict__ interior_mask, int force_border, const float* __restrict__ weights,
                                   float num_interior, const int32_t* __restrict__ masked_count, int kind, TY* __restrict__ dy,
                                   int y_cs, float* dprev, int64_t dprev_bs, int64_t N, int F, float keep_prev) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    // LDS: g2 tile [64][g2_cs] (upstream gradient rows, if any) | dy tile [64][y_cs] | weights, std 2 x 64 floats | interior mask 64 floats
    TY* g2tile = reinterpret_cast<TY*>(fsm);
    TY* dytile = g2tile + (g_next2 ? FLAT_P * g2_cs : 0);
    float* cst = reinterpret_cast<float*>(dytile + FLAT_P * y_cs);
    float* msk = cst + 2 * 64;
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const unsigned rcp = ((1u << 20) + F - 1) / F;
    const float denom = num_interior - (masked_count ? (float)(*masked_count) : 0.0f);
    const float scale = gloss ? gloss[(int64_t)b * gloss_stride] / denom : 0.0f;
    if (tid < 64) {
        const bool ok = tid < F;
        cst[tid] = ok ? weights[tid] : 0.f;
        cst[64 + tid] = (ok && std) ? std[tid] : 1.f;
    }
    // channels F .. y_cs - 1 of dy are zero: cleared once, the flat phase only ever writes channels < F
    for (int i = tid; i < FLAT_P * y_cs; i += 256) dytile[i] = from_f32<TY>(0.f);
    const float* g1b = g_next ? g_next + (int64_t)b * g_next_bs : nullptr;
    const TY* g2b = g_next2 ? g_next2 + (int64_t)b * N * g2_cs : nullptr;
    const float* nsb = SAVED ? nullptr : new_state + (int64_t)b * new_bs;
    const bf16* lgb = SAVED ? reinterpret_cast<const bf16*>(new_state) + (int64_t)b * new_bs : nullptr;
    const float* tgtb = SAVED ? nullptr : target + (int64_t)b * tgt_bs;
    TY* dyb = dy + (int64_t)b * N * y_cs;
    float* dpb = dprev ? dprev + (int64_t)b * dprev_bs : nullptr;
    const int gslots = g2_cs * (int)sizeof(TY) / 16, yslots = y_cs * (int)sizeof(TY) / 16;
    const int64_t ntiles = (N + FLAT_P - 1) / FLAT_P;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t n0 = t * FLAT_P;
        const int np = (int)((N - n0) < FLAT_P ? (N - n0) : FLAT_P);
        __syncthreads();   // (the previous tile's dy rows are out)
        if (g2b)
            for (int sl = tid; sl < np * gslots; sl += 256)
                reinterpret_cast<norm4u*>(g2tile)[sl] = reinterpret_cast<const norm4u*>(g2b + n0 * g2_cs)[sl];
        if (tid < 64) msk[tid] = tid < np ? interior_mask[n0 + tid] : 0.f;
        __syncthreads();
        const int nel = np * F;
        for (int e0 = 4 * tid; e0 < nel; e0 += 4 * 256) {
            const int64_t E = n0 * F + e0;
            v4f ns = {0, 0, 0, 0}, tg = {0, 0, 0, 0}, lgv = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
            if (SAVED) {
                lgv = load4f(lgb + E);
            } else {
                ns = *reinterpret_cast<const v4f*>(nsb + E);
                tg = *reinterpret_cast<const v4f*>(tgtb + E);
            }
            if (g1b) g1 = *reinterpret_cast<const v4f*>(g1b + E);
            int pl, f;
            flat_pf(e0, F, rcp, pl, f);
            v4f gp;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float im = msk[pl];
                const float sc = scale * im;
                const float blend = force_border ? im : 1.0f;
                float g = sc * cst[f] * (SAVED ? lgv[j] : loss_elem_grad(ns[j], tg[j], 1.0f, kind));
                if (g1b) g += g1[j];
                if (g2b) g += to_f32<TY>(g2tile[pl * g2_cs + f]);
                gp[j] = g * blend;
                dytile[pl * y_cs + f] = from_f32<TY>(std ? gp[j] * cst[64 + f] : gp[j]);
                if (++f == F) { f = 0; ++pl; }
            }
            if (dpb) *reinterpret_cast<v4f*>(dpb + E) = gp * keep_prev;
        }
        __syncthreads();
        for (int sl = tid; sl < np * yslots; sl += 256)
            reinterpret_cast<norm4u*>(dyb + n0 * y_cs)[sl] = reinterpret_cast<const norm4u*>(dytile)[sl];
    }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// the flat kernels' conditions (beyond "the 16-byte kernels do not apply"): no NaN masks (checked by the callers), F <= 64,
// whole 16-byte slots per row of the row tensors, N * F % 4 == 0, 16-byte aligned bases and batch strides
// P4C_FORCE_FLAT_STEP=1 (tests): the flat kernels also where the 16-byte ones apply -- same results element for element
static inline bool force_flat() {
    const char* e = diag_env("P4C_FORCE_FLAT_STEP");
    return e && e[0] == '1';
}
static inline bool flat_ok(int F, int64_t N, int row_cs, int esz) {
    const char* e = diag_env("P4C_NO_FLAT_STEP");   // (read per call: the parity tests switch the flat kernels off)
    if (e && e[0] == '1') return false;
    return F > 0 && F <= 64 && row_cs >= F && (row_cs * esz) % 16 == 0 && row_cs <= 256 && (N * F) % 4 == 0;
}

// ------------------------------------------------------------------ fused AR update + loss (training path)
// grid: (nblk, B).  One (b, t=i) column of the loss.
template <typename TY>
__global__ void __launch_bounds__(256)
    ar_update_loss_fwd_kernel(const float* __restrict__ prev, int64_t prev_bs, const TY* __restrict__ y, int y_cs,
                              const float* __restrict__ target, int64_t tgt_bs, const float* __restrict__ std,
                              const float* __restrict__ mean, const float* __restrict__ border_mask,
                              const float* __restrict__ interior_mask, float* __restrict__ new_state, int64_t new_bs,
                              const float* __restrict__ weights, int kind, int mask_mode, float* __restrict__ partial,
                              int64_t N, int F, float keep_prev, int FP, int iters) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const bool from_nan = mask_mode == P4C_MASK_FROM_NAN;
    float w[LOSS_MAX_ITERS], sd[LOSS_MAX_ITERS], mn[LOSS_MAX_ITERS];
#pragma unroll
    for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
        const int f = c0 + it * FP;
        const bool ok = it < iters && f < F;
        w[it] = ok ? weights[f] : 0.0f;
        sd[it] = (ok && std) ? std[f] : 1.0f;
        mn[it] = (ok && mean) ? mean[f] : 0.0f;
    }
    float acc = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < N; n += stride) {
        const float im = interior_mask[n];
        const float bm = border_mask ? border_mask[n] : 0.0f;
        float s = 0.0f;
#pragma unroll
        for (int it = 0; it < LOSS_MAX_ITERS; ++it) {
            const int f = c0 + it * FP;
            if (it < iters && f < F) {
                const int64_t e = n * F + f;
                const float yv = to_f32<TY>(y[((int64_t)b * N + n) * y_cs + f]);
                float pv = 0.0f;
                if (prev) {
                    pv = prev[(int64_t)b * prev_bs + e];
                    if (from_nan) pv = nan_to_zero(pv);
                }
                float tg = target[(int64_t)b * tgt_bs + e];
                float m = 1.0f;
                if (from_nan) {
                    m = (tg != tg) ? 0.0f : 1.0f;
                    tg = nan_to_zero(tg);
                }
                float pr;
                if (std) {
                    pr = pv * keep_prev + yv * sd[it];
                    pr = pr + mn[it];
                } else {
                    pr = pv * keep_prev + yv;
                }
                if (border_mask) pr = bm * tg + im * pr;
                new_state[(int64_t)b * new_bs + e] = pr;
                s += loss_elem(pr, tg, m, kind) * w[it];
            }
        }
        acc += s * im;
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

template <typename TY, typename TG>
__global__ void __launch_bounds__(256)
    ar_update_loss_bwd_kernel(const float* __restrict__ g_next, int64_t g_next_bs, const TG* __restrict__ g_next2,
                              int g2_cs, const float* __restrict__ gloss, int64_t gloss_stride,
                              const float* __restrict__ new_state, int64_t new_bs, const float* __restrict__ target,
                              int64_t tgt_bs, const float* __restrict__ std, const float* __restrict__ interior_mask,
                              int force_border, const float* __restrict__ weights, float num_interior,
                              const int32_t* __restrict__ masked_count, int kind, int mask_mode, TY* __restrict__ dy,
                              int y_cs, float* __restrict__ dprev, int64_t dprev_bs, int64_t N, int F, float keep_prev,
                              int FP, int iters) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const bool from_nan = mask_mode == P4C_MASK_FROM_NAN;
    const float denom = num_interior - (masked_count ? (float)(*masked_count) : 0.0f);
    const float scale = gloss ? gloss[(int64_t)b * gloss_stride] / denom : 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 4 * PP;
    for (int64_t n = ((int64_t)blockIdx.x * 4 + wv) * PP + pp; n < N; n += stride) {
        const float im = interior_mask[n];
        const float sc = scale * im;
        const float blend = force_border ? im : 1.0f;
        for (int it = 0; it < iters; ++it) {
            const int f = c0 + it * FP;
            if (f >= y_cs) continue;
            float gy = 0.0f;
            if (f < F) {
                const int64_t e = n * F + f;
                float tg = target[(int64_t)b * tgt_bs + e];
                float m = 1.0f;
                if (from_nan) {
                    m = (tg != tg) ? 0.0f : 1.0f;
                    tg = nan_to_zero(tg);
                }
                float g = sc * weights[f] * loss_elem_grad(new_state[(int64_t)b * new_bs + e], tg, m, kind);
                if (g_next) g += g_next[(int64_t)b * g_next_bs + e];
                if (g_next2) g += to_f32<TG>(g_next2[((int64_t)b * N + n) * g2_cs + f]);
                const float gp = g * blend;
                gy = std ? gp * std[f] : gp;
                if (dprev) dprev[(int64_t)b * dprev_bs + e] = gp * keep_prev;
            }
            dy[((int64_t)b * N + n) * y_cs + f] = from_f32<TY>(gy);
        }
    }
}

static inline int loss_blocks(int64_t N, int PP, int nbt) {
    int64_t waves = (N + PP - 1) / PP;
    int64_t blocks = (waves + 3) / 4;
    // a few grid points per lane at least, bounded by the workspace contract and by ~8 blocks/CU overall
    int64_t cap = ((int64_t)num_cus() * 8 + nbt - 1) / nbt;
    if (cap < 1) cap = 1;
    if (blocks > cap) blocks = cap;
    if (blocks > LOSS_MAX_BLOCKS) blocks = LOSS_MAX_BLOCKS;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

static inline bool mask_strides(int mask_mode, const void* mask, int64_t tgt_bs, int64_t tgt_ts, int T, int64_t N, int F,
                                int64_t& mbs, int64_t& mts) {
    // explicit masks are dense (B,T,N,F); FROM_NAN reads the target itself
    mts = N * F;
    mbs = (int64_t)T * N * F;
    if (mask_mode == P4C_MASK_F32 || mask_mode == P4C_MASK_U8) return mask != nullptr;
    return true;
}

}  // namespace p4c

using namespace p4c;

extern "C" size_t p4c_loss_workspace_bytes(int B, int T, int64_t N, int F) {
    (void)N;
    return (size_t)B * (size_t)T * LOSS_MAX_BLOCKS * (size_t)(F > 1 ? F : 1) * sizeof(float);
}

extern "C" int p4c_mask_all_zero_count(const void* mask_or_target, int mask_mode, int64_t bs, int64_t ts, int B, int T,
                                       int64_t N, int F, int32_t* count, p4c_stream_t stream) {
    P4C_CHECK_ARG(count, "p4c_mask_all_zero_count: null count");
    P4C_CHECK_HIP(zero_words_async(count, sizeof(int32_t), as_stream(stream)));   // (not a memset: common.hpp)
    if (mask_mode == P4C_MASK_NONE) return P4C_OK;
    P4C_CHECK_ARG(mask_or_target, "p4c_mask_all_zero_count: null mask");
    P4C_CHECK_ARG(F <= 64 * LOSS_MAX_ITERS, "p4c_mask_all_zero_count: F too large");
    const int FP = pow2_ge64(F), iters = (F + FP - 1) / FP;
    MaskArg ma{mask_or_target, mask_mode};
    int64_t waves = (N + (64 / FP) - 1) / (64 / FP);
    int blocks = (int)((waves + 3) / 4);
    if (blocks > num_cus() * 8) blocks = num_cus() * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(mask_all_zero_count_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), ma, bs, ts, B, T, N, F,
                       FP, iters, count);
    P4C_CHECK_LAUNCH("p4c_mask_all_zero_count");
    return P4C_OK;
}

extern "C" int p4c_weighted_loss_fwd(const float* pred, int64_t pred_bs, int64_t pred_ts, const float* target,
                                     int64_t tgt_bs, int64_t tgt_ts, const void* mask, int mask_mode,
                                     const float* weights, const float* interior_mask, float num_interior,
                                     const int32_t* masked_count, int kind, float* out, void* workspace, int B, int T,
                                     int64_t N, int F, p4c_stream_t stream) {
    P4C_CHECK_ARG(pred && target && weights && interior_mask && out && workspace, "p4c_weighted_loss_fwd: null pointer");
    P4C_CHECK_ARG(F > 0 && F <= 64 * LOSS_MAX_ITERS, "p4c_weighted_loss_fwd: F=%d unsupported", F);
    P4C_CHECK_ARG(kind == P4C_LOSS_MSE || kind == P4C_LOSS_L1, "p4c_weighted_loss_fwd: bad loss kind");
    int64_t mbs, mts;
    P4C_CHECK_ARG(mask_strides(mask_mode, mask, tgt_bs, tgt_ts, T, N, F, mbs, mts), "p4c_weighted_loss_fwd: null mask");
    const int FP = pow2_ge64(F), iters = (F + FP - 1) / FP;
    const int nblk = loss_blocks(N, 64 / FP, B * T);
    MaskArg ma{mask, mask_mode};
    hipLaunchKernelGGL(weighted_loss_partial_kernel, dim3(nblk, B * T), dim3(256), 0, as_stream(stream), pred, pred_bs,
                       pred_ts, target, tgt_bs, tgt_ts, ma, mbs, mts, weights, interior_mask, kind, (float*)workspace, T,
                       N, F, FP, iters);
    P4C_CHECK_LAUNCH("p4c_weighted_loss_fwd(partial)");
    hipLaunchKernelGGL(weighted_loss_final_kernel, dim3(B * T), dim3(64), 0, as_stream(stream),
                       (const float*)workspace, nblk, num_interior, masked_count, out, (int64_t)1, B * T);
    P4C_CHECK_LAUNCH("p4c_weighted_loss_fwd(final)");
    return P4C_OK;
}

extern "C" int p4c_weighted_loss_map(const float* pred, int64_t pred_bs, int64_t pred_ts, const float* target,
                                     int64_t tgt_bs, int64_t tgt_ts, const void* mask, int mask_mode,
                                     const float* weights, int kind, float* out_map, int B, int T, int64_t N, int F,
                                     p4c_stream_t stream) {
    P4C_CHECK_ARG(pred && target && weights && out_map, "p4c_weighted_loss_map: null pointer");
    P4C_CHECK_ARG(F > 0 && F <= 64 * LOSS_MAX_ITERS, "p4c_weighted_loss_map: F=%d unsupported", F);
    int64_t mbs, mts;
    P4C_CHECK_ARG(mask_strides(mask_mode, mask, tgt_bs, tgt_ts, T, N, F, mbs, mts), "p4c_weighted_loss_map: null mask");
    const int FP = pow2_ge64(F), iters = (F + FP - 1) / FP;
    const int nblk = loss_blocks(N, 64 / FP, B * T);
    MaskArg ma{mask, mask_mode};
    hipLaunchKernelGGL(weighted_loss_map_kernel, dim3(nblk, B * T), dim3(256), 0, as_stream(stream), pred, pred_bs,
                       pred_ts, target, tgt_bs, tgt_ts, ma, mbs, mts, weights, kind, out_map, T, N, F, FP, iters);
    P4C_CHECK_LAUNCH("p4c_weighted_loss_map");
    return P4C_OK;
}

extern "C" int p4c_weighted_loss_bwd(const float* gout, const float* pred, int64_t pred_bs, int64_t pred_ts,
                                     const float* target, int64_t tgt_bs, int64_t tgt_ts, const void* mask,
                                     int mask_mode, const float* weights, const float* interior_mask,
                                     float num_interior, const int32_t* masked_count, int kind, float* dpred,
                                     int64_t dpred_bs, int64_t dpred_ts, int B, int T, int64_t N, int F,
                                     p4c_stream_t stream) {
    P4C_CHECK_ARG(gout && pred && target && weights && interior_mask && dpred, "p4c_weighted_loss_bwd: null pointer");
    P4C_CHECK_ARG(F > 0 && F <= 64 * LOSS_MAX_ITERS, "p4c_weighted_loss_bwd: F=%d unsupported", F);
    int64_t mbs, mts;
    P4C_CHECK_ARG(mask_strides(mask_mode, mask, tgt_bs, tgt_ts, T, N, F, mbs, mts), "p4c_weighted_loss_bwd: null mask");
    const int FP = pow2_ge64(F), iters = (F + FP - 1) / FP;
    const int nblk = loss_blocks(N, 64 / FP, B * T);
    MaskArg ma{mask, mask_mode};
    hipLaunchKernelGGL(weighted_loss_bwd_kernel, dim3(nblk, B * T), dim3(256), 0, as_stream(stream), gout, pred, pred_bs,
                       pred_ts, target, tgt_bs, tgt_ts, ma, mbs, mts, weights, interior_mask, num_interior, masked_count,
                       kind, dpred, dpred_bs, dpred_ts, T, N, F, FP, iters);
    P4C_CHECK_LAUNCH("p4c_weighted_loss_bwd");
    return P4C_OK;
}

extern "C" int p4c_scaled_loss_fwd(const float* pred, int64_t pred_bs, int64_t pred_ts, const float* target,
                                   int64_t tgt_bs, int64_t tgt_ts, const void* mask, int mask_mode, const float* std,
                                   const float* interior_mask, float num_interior, const int32_t* masked_count,
                                   int kind, float* out, void* workspace, int B, int T, int64_t N, int F,
                                   p4c_stream_t stream) {
    P4C_CHECK_ARG(pred && target && std && interior_mask && out && workspace, "p4c_scaled_loss_fwd: null pointer");
    P4C_CHECK_ARG(F > 0 && F <= 64 * LOSS_MAX_ITERS, "p4c_scaled_loss_fwd: F=%d unsupported", F);
    int64_t mbs, mts;
    P4C_CHECK_ARG(mask_strides(mask_mode, mask, tgt_bs, tgt_ts, T, N, F, mbs, mts), "p4c_scaled_loss_fwd: null mask");
    const int FP = pow2_ge64(F), iters = (F + FP - 1) / FP;
    const int nblk = loss_blocks(N, 64 / FP, B * T);
    MaskArg ma{mask, mask_mode};
    hipLaunchKernelGGL(scaled_loss_partial_kernel, dim3(nblk, B * T), dim3(256), 0, as_stream(stream), pred, pred_bs,
                       pred_ts, target, tgt_bs, tgt_ts, ma, mbs, mts, interior_mask, kind, (float*)workspace, T, N, F, FP,
                       iters);
    P4C_CHECK_LAUNCH("p4c_scaled_loss_fwd(partial)");
    const int tot = B * T * F;
    hipLaunchKernelGGL(scaled_loss_final_kernel, dim3((tot + 127) / 128), dim3(128), 0, as_stream(stream),
                       (const float*)workspace, nblk, num_interior, masked_count, std, kind, out, B * T, F);
    P4C_CHECK_LAUNCH("p4c_scaled_loss_fwd(final)");
    return P4C_OK;
}

extern "C" int p4c_acc_sums(const float* pred, int64_t pred_bs, int64_t pred_ts, const float* target, int64_t tgt_bs,
                            int64_t tgt_ts, const void* mask, int mask_mode, const float* climate_means, float* out,
                            void* workspace, int B, int T, int64_t N, int F, p4c_stream_t stream) {
    P4C_CHECK_ARG(pred && target && climate_means && out && workspace, "p4c_acc_sums: null pointer");
    P4C_CHECK_ARG(B > 0 && T > 0 && N > 0 && F > 0 && F <= 64 * LOSS_MAX_ITERS, "p4c_acc_sums: bad dims");
    P4C_CHECK_ARG(mask_mode == P4C_MASK_NONE || mask_mode == P4C_MASK_FROM_NAN || mask, "p4c_acc_sums: mask pointer needed");
    int nblk;
    if ((mask_mode == P4C_MASK_NONE || mask_mode == P4C_MASK_FROM_NAN) && F % 4 == 0 && F <= 64 && pred_bs % 4 == 0 &&
        pred_ts % 4 == 0 && tgt_bs % 4 == 0 && tgt_ts % 4 == 0 && aligned16(pred) && aligned16(target) && aligned16(climate_means)) {
        const int FP4 = pow2_ge64(F / 4);
        nblk = loss_blocks(N, 64 / FP4, B * T);
        hipLaunchKernelGGL(acc_partial_v4_kernel, dim3(nblk, B * T), dim3(256), 0, as_stream(stream), pred, pred_bs, pred_ts,
                           target, tgt_bs, tgt_ts, mask_mode == P4C_MASK_FROM_NAN ? 1 : 0, climate_means, (float*)workspace, T, N,
                           F, FP4);
    } else {
        const int FP = pow2_ge64(F), iters = (F + FP - 1) / FP;
        nblk = loss_blocks(N, 64 / FP, B * T);
        const int64_t mbs = (int64_t)T * N * F, mts = N * F;
        MaskArg ma{mask, mask_mode};
        hipLaunchKernelGGL(acc_partial_kernel, dim3(nblk, B * T), dim3(256), 0, as_stream(stream), pred, pred_bs, pred_ts,
                           target, tgt_bs, tgt_ts, ma, mbs, mts, climate_means, (float*)workspace, T, N, F, FP, iters);
    }
    P4C_CHECK_LAUNCH("p4c_acc_sums(partial)");
    const int tot = 3 * B * T * F;
    hipLaunchKernelGGL(acc_final_kernel, dim3((tot + 127) / 128), dim3(128), 0, as_stream(stream), (const float*)workspace,
                       nblk, (float)N, out, B * T, F);
    P4C_CHECK_LAUNCH("p4c_acc_sums(final)");
    return P4C_OK;
}

extern "C" int p4c_nan_moments(const float* x, const float* x_next, int64_t batch_stride, float* out, void* workspace,
                               int B, int64_t rows, int F, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && out && workspace, "p4c_nan_moments: null pointer");
    P4C_CHECK_ARG(B > 0 && rows > 0 && F > 0 && F <= 64 * LOSS_MAX_ITERS, "p4c_nan_moments: bad dims");
    const int FP = pow2_ge64(F), iters = (F + FP - 1) / FP;
    const int nblk = loss_blocks(rows, 64 / FP, B);
    hipLaunchKernelGGL(nan_moments_partial_kernel, dim3(nblk, B), dim3(256), 0, as_stream(stream), x, x_next, batch_stride, rows,
                       (float*)workspace, F, FP, iters);
    P4C_CHECK_LAUNCH("p4c_nan_moments(partial)");
    const int tot = 5 * B * F;
    hipLaunchKernelGGL(nan_moments_final_kernel, dim3((tot + 127) / 128), dim3(128), 0, as_stream(stream), (const float*)workspace,
                       nblk, out, B, F);
    P4C_CHECK_LAUNCH("p4c_nan_moments(final)");
    return P4C_OK;
}

static int ar_update_loss_fwd_impl(const float* prev, int64_t prev_bs, const void* y, int y_dtype, int y_cs,
                                   const float* target, int64_t tgt_bs, const float* std, const float* mean,
                                   const float* border_mask, const float* interior_mask, float* new_state,
                                   int64_t new_bs, const float* weights, float num_interior,
                                   const int32_t* masked_count, int kind, int mask_mode, float* loss_out,
                                   int64_t loss_stride, void* workspace, int B, int64_t N, int F, float keep_prev,
                                   const NextX* next, p4c_stream_t stream) {
    P4C_CHECK_ARG(y && target && interior_mask && new_state && weights && loss_out && workspace,
                  "p4c_ar_update_loss_fwd: null pointer");
    P4C_CHECK_ARG(prev || keep_prev == 0.0f, "p4c_ar_update_loss_fwd: prev is null but keep_prev != 0");
    P4C_CHECK_ARG((std == nullptr) == (mean == nullptr), "p4c_ar_update_loss_fwd: std and mean go together");
    P4C_CHECK_ARG(mask_mode == P4C_MASK_NONE || mask_mode == P4C_MASK_FROM_NAN,
                  "p4c_ar_update_loss_fwd: only MASK_NONE / MASK_FROM_NAN are fused");
    P4C_CHECK_ARG(F > 0 && F <= 64 * LOSS_MAX_ITERS && y_cs >= F, "p4c_ar_update_loss_fwd: bad F / y_cs");
    if (!force_flat() && (y_dtype == P4C_F32 || y_dtype == P4C_BF16) && F % 4 == 0 && F <= 64 && y_cs % 4 == 0 && prev_bs % 4 == 0 && tgt_bs % 4 == 0 && new_bs % 4 == 0 &&
        aligned16(prev) && aligned16(y) && aligned16(target) && aligned16(new_state) && aligned16(weights) && aligned16(std) &&
        aligned16(mean) &&
        (!next || !next->x || (next->c_pad % 4 == 0 && next->Fs % 4 == 0 && next->c_pad / 4 - F / 4 <= pow2_ge64(F / 4))) &&
        (!next || !next->lgrad || (next->lgrad_bs % 4 == 0 && aligned16(next->lgrad)))) {
        const int FP4 = pow2_ge64(F / 4);
        const int nblk4 = loss_blocks(N, 64 / FP4, B);
        NextX nx{};
        if (next) nx = *next;
#define P4C_LAUNCH_V4(TY, NEXTF)                                                                                              \
    hipLaunchKernelGGL((ar_update_loss_fwd_v4_kernel<TY, NEXTF>), dim3(nblk4, B), dim3(256), 0, as_stream(stream), prev, prev_bs, \
                       (const TY*)y, y_cs, target, tgt_bs, std, mean, border_mask, interior_mask, new_state, new_bs, weights,  \
                       kind, mask_mode, (float*)workspace, N, F, keep_prev, FP4, nx)
        if (y_dtype == P4C_F32) {
            if (next) P4C_LAUNCH_V4(float, true); else P4C_LAUNCH_V4(float, false);
        } else {
            if (next) P4C_LAUNCH_V4(bf16, true); else P4C_LAUNCH_V4(bf16, false);
        }
#undef P4C_LAUNCH_V4
        P4C_CHECK_LAUNCH("p4c_ar_update_loss_fwd(v4)");
        hipLaunchKernelGGL(weighted_loss_final_kernel, dim3(B), dim3(64), 0, as_stream(stream), (const float*)workspace,
                           nblk4, num_interior, masked_count, loss_out, loss_stride, B);
        P4C_CHECK_LAUNCH("p4c_ar_update_loss_fwd(final)");
        return P4C_OK;
    }
    // ---- any other feature count (the shipped Titan configuration has F = 21): the flat kernels
    {
        const int esz = y_dtype == P4C_BF16 ? 2 : 4;
        bool ok = (y_dtype == P4C_F32 || y_dtype == P4C_BF16) && mask_mode == P4C_MASK_NONE && flat_ok(F, N, y_cs, esz) && prev_bs % 4 == 0 &&
                  tgt_bs % 4 == 0 && new_bs % 4 == 0 && aligned16(prev) && aligned16(y) && aligned16(target) && aligned16(new_state);
        if (ok && next && next->x)
            ok = (next->c_pad * esz) % 16 == 0 && next->c_pad <= 256 && next->c_pad >= F + next->Fs + next->Ff && aligned16(next->x);
        if (ok && next && next->lgrad) ok = next->lgrad_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(next->lgrad) & 7) == 0;
        if (ok) {
            NextX nx{};
            if (next) nx = *next;
            const int64_t ntiles = (N + FLAT_P - 1) / FLAT_P;
            int nblk = loss_blocks(N, FLAT_P / 4, B);
            if (nblk > ntiles) nblk = (int)ntiles;
            const size_t smem = (size_t)FLAT_P * y_cs * esz + (nx.x ? (size_t)FLAT_P * nx.c_pad * esz : 0) + 5 * 64 * sizeof(float);
            if (y_dtype == P4C_F32) {
                P4C_TRY(ensure_dyn_smem((const void*)ar_update_loss_fwd_flat_kernel<float>, (int)smem));
                hipLaunchKernelGGL(ar_update_loss_fwd_flat_kernel<float>, dim3(nblk, B), dim3(256), smem, as_stream(stream), prev, prev_bs,
                                   (const float*)y, y_cs, target, tgt_bs, std, mean, border_mask, interior_mask, new_state, new_bs, weights,
                                   kind, (float*)workspace, N, F, keep_prev, nx, FlatFront{});
            } else {
                P4C_TRY(ensure_dyn_smem((const void*)ar_update_loss_fwd_flat_kernel<bf16>, (int)smem));
                hipLaunchKernelGGL(ar_update_loss_fwd_flat_kernel<bf16>, dim3(nblk, B), dim3(256), smem, as_stream(stream), prev, prev_bs,
                                   (const bf16*)y, y_cs, target, tgt_bs, std, mean, border_mask, interior_mask, new_state, new_bs, weights,
                                   kind, (float*)workspace, N, F, keep_prev, nx, FlatFront{});
            }
            P4C_CHECK_LAUNCH("p4c_ar_update_loss_fwd(flat)");
            hipLaunchKernelGGL(weighted_loss_final_kernel, dim3(B), dim3(64), 0, as_stream(stream), (const float*)workspace, nblk,
                               num_interior, masked_count, loss_out, loss_stride, B);
            P4C_CHECK_LAUNCH("p4c_ar_update_loss_fwd(final)");
            return P4C_OK;
        }
    }
    if (next) return fail(P4C_ERR_UNSUPPORTED, "p4c_ar_update_loss_fwd_next: needs the 16-byte path (F %% 4 == 0, F <= 64, aligned rows) or the "
                                               "flat path (F <= 64, N * F %% 4 == 0, whole 16-byte slots per row, aligned rows, no NaN masks)");
    const int FP = pow2_ge64(F), iters = (F + FP - 1) / FP;
    const int nblk = loss_blocks(N, 64 / FP, B);
#define P4C_LAUNCH_FWD(TY)                                                                                              \
    hipLaunchKernelGGL(ar_update_loss_fwd_kernel<TY>, dim3(nblk, B), dim3(256), 0, as_stream(stream), prev, prev_bs,      \
                       (const TY*)y, y_cs, target, tgt_bs, std, mean, border_mask, interior_mask, new_state, new_bs,    \
                       weights, kind, mask_mode, (float*)workspace, N, F, keep_prev, FP, iters)
    if (y_dtype == P4C_F32)
        P4C_LAUNCH_FWD(float);
    else if (y_dtype == P4C_BF16)
        P4C_LAUNCH_FWD(bf16);
    else
        return fail(P4C_ERR_INVALID, "p4c_ar_update_loss_fwd: bad dtype %d", y_dtype);
#undef P4C_LAUNCH_FWD
    P4C_CHECK_LAUNCH("p4c_ar_update_loss_fwd");
    hipLaunchKernelGGL(weighted_loss_final_kernel, dim3(B), dim3(64), 0, as_stream(stream),
                       (const float*)workspace, nblk, num_interior, masked_count, loss_out, loss_stride, B);
    P4C_CHECK_LAUNCH("p4c_ar_update_loss_fwd(final)");
    return P4C_OK;
}

extern "C" int p4c_ar_update_loss_fwd(const float* prev, int64_t prev_bs, const void* y, int y_dtype, int y_cs,
                                      const float* target, int64_t tgt_bs, const float* std, const float* mean,
                                      const float* border_mask, const float* interior_mask, float* new_state,
                                      int64_t new_bs, const float* weights, float num_interior,
                                      const int32_t* masked_count, int kind, int mask_mode, float* loss_out,
                                      int64_t loss_stride, void* workspace, int B, int64_t N, int F, float keep_prev,
                                      p4c_stream_t stream) {
    return ar_update_loss_fwd_impl(prev, prev_bs, y, y_dtype, y_cs, target, tgt_bs, std, mean, border_mask, interior_mask,
                                   new_state, new_bs, weights, num_interior, masked_count, kind, mask_mode, loss_out,
                                   loss_stride, workspace, B, N, F, keep_prev, nullptr, stream);
}

extern "C" int p4c_ar_update_loss_fwd_next(const float* prev, int64_t prev_bs, const void* y, int y_dtype, int y_cs,
                                           const float* target, int64_t tgt_bs, const float* std, const float* mean,
                                           const float* border_mask, const float* interior_mask, float* new_state,
                                           int64_t new_bs, const float* weights, float num_interior,
                                           const int32_t* masked_count, int kind, int mask_mode, float* loss_out,
                                           int64_t loss_stride, void* workspace, int B, int64_t N, int F, float keep_prev,
                                           void* x_next, int c_pad, const float* statics, int64_t statics_bs, int Fs,
                                           const float* forcing_next, int64_t forcing_bs, int Ff, p4c_stream_t stream) {
    P4C_CHECK_ARG(x_next && statics && forcing_next, "p4c_ar_update_loss_fwd_next: null pointer");
    P4C_CHECK_ARG(c_pad >= F + Fs + Ff && Fs >= 0 && Ff >= 0, "p4c_ar_update_loss_fwd_next: c_pad must be >= F + Fs + Ff");
    P4C_CHECK_ARG(mask_mode == P4C_MASK_NONE, "p4c_ar_update_loss_fwd_next: the NaN-mask input channel is built by p4c_build_x");
    NextX nx{x_next, c_pad, statics, statics_bs, Fs, forcing_next, forcing_bs, Ff, nullptr, 0};
    return ar_update_loss_fwd_impl(prev, prev_bs, y, y_dtype, y_cs, target, tgt_bs, std, mean, border_mask, interior_mask,
                                   new_state, new_bs, weights, num_interior, masked_count, kind, mask_mode, loss_out,
                                   loss_stride, workspace, B, N, F, keep_prev, &nx, stream);
}

// The same fused step that ALSO saves d loss_elem / d pred of every element as bf16 rows (lgrad: (N, F) per sample, batch stride
// lgrad_bs elements) for p4c_ar_update_loss_bwd_saved.  x_next may be NULL (the last AR step: nothing follows).
extern "C" int p4c_ar_update_loss_fwd_next_saved(const float* prev, int64_t prev_bs, const void* y, int y_dtype, int y_cs,
                                                 const float* target, int64_t tgt_bs, const float* std, const float* mean,
                                                 const float* border_mask, const float* interior_mask, float* new_state,
                                                 int64_t new_bs, const float* weights, float num_interior,
                                                 const int32_t* masked_count, int kind, int mask_mode, float* loss_out,
                                                 int64_t loss_stride, void* workspace, int B, int64_t N, int F, float keep_prev,
                                                 void* x_next, int c_pad, const float* statics, int64_t statics_bs, int Fs,
                                                 const float* forcing_next, int64_t forcing_bs, int Ff, void* lgrad, int64_t lgrad_bs,
                                                 p4c_stream_t stream) {
    P4C_CHECK_ARG(lgrad && lgrad_bs % 4 == 0, "p4c_ar_update_loss_fwd_next_saved: lgrad rows must be 8-byte aligned");
    if (x_next) {
        P4C_CHECK_ARG(statics && forcing_next, "p4c_ar_update_loss_fwd_next_saved: null pointer");
        P4C_CHECK_ARG(c_pad >= F + Fs + Ff && Fs >= 0 && Ff >= 0, "p4c_ar_update_loss_fwd_next_saved: c_pad must be >= F + Fs + Ff");
        P4C_CHECK_ARG(mask_mode == P4C_MASK_NONE, "p4c_ar_update_loss_fwd_next_saved: the NaN-mask input channel is built by p4c_build_x");
    }
    NextX nx{x_next, c_pad, statics, statics_bs, Fs, forcing_next, forcing_bs, Ff, lgrad, lgrad_bs};
    return ar_update_loss_fwd_impl(prev, prev_bs, y, y_dtype, y_cs, target, tgt_bs, std, mean, border_mask, interior_mask,
                                   new_state, new_bs, weights, num_interior, masked_count, kind, mask_mode, loss_out,
                                   loss_stride, workspace, B, N, F, keep_prev, &nx, stream);
}

// ------------------------------------------------------------------ output convolution + state update + loss in ONE pass
// The network's last layer -- mfai's HalfUNet ends in a 1x1 convolution on relu(norm(a)) (py4cast/lightning.py:591-596) -- and the rest
// of the AR step (residual / scaled-residual update, border forcing, weighted loss, next step's network input, saved loss gradient:
// lightning.py:599-633, losses.py:130-169, lightning.py:711-767) as one kernel: the convolution's output y (B,N,64) is never written
// to memory and read back (134 MB each way per AR step at 2 x 512 x 512) and its launch disappears.  This is the north star's "fused
// normalise-residual-loss epilogue" of the model's last convolution.
//   * y is rounded to bf16 exactly where the two-kernel route stores it, and the update is the SAME loop body as
//     ar_update_loss_fwd_v4_kernel (the reference's op order): new state, next input and saved loss gradients are the two-kernel
//     route's bits; the loss is the same sum in another order.
// 16-byte path conditions as p4c_ar_update_loss_fwd_next; no NaN masks (mask_mode NONE).
struct OutConvArgs {
    const bf16* a; const float* a_scale; const float* a_shift; const float* wout; int cout;
    const float* prev; int64_t prev_bs; const float* target; int64_t tgt_bs; const float* std; const float* mean;
    const float* border_mask; const float* interior_mask; float* new_state; int64_t new_bs; const float* weights;
    int kind; float* partial; int64_t N; int F; float keep_prev;
    NextX nx;
};

// Layout of the work: a wave takes 32 consecutive grid points at a time.
//   front end: y^T = W relu(norm(a))^T on the matrix cores -- A = W (M = output feature; bf16 fragments in LDS, laid once per workgroup),
//     B = the activations (N = grid point: a lane's fragment is 16 bytes of its point's row, straight from HBM, normalised in
//     registers with the sample's scale / shift from LDS) -- the accumulator then holds, per lane, 4 consecutive features of its
//     grid point per register quad: rounded to bf16 they go to a per-wave LDS tile [32 points][64 features] as 8-byte pieces;
//   back end: the loop body of ar_update_loss_fwd_v4_kernel, unchanged -- a lane owns 4 consecutive features of a grid point, 16-byte
//     accesses to prev / target / new state / next input / saved loss gradients -- with y read from the LDS tile instead of HBM.
__global__ void __launch_bounds__(256) out_conv_update_loss_fwd_kernel(OutConvArgs g, int FP4) {
    __shared__ float red[4];
    __shared__ bf16x8_t wimg[2 * 4 * 64];                   // A fragments: (T, ks, lane) = W[32 T + (l & 31)][16 ks + 8 (l >> 5) .. + 7]
    __shared__ float lsc[64], lsh[64];
    __shared__ __attribute__((aligned(16))) bf16 ytile[4][32 * 64];   // per wave: y of its 32 grid points, bf16, [point][feature]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int64_t N = g.N;
    const int F = g.F;
    for (int t = threadIdx.x; t < 2 * 4 * 64; t += 256) {
        const int l = t & 63, ks = (t >> 6) & 3, T = t >> 8;
        const int co = 32 * T + (l & 31), k0 = 16 * ks + 8 * (l >> 5);
        v4f lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (co < g.cout) {
            lo = *reinterpret_cast<const v4f*>(g.wout + (int64_t)co * 64 + k0);
            hi = *reinterpret_cast<const v4f*>(g.wout + (int64_t)co * 64 + k0 + 4);
        }
        bf16x8_t w8;
#pragma unroll
        for (int j = 0; j < 4; ++j) { w8[j] = (__bf16)lo[j]; w8[4 + j] = (__bf16)hi[j]; }
        wimg[t] = w8;
    }
    if (threadIdx.x < 64) {
        lsc[threadIdx.x] = g.a_scale[b * 64 + threadIdx.x];
        lsh[threadIdx.x] = g.a_shift[b * 64 + threadIdx.x];
    }
    __syncthreads();
    // ---- back end: per-lane constants exactly as in ar_update_loss_fwd_v4_kernel<bf16, true>
    const int PP = 64 / FP4;
    const int pp = lane / FP4, q = lane % FP4;
    const bool act = 4 * q < F;
    v4f w = {0, 0, 0, 0}, sd = {1, 1, 1, 1}, mn = {0, 0, 0, 0};
    if (act) {
        w = *reinterpret_cast<const v4f*>(g.weights + 4 * q);
        if (g.std) {
            sd = *reinterpret_cast<const v4f*>(g.std + 4 * q);
            mn = *reinterpret_cast<const v4f*>(g.mean + 4 * q);
        }
    }
    const NextX& nx = g.nx;
    int tkind = 4;  // 0: 16 bytes from one source, 1: last 1..3 forcing values, 2: zeros, 4: none
    int tvalid = 0;
    const float* tsrc = nullptr;
    int64_t tstride = 0;
    bf16* xn = nullptr;
    bf16* lgr = nullptr;
    if (nx.lgrad) lgr = reinterpret_cast<bf16*>(nx.lgrad) + (int64_t)b * nx.lgrad_bs;
    if (nx.x) {
        xn = reinterpret_cast<bf16*>(nx.x) + (int64_t)b * N * nx.c_pad;
        const int c0 = F + 4 * q, o_forc = F + nx.Fs, c_in = F + nx.Fs + nx.Ff;
        if (c0 < nx.c_pad) {
            if (c0 >= c_in) tkind = 2;
            else if (c0 + 3 < o_forc) { tkind = 0; tsrc = nx.statics + (int64_t)b * nx.statics_bs + (c0 - F); tstride = nx.Fs; }
            else {
                tsrc = nx.forcing + (int64_t)b * nx.forcing_bs + (c0 - o_forc);
                tstride = nx.Ff;
                tvalid = c_in - c0 < 4 ? c_in - c0 : 4;
                tkind = tvalid == 4 ? 0 : 1;
            }
        }
    }
    const bf16* ab = g.a + (int64_t)b * N * 64;
    bf16* yt = ytile[wv];
    float acc = 0.0f;
    const int64_t ntiles = (N + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t n0 = tile * 32;
        // ---- front end  (measured SLOWER, profiles/r04_step_ab_runs.txt: prefetching the next tile's rows of `a` during the back end
        // -- 5.00 vs 4.87 ms per step --, and staging the rows through the LDS tile with whole-row loads -- 5.10 vs 4.95)
        {
            f32x16_t y0, y1;
#pragma unroll
            for (int i = 0; i < 16; ++i) y0[i] = y1[i] = 0.f;
            const int64_t np = n0 + r;
            u32x4_t cur[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                cur[ks] = u32x4_t{0u, 0u, 0u, 0u};
                if (np < N) cur[ks] = *reinterpret_cast<const u32x4_t*>(ab + np * 64 + 16 * ks + 8 * h);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const v4f s0 = *reinterpret_cast<const v4f*>(lsc + 16 * ks + 8 * h), s1 = *reinterpret_cast<const v4f*>(lsc + 16 * ks + 8 * h + 4);
                const v4f t0 = *reinterpret_cast<const v4f*>(lsh + 16 * ks + 8 * h), t1 = *reinterpret_cast<const v4f*>(lsh + 16 * ks + 8 * h + 4);
                const float scv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
                const float shv[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
                u32x4_t o;
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    // relu(a * scale + shift), fp32, one rounding to bf16: what the 1x1 convolution's loader stages (conv_rows.hip: xform2<2>)
                    const unsigned int wd = cur[ks][k2];
                    const float lo = __builtin_fmaf(__builtin_bit_cast(float, wd << 16), scv[2 * k2], shv[2 * k2]);
                    const float hi = __builtin_fmaf(__builtin_bit_cast(float, wd & 0xffff0000u), scv[2 * k2 + 1], shv[2 * k2 + 1]);
                    const s16x2_t z = {0, 0};
                    o[k2] = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, pack_bf16(lo, hi)), z));
                }
                const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, o);
                y0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wimg[(0 * 4 + ks) * 64 + lane], bf, y0, 0, 0, 0);
                y1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wimg[(1 * 4 + ks) * 64 + lane], bf, y1, 0, 0, 0);
            }
            // C[feature][point]: lane = point r (+ half h), register quad gq -> features 32 T + 8 gq + 4 h .. + 3
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x16_t& yy = T == 0 ? y0 : y1;
                    u32x2_t o;
                    o[0] = pack_bf16(yy[4 * gq], yy[4 * gq + 1]);
                    o[1] = pack_bf16(yy[4 * gq + 2], yy[4 * gq + 3]);
                    // (8-byte slot 8 T + 2 gq + h of the point's row, XOR-ed with the point index: the 16 lanes of a store group share
                    // the slot index and would hit the same two banks otherwise)
                    *reinterpret_cast<u32x2_t*>(yt + r * 64 + 4 * ((8 * T + 2 * gq + h) ^ (r & 15))) = o;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the tile is this wave's own: no workgroup barrier)
        // ---- back end: ar_update_loss_fwd_v4_kernel's loop body on the tile's grid points, y from LDS
        for (int it = 0; it < 32; it += PP) {
            const int pl = it + pp;
            const int64_t n = n0 + pl;
            if (n >= N) continue;
            if (xn && tkind != 4) {
                v4f tq = {0, 0, 0, 0};
                if (tkind == 0) tq = *reinterpret_cast<const v4f_a4*>(tsrc + n * tstride);
                else if (tkind == 1) {
                    const float* pf = tsrc + n * tstride;
                    tq[0] = pf[0];
                    if (tvalid > 1) tq[1] = pf[1];
                    if (tvalid > 2) tq[2] = pf[2];
                }
                store4f(xn + n * nx.c_pad + F + 4 * q, tq);
            }
            if (!act) continue;
            const float im = g.interior_mask[n];
            const float bm = g.border_mask ? g.border_mask[n] : 0.0f;
            const int64_t e = n * F + 4 * q;
            const v4f yv = load4f(yt + pl * 64 + 4 * (q ^ (pl & 15)));
            v4f pv = {0, 0, 0, 0};
            if (g.prev) pv = *reinterpret_cast<const v4f*>(g.prev + (int64_t)b * g.prev_bs + e);
            const v4f tg = *reinterpret_cast<const v4f*>(g.target + (int64_t)b * g.tgt_bs + e);
            v4f o, lg;
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p0 = pv[j], t0 = tg[j];
                float pr;
                if (g.std) {
                    pr = p0 * g.keep_prev + yv[j] * sd[j];
                    pr = pr + mn[j];
                } else {
                    pr = p0 * g.keep_prev + yv[j];
                }
                if (g.border_mask) pr = bm * t0 + im * pr;
                o[j] = pr;
                s += loss_elem(pr, t0, 1.0f, g.kind) * w[j];
                lg[j] = loss_elem_grad(pr, t0, 1.0f, g.kind);
            }
            *reinterpret_cast<v4f*>(g.new_state + (int64_t)b * g.new_bs + e) = o;
            if (xn) store4f(xn + n * nx.c_pad + 4 * q, o);
            if (lgr) store4f(lgr + e, lg);
            acc += s * im;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the tile's reads are done before the next tile overwrites it)
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) g.partial[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int p4c_out_conv_update_loss_fwd(const void* a, const float* a_scale, const float* a_shift, const float* wout, int cout,
                                            const float* prev, int64_t prev_bs, const float* target, int64_t tgt_bs, const float* std,
                                            const float* mean, const float* border_mask, const float* interior_mask, float* new_state,
                                            int64_t new_bs, const float* weights, float num_interior, const int32_t* masked_count,
                                            int kind, float* loss_out, int64_t loss_stride, void* workspace, int B, int64_t N, int F,
                                            float keep_prev, void* x_next, int c_pad, const float* statics, int64_t statics_bs, int Fs,
                                            const float* forcing_next, int64_t forcing_bs, int Ff, void* lgrad, int64_t lgrad_bs,
                                            p4c_stream_t stream) {
    P4C_CHECK_ARG(a && a_scale && a_shift && wout && target && interior_mask && new_state && weights && loss_out && workspace,
                  "p4c_out_conv_update_loss_fwd: null pointer");
    P4C_CHECK_ARG(prev || keep_prev == 0.0f, "p4c_out_conv_update_loss_fwd: prev is null but keep_prev != 0");
    P4C_CHECK_ARG((std == nullptr) == (mean == nullptr), "p4c_out_conv_update_loss_fwd: std and mean go together");
    P4C_CHECK_ARG(cout >= F && cout <= 64 && F > 0, "p4c_out_conv_update_loss_fwd: needs 0 < F <= cout <= 64");
    P4C_CHECK_ARG(aligned16(a) && aligned16(wout), "p4c_out_conv_update_loss_fwd: a and wout must be 16-byte aligned");
    // The flat AR-step kernel with the convolution as its front end serves ANY feature count -- and where both forms apply it is the
    // faster one (2 x 512 x 512 x 60: 127 against 138 us per launch, profiles/r04_step_ab_runs.txt block 21; same bits) -- so the
    // 16-byte form below is what is left when the flat path's conditions fail.  P4C_TAIL_V4=1: prefer the 16-byte form (A/B).
    bool flat_possible = flat_ok(F, N, 64, 2) && prev_bs % 4 == 0 && tgt_bs % 4 == 0 && new_bs % 4 == 0 && aligned16(prev) && aligned16(target) &&
                         aligned16(new_state);
    if (flat_possible && x_next) flat_possible = statics && forcing_next && (c_pad * 2) % 16 == 0 && c_pad <= 256 && c_pad >= F + Fs + Ff && aligned16(x_next);
    if (flat_possible && lgrad) flat_possible = lgrad_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(lgrad) & 7) == 0;
    const char* tv4 = diag_env("P4C_TAIL_V4");
    const bool prefer_v4 = tv4 && tv4[0] == '1' && !force_flat();
    if (F % 4 != 0 || force_flat() || (flat_possible && !prefer_v4)) {
        const bool ok = flat_possible;
        if (!ok && F % 4 != 0)
            return fail(P4C_ERR_UNSUPPORTED, "p4c_out_conv_update_loss_fwd: F %% 4 != 0 needs the flat path (F <= 64, N * F %% 4 == 0, whole 16-byte "
                                             "slots per row of x_next, aligned rows)");
        if (ok) {
        const NextX nx{x_next, c_pad, statics, statics_bs, Fs, forcing_next, forcing_bs, Ff, lgrad, lgrad_bs};
        const FlatFront fa{(const bf16*)a, a_scale, a_shift, wout, cout};
        const int64_t ntiles = (N + FLAT_P - 1) / FLAT_P;
        int nblk = loss_blocks(N, FLAT_P / 4, B);
        if (nblk > ntiles) nblk = (int)ntiles;
        const size_t smem = (size_t)FLAT_P * 64 * 2 + (x_next ? (size_t)FLAT_P * c_pad * 2 : 0) + 5 * 64 * sizeof(float) + 2 * 4 * 64 * 16 +
                            2 * 64 * sizeof(float);
        P4C_TRY(ensure_dyn_smem((const void*)ar_update_loss_fwd_flat_kernel<bf16, true>, (int)smem));
        hipLaunchKernelGGL((ar_update_loss_fwd_flat_kernel<bf16, true>), dim3(nblk, B), dim3(256), smem, as_stream(stream), prev, prev_bs,
                           (const bf16*)nullptr, 64, target, tgt_bs, std, mean, border_mask, interior_mask, new_state, new_bs, weights, kind,
                           (float*)workspace, N, F, keep_prev, nx, fa);
        P4C_CHECK_LAUNCH("p4c_out_conv_update_loss_fwd(flat)");
        hipLaunchKernelGGL(weighted_loss_final_kernel, dim3(B), dim3(64), 0, as_stream(stream), (const float*)workspace, nblk, num_interior,
                           masked_count, loss_out, loss_stride, B);
        P4C_CHECK_LAUNCH("p4c_out_conv_update_loss_fwd(final)");
        return P4C_OK;
        }
    }
    P4C_CHECK_ARG(prev_bs % 4 == 0 && tgt_bs % 4 == 0 && new_bs % 4 == 0 && aligned16(prev) && aligned16(target) && aligned16(new_state) &&
                      aligned16(weights) && aligned16(std) && aligned16(mean),
                  "p4c_out_conv_update_loss_fwd: rows must be 16-byte aligned (as p4c_ar_update_loss_fwd_next)");
    if (x_next) {
        P4C_CHECK_ARG(statics && forcing_next, "p4c_out_conv_update_loss_fwd: null pointer");
        P4C_CHECK_ARG(c_pad % 4 == 0 && c_pad >= F + Fs + Ff && Fs % 4 == 0 && Fs >= 0 && Ff >= 0,
                      "p4c_out_conv_update_loss_fwd: c_pad / Fs must be multiples of 4 and c_pad >= F + Fs + Ff");
        P4C_CHECK_ARG(c_pad / 4 - F / 4 <= pow2_ge64(F / 4), "p4c_out_conv_update_loss_fwd: too many tail channels for one pass");
    }
    P4C_CHECK_ARG(!lgrad || (lgrad_bs % 4 == 0 && aligned16(lgrad)), "p4c_out_conv_update_loss_fwd: lgrad must be 16-byte aligned rows");
    const int FP4 = pow2_ge64(F / 4);
    const int nblk = loss_blocks(N, 32, B);
    OutConvArgs g{(const bf16*)a, a_scale, a_shift, wout, cout, prev, prev_bs, target, tgt_bs, std, mean, border_mask, interior_mask,
                  new_state, new_bs, weights, kind, (float*)workspace, N, F, keep_prev,
                  NextX{x_next, c_pad, statics, statics_bs, Fs, forcing_next, forcing_bs, Ff, lgrad, lgrad_bs}};
    hipLaunchKernelGGL(out_conv_update_loss_fwd_kernel, dim3(nblk, B), dim3(256), 0, as_stream(stream), g, FP4);
    P4C_CHECK_LAUNCH("p4c_out_conv_update_loss_fwd");
    hipLaunchKernelGGL(weighted_loss_final_kernel, dim3(B), dim3(64), 0, as_stream(stream), (const float*)workspace, nblk, num_interior,
                       masked_count, loss_out, loss_stride, B);
    P4C_CHECK_LAUNCH("p4c_out_conv_update_loss_fwd(final)");
    return P4C_OK;
}

// Backward of the fused step from the saved loss gradients (bf16 rows written by p4c_ar_update_loss_fwd_next_saved) instead of the
// new state and the target: 120 instead of 480 bytes read per grid point at F = 60.  16-byte path only (F % 4 == 0, aligned rows).
extern "C" int p4c_ar_update_loss_bwd_saved(const float* g_next, int64_t g_next_bs, const void* g_next2, int g2_dtype, int g2_cs,
                                            const float* gloss, int64_t gloss_stride, const void* lgrad, int64_t lgrad_bs,
                                            const float* std, const float* interior_mask, int force_border, const float* weights,
                                            float num_interior, const int32_t* masked_count, int kind, int mask_mode, void* dy,
                                            int dy_dtype, int y_cs, float* dprev, int64_t dprev_bs, int B, int64_t N, int F,
                                            float keep_prev, p4c_stream_t stream) {
    P4C_CHECK_ARG(lgrad && interior_mask && weights && dy, "p4c_ar_update_loss_bwd_saved: null pointer");
    P4C_CHECK_ARG(F > 0 && y_cs >= F && (dy_dtype == P4C_F32 || dy_dtype == P4C_BF16), "p4c_ar_update_loss_bwd_saved: bad F / y_cs / dtype");
    P4C_CHECK_ARG(dy_dtype == g2_dtype || !g_next2, "p4c_ar_update_loss_bwd_saved: g_next2 dtype must equal dy dtype");
    const bool ok = !force_flat() && F % 4 == 0 && y_cs % 4 == 0 && y_cs / 4 <= 64 && g_next_bs % 4 == 0 && lgrad_bs % 4 == 0 && dprev_bs % 4 == 0 &&
                    g2_cs % 4 == 0 && aligned16(g_next) && aligned16(g_next2) && aligned16(lgrad) && aligned16(dy) && aligned16(dprev) &&
                    aligned16(weights) && aligned16(std);
    if (!ok) {   // any other feature count: the flat kernel
        const int esz = dy_dtype == P4C_BF16 ? 2 : 4;
        const bool fok = mask_mode == P4C_MASK_NONE && flat_ok(F, N, y_cs, esz) && (!g_next2 || flat_ok(F, N, g2_cs, esz)) && g_next_bs % 4 == 0 &&
                         lgrad_bs % 4 == 0 && dprev_bs % 4 == 0 && aligned16(g_next) && aligned16(g_next2) &&
                         (reinterpret_cast<uintptr_t>(lgrad) & 7) == 0 && aligned16(dy) && aligned16(dprev);
        if (!fok)
            return fail(P4C_ERR_UNSUPPORTED, "p4c_ar_update_loss_bwd_saved: needs the 16-byte path (F %% 4 == 0, aligned rows) or the flat path "
                                             "(F <= 64, N * F %% 4 == 0, whole 16-byte slots per row, aligned rows, no NaN masks)");
        const int64_t ntiles = (N + FLAT_P - 1) / FLAT_P;
        int nblk = loss_blocks(N, FLAT_P / 4, B);
        if (nblk > ntiles) nblk = (int)ntiles;
        const size_t smem = (size_t)FLAT_P * y_cs * esz + (g_next2 ? (size_t)FLAT_P * g2_cs * esz : 0) + 3 * 64 * sizeof(float);
        if (dy_dtype == P4C_F32) {
            P4C_TRY(ensure_dyn_smem((const void*)ar_update_loss_bwd_flat_kernel<float, true>, (int)smem));
            hipLaunchKernelGGL((ar_update_loss_bwd_flat_kernel<float, true>), dim3(nblk, B), dim3(256), smem, as_stream(stream), g_next, g_next_bs,
                               (const float*)g_next2, g2_cs, gloss, gloss_stride, (const float*)lgrad, lgrad_bs, nullptr, 0, std, interior_mask,
                               force_border, weights, num_interior, masked_count, kind, (float*)dy, y_cs, dprev, dprev_bs, N, F, keep_prev);
        } else {
            P4C_TRY(ensure_dyn_smem((const void*)ar_update_loss_bwd_flat_kernel<bf16, true>, (int)smem));
            hipLaunchKernelGGL((ar_update_loss_bwd_flat_kernel<bf16, true>), dim3(nblk, B), dim3(256), smem, as_stream(stream), g_next, g_next_bs,
                               (const bf16*)g_next2, g2_cs, gloss, gloss_stride, (const float*)lgrad, lgrad_bs, nullptr, 0, std, interior_mask,
                               force_border, weights, num_interior, masked_count, kind, (bf16*)dy, y_cs, dprev, dprev_bs, N, F, keep_prev);
        }
        P4C_CHECK_LAUNCH("p4c_ar_update_loss_bwd_saved(flat)");
        return P4C_OK;
    }
    const int FP4 = pow2_ge64(y_cs / 4);
    const int nblk4 = loss_blocks(N, 64 / FP4, B);
    if (dy_dtype == P4C_F32)
        hipLaunchKernelGGL((ar_update_loss_bwd_v4_kernel<float, true>), dim3(nblk4, B), dim3(256), 0, as_stream(stream), g_next, g_next_bs,
                           (const float*)g_next2, g2_cs, gloss, gloss_stride, (const float*)lgrad, lgrad_bs, nullptr, 0, std, interior_mask,
                           force_border, weights, num_interior, masked_count, kind, mask_mode, (float*)dy, y_cs, dprev, dprev_bs, N, F,
                           keep_prev, FP4);
    else
        hipLaunchKernelGGL((ar_update_loss_bwd_v4_kernel<bf16, true>), dim3(nblk4, B), dim3(256), 0, as_stream(stream), g_next, g_next_bs,
                           (const bf16*)g_next2, g2_cs, gloss, gloss_stride, (const float*)lgrad, lgrad_bs, nullptr, 0, std, interior_mask,
                           force_border, weights, num_interior, masked_count, kind, mask_mode, (bf16*)dy, y_cs, dprev, dprev_bs, N, F,
                           keep_prev, FP4);
    P4C_CHECK_LAUNCH("p4c_ar_update_loss_bwd_saved");
    return P4C_OK;
}

extern "C" int p4c_ar_update_loss_bwd(const float* g_next, int64_t g_next_bs, const void* g_next2, int g2_dtype,
                                      int g2_cs, const float* gloss, int64_t gloss_stride, const float* new_state,
                                      int64_t new_bs, const float* target, int64_t tgt_bs, const float* std,
                                      const float* interior_mask, int force_border, const float* weights,
                                      float num_interior, const int32_t* masked_count, int kind, int mask_mode,
                                      void* dy, int dy_dtype, int y_cs, float* dprev, int64_t dprev_bs, int B,
                                      int64_t N, int F, float keep_prev, p4c_stream_t stream) {
    P4C_CHECK_ARG(new_state && target && interior_mask && weights && dy, "p4c_ar_update_loss_bwd: null pointer");
    P4C_CHECK_ARG(mask_mode == P4C_MASK_NONE || mask_mode == P4C_MASK_FROM_NAN,
                  "p4c_ar_update_loss_bwd: only MASK_NONE / MASK_FROM_NAN are fused");
    P4C_CHECK_ARG(F > 0 && F <= 64 * LOSS_MAX_ITERS && y_cs >= F, "p4c_ar_update_loss_bwd: bad F / y_cs");
    P4C_CHECK_ARG(dy_dtype == g2_dtype || !g_next2, "p4c_ar_update_loss_bwd: g_next2 dtype must equal dy dtype");
    if (!force_flat() && (dy_dtype == P4C_F32 || dy_dtype == P4C_BF16) && F % 4 == 0 && y_cs % 4 == 0 && y_cs <= 256 && g_next_bs % 4 == 0 && new_bs % 4 == 0 &&
        tgt_bs % 4 == 0 && dprev_bs % 4 == 0 && g2_cs % 4 == 0 && aligned16(g_next) && aligned16(g_next2) &&
        aligned16(new_state) && aligned16(target) && aligned16(dy) && aligned16(dprev) && aligned16(weights) &&
        aligned16(std)) {
        const int FP4 = pow2_ge64(y_cs / 4);
        if (y_cs / 4 <= 64) {
            const int nblk4 = loss_blocks(N, 64 / FP4, B);
            if (dy_dtype == P4C_F32)
                hipLaunchKernelGGL(ar_update_loss_bwd_v4_kernel<float>, dim3(nblk4, B), dim3(256), 0, as_stream(stream), g_next,
                                   g_next_bs, (const float*)g_next2, g2_cs, gloss, gloss_stride, new_state, new_bs, target,
                                   tgt_bs, std, interior_mask, force_border, weights, num_interior, masked_count, kind,
                                   mask_mode, (float*)dy, y_cs, dprev, dprev_bs, N, F, keep_prev, FP4);
            else
                hipLaunchKernelGGL(ar_update_loss_bwd_v4_kernel<bf16>, dim3(nblk4, B), dim3(256), 0, as_stream(stream), g_next,
                                   g_next_bs, (const bf16*)g_next2, g2_cs, gloss, gloss_stride, new_state, new_bs, target,
                                   tgt_bs, std, interior_mask, force_border, weights, num_interior, masked_count, kind,
                                   mask_mode, (bf16*)dy, y_cs, dprev, dprev_bs, N, F, keep_prev, FP4);
            P4C_CHECK_LAUNCH("p4c_ar_update_loss_bwd(v4)");
            return P4C_OK;
        }
    }
    {   // any other feature count: the flat kernel (see ar_update_loss_bwd_flat_kernel)
        const int esz = dy_dtype == P4C_BF16 ? 2 : 4;
        if ((dy_dtype == P4C_F32 || dy_dtype == P4C_BF16) && mask_mode == P4C_MASK_NONE && flat_ok(F, N, y_cs, esz) &&
            (!g_next2 || flat_ok(F, N, g2_cs, esz)) && g_next_bs % 4 == 0 && new_bs % 4 == 0 && tgt_bs % 4 == 0 && dprev_bs % 4 == 0 &&
            aligned16(g_next) && aligned16(g_next2) && aligned16(new_state) && aligned16(target) && aligned16(dy) && aligned16(dprev)) {
            const int64_t ntiles = (N + FLAT_P - 1) / FLAT_P;
            int nblk = loss_blocks(N, FLAT_P / 4, B);
            if (nblk > ntiles) nblk = (int)ntiles;
            const size_t smem = (size_t)FLAT_P * y_cs * esz + (g_next2 ? (size_t)FLAT_P * g2_cs * esz : 0) + 3 * 64 * sizeof(float);
            if (dy_dtype == P4C_F32) {
                P4C_TRY(ensure_dyn_smem((const void*)ar_update_loss_bwd_flat_kernel<float, false>, (int)smem));
                hipLaunchKernelGGL((ar_update_loss_bwd_flat_kernel<float, false>), dim3(nblk, B), dim3(256), smem, as_stream(stream), g_next,
                                   g_next_bs, (const float*)g_next2, g2_cs, gloss, gloss_stride, new_state, new_bs, target, tgt_bs, std,
                                   interior_mask, force_border, weights, num_interior, masked_count, kind, (float*)dy, y_cs, dprev, dprev_bs, N,
                                   F, keep_prev);
            } else {
                P4C_TRY(ensure_dyn_smem((const void*)ar_update_loss_bwd_flat_kernel<bf16, false>, (int)smem));
                hipLaunchKernelGGL((ar_update_loss_bwd_flat_kernel<bf16, false>), dim3(nblk, B), dim3(256), smem, as_stream(stream), g_next,
                                   g_next_bs, (const bf16*)g_next2, g2_cs, gloss, gloss_stride, new_state, new_bs, target, tgt_bs, std,
                                   interior_mask, force_border, weights, num_interior, masked_count, kind, (bf16*)dy, y_cs, dprev, dprev_bs, N,
                                   F, keep_prev);
            }
            P4C_CHECK_LAUNCH("p4c_ar_update_loss_bwd(flat)");
            return P4C_OK;
        }
    }
    const int FP = pow2_ge64(y_cs), iters = (y_cs + FP - 1) / FP;
    P4C_CHECK_ARG(iters <= LOSS_MAX_ITERS, "p4c_ar_update_loss_bwd: y_cs too large");
    const int nblk = loss_blocks(N, 64 / FP, B);
#define P4C_LAUNCH_BWD(TY)                                                                                              \
    hipLaunchKernelGGL((ar_update_loss_bwd_kernel<TY, TY>), dim3(nblk, B), dim3(256), 0, as_stream(stream), g_next,       \
                       g_next_bs, (const TY*)g_next2, g2_cs, gloss, gloss_stride, new_state, new_bs, target, tgt_bs,    \
                       std, interior_mask, force_border, weights, num_interior, masked_count, kind, mask_mode, (TY*)dy, \
                       y_cs, dprev, dprev_bs, N, F, keep_prev, FP, iters)
    if (dy_dtype == P4C_F32)
        P4C_LAUNCH_BWD(float);
    else if (dy_dtype == P4C_BF16)
        P4C_LAUNCH_BWD(bf16);
    else
        return fail(P4C_ERR_INVALID, "p4c_ar_update_loss_bwd: bad dtype %d", dy_dtype);
#undef P4C_LAUNCH_BWD
    P4C_CHECK_LAUNCH("p4c_ar_update_loss_bwd");
    return P4C_OK;
}
