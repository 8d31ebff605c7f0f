// K1 build_x and K2 ar_update (+ backward): the non-model part of the AR rollout.
// Reference: py4cast/lightning.py:495-676 (_common_step), 711-767 (_next_x).
//
// All kernels here are HBM-bound streaming passes over (B, N, features) rows.  The feature
// dimension is the contiguous one in the reference's NamedTensor layout, so a wave walks
// grid points with its lanes spread over the channels of one (or several) grid points:
// loads/stores are contiguous per grid point and no index division happens per element.
//
// This translation unit is compiled with -ffp-contract=off: K2 evaluates the reference's
// expression in the reference's order, one rounding per op, so fp32 results are bit-identical
// to the torch op chain.
#include <math.h>

#include "common.hpp"

namespace p4c {

// lanes are split into 64/FP segments of FP lanes; segment = one grid point, lane%FP = channel
__host__ __device__ static inline int pow2_ge(int v) {
    int p = 1;
    while (p < v && p < 64) p <<= 1;
    return p;
}

constexpr int K1_MAX_ITERS = 8;  // channels handled per lane: c_pad <= 64*K1_MAX_ITERS

// mask_tensor (lightning.py:769-785) as addressing: grid point (yy, xx) lies in a cleared block iff flat index
// (yy / block_h) * W + (xx / block_w) was drawn -- `selected` is that H*W byte table (1 = drawn).
struct BlockMask {
    const uint8_t* selected;
    int W, block_h, block_w;
    __device__ bool hit(int64_t n) const {
        const int yy = (int)(n / W), xx = (int)(n - (int64_t)yy * W);
        return selected[(int64_t)(yy / block_h) * W + (xx / block_w)] != 0;
    }
};

template <typename TX>
__global__ void __launch_bounds__(256) build_x_kernel(const float* __restrict__ prev, int64_t prev_bs, int64_t prev_ts,
                                                      const float* __restrict__ statics, int64_t statics_bs,
                                                      const float* __restrict__ forcing, int64_t forcing_bs,
                                                      TX* __restrict__ x, int c_pad, int B, int T_in, int64_t N,
                                                      int F, int Fs, int Ff, int mask_on_nan, int n_prev_ch,
                                                      int FP, int iters, BlockMask bm) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int PP = 64 / FP;
    const int pp = lane / FP;
    const int c0 = lane % FP;
    const int o_stat = n_prev_ch, o_forc = n_prev_ch + Fs, o_mask = n_prev_ch + Fs + Ff;
    const int c_in = o_mask + (mask_on_nan ? 1 : 0);
    const int64_t total = (int64_t)B * N;
    const unsigned long long segmask = (FP == 64) ? ~0ull : (((1ull << FP) - 1ull) << (pp * FP));

    for (int64_t base = (int64_t)wave * PP; base < total; base += (int64_t)nwaves * PP) {
        const int64_t pix = base + pp;
        const bool live = pix < total;
        const int b = live ? (int)(pix / N) : 0;
        const int64_t n = live ? (pix - (int64_t)b * N) : 0;
        float vals[K1_MAX_ITERS];
        bool any_nan = false;
#pragma unroll
        for (int it = 0; it < K1_MAX_ITERS; ++it) {
            if (it >= iters) break;
            const int c = c0 + it * FP;
            float v = 0.0f;
            bool counts = false;  // channel takes part in the NaN union (inputs + forcing only)
            if (live && c < c_pad) {
                if (c < o_stat) {
                    const int t = c / F, f = c - t * F;
                    v = prev[(int64_t)b * prev_bs + (int64_t)t * prev_ts + n * F + f];
                    counts = true;
                } else if (c < o_forc) {
                    v = statics[(int64_t)b * statics_bs + n * Fs + (c - o_stat)];
                } else if (c < o_mask) {
                    v = forcing[(int64_t)b * forcing_bs + n * Ff + (c - o_forc)];
                    counts = true;
                }
            }
            vals[it] = v;
            if (mask_on_nan) {
                const bool isn = counts && (v != v);
                const unsigned long long bal = __ballot(isn);
                any_nan = any_nan || ((bal & segmask) != 0ull);
                if (counts && isn) vals[it] = 0.0f;  // nan_to_num on inputs and forcing (lightning.py:755-757)
            }
        }
        if (!live) continue;
        TX* xrow = x + pix * (int64_t)c_pad;
#pragma unroll
        for (int it = 0; it < K1_MAX_ITERS; ++it) {
            if (it >= iters) break;
            const int c = c0 + it * FP;
            if (c < c_pad) {
                float v = vals[it];
                if (mask_on_nan && c == c_in - 1) v = any_nan ? 0.0f : 1.0f;  // ~combined_mask (lightning.py:750-752)
                // masked-auto-encoder blocks (lightning.py:769-785: `x * mask`, mask False on the drawn blocks): the product
                // with 0.0 is taken literally (sign of zero, NaN propagation), bit for bit what torch's x * False gives
                if (bm.selected && c < c_in && bm.hit(n)) v = v * 0.0f;
                xrow[c] = from_f32<TX>(v);
            }
        }
    }
}

// 16-byte vectorised build_x (fp32 output, no NaN masking, c_pad % 4 == 0): a lane produces one float4 of the
// output row; quads lying wholly inside the previous-state or statics block are one 16-byte load, the rest
// (forcing rows are 5 floats = unaligned) are gathered element by element.  Stores are always 16 bytes.
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));  // 4-byte aligned 16-byte access (forcing rows)

template <typename TX, bool ODD = false>
__global__ void __launch_bounds__(256)
    build_x_v4_kernel(const float* __restrict__ prev, int64_t prev_bs, int64_t prev_ts, const float* __restrict__ statics,
                      int64_t statics_bs, const float* __restrict__ forcing, int64_t forcing_bs, TX* __restrict__ x,
                      int c_pad, int B, int T_in, int64_t N, int F, int Fs, int Ff, int n_prev_ch, int FP4, int vec_prev,
                      int vec_stat) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int b = blockIdx.y;  // one sample per grid row: no 64-bit division per grid point
    const int PP = 64 / FP4;
    const int pp = lane / FP4, q = lane % FP4;
    const int c0 = 4 * q;
    const int o_stat = n_prev_ch, o_forc = n_prev_ch + Fs, c_in = n_prev_ch + Fs + Ff;
    const int64_t total = N;
    if (c0 >= c_pad) return;
    // classify the quad once (it is the same for every grid point): every lane ends up with ONE source row pointer and
    // row stride, so the whole wave runs the same load instruction (no per-source branches in the loop):
    //   kind 0: 4 values from one source tensor (state / statics / forcing; forcing rows are 4-byte aligned only, which
    //           a global dwordx4 load accepts);  kind 1: the last 1..3 values of the forcing row (scalar loads);
    //   kind 2: zero padding;  kind 3: a quad straddling two sources (generic per-element gather)
    int kind = 3, nvalid = 0;
    const float* src = nullptr;
    int64_t rstride = 0;
    if (c0 >= c_in) {
        kind = 2;
    } else if (c0 + 3 < o_stat) {
        const int t_idx = c0 / F, f_idx = c0 - t_idx * F;
        if (vec_prev && f_idx + 3 < F) { kind = 0; src = prev + (int64_t)b * prev_bs + (int64_t)t_idx * prev_ts + f_idx; rstride = F; }
    } else if (c0 >= o_stat && c0 + 3 < o_forc) {
        if (vec_stat) { kind = 0; src = statics + (int64_t)b * statics_bs + (c0 - o_stat); rstride = Fs; }
    } else if (c0 >= o_forc) {
        src = forcing + (int64_t)b * forcing_bs + (c0 - o_forc);
        rstride = Ff;
        nvalid = c_in - c0 < 4 ? c_in - c0 : 4;
        kind = nvalid == 4 ? 0 : 1;
    }
    // 4 grid points per thread and trip: the 4 loads are independent and all issued before the first store
    // (memory-level parallelism; one load in flight per lane left this pass latency-bound)
    constexpr int UNR = 4;
    for (int64_t base = (int64_t)wave * PP + pp; base < total; base += (int64_t)nwaves * PP * UNR) {
        v4f v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t n = base + (int64_t)u * nwaves * PP;
            v[u] = v4f{0.f, 0.f, 0.f, 0.f};
            if (n >= total) continue;
            if (kind == 0) {
                v[u] = *reinterpret_cast<const v4f_a4*>(src + n * rstride);
            } else if (kind == 1) {
                const float* p = src + n * rstride;
                v[u][0] = p[0];
                if (nvalid > 1) v[u][1] = p[1];
                if (nvalid > 2) v[u][2] = p[2];
            } else if (kind == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + j;
                    float e = 0.f;
                    if (c < o_stat) {
                        const int t = c / F, f = c - t * F;
                        e = prev[(int64_t)b * prev_bs + (int64_t)t * prev_ts + n * F + f];
                    } else if (c < o_forc) {
                        e = statics[(int64_t)b * statics_bs + n * Fs + (c - o_stat)];
                    } else if (c < c_in) {
                        e = forcing[(int64_t)b * forcing_bs + n * Ff + (c - o_forc)];
                    }
                    v[u][j] = e;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t n = base + (int64_t)u * nwaves * PP;
            if (n < total) {
                TX* xr = x + ((int64_t)b * N + n) * (int64_t)c_pad + c0;
                if constexpr (ODD) {   // fp32 rows of c_pad % 4 != 0 channels: 4-byte aligned 16-byte stores + a scalar tail
                    if (c0 + 3 < c_pad) *reinterpret_cast<v4f_a4*>(xr) = v[u];
                    else
                        for (int j = 0; j < c_pad - c0; ++j) xr[j] = v[u][j];
                } else {
                    store4f(xr, v[u]);
                }
            }
        }
    }
}

template <typename TX>
__global__ void __launch_bounds__(256) build_x_bwd_kernel(const TX* __restrict__ dx, int c_pad, float* __restrict__ dprev,
                                                          int B, int T_in, int64_t N, int F, int FP, int iters,
                                                          BlockMask bm) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const int nch = T_in * F;
    const int64_t total = (int64_t)B * N;
    for (int64_t base = (int64_t)wave * PP; base < total; base += (int64_t)nwaves * PP) {
        const int64_t pix = base + pp;
        if (pix >= total) continue;
        const int b = (int)(pix / N);
        const int64_t n = pix - (int64_t)b * N;
        for (int it = 0; it < iters; ++it) {
            const int c = c0 + it * FP;
            if (c < nch) {
                const int t = c / F, f = c - t * F;
                float g = to_f32<TX>(dx[pix * (int64_t)c_pad + c]);
                if (bm.selected && bm.hit(n)) g = g * 0.0f;   // adjoint of x * mask
                dprev[(((int64_t)b * T_in + t) * N + n) * F + f] = g;
            }
        }
    }
}

// ---------------------------------------------------------------------------- K2
template <typename TY>
__global__ void __launch_bounds__(256)
    ar_update_fwd_kernel(const float* __restrict__ prev, int64_t prev_bs, const TY* __restrict__ y, int y_cs,
                         const float* __restrict__ border_state, int64_t border_bs, const float* __restrict__ std,
                         const float* __restrict__ mean, const float* __restrict__ border_mask,
                         const float* __restrict__ interior_mask, float* __restrict__ new_state, int64_t new_bs, int B,
                         int64_t N, int F, float keep_prev, int nan_to_num, int FP, int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const int64_t total = (int64_t)B * N;
    for (int64_t base = (int64_t)wave * PP; base < total; base += (int64_t)nwaves * PP) {
        const int64_t pix = base + pp;
        if (pix >= total) continue;
        const int b = (int)(pix / N);
        const int64_t n = pix - (int64_t)b * N;
        float bm = 0.f, im = 1.f;
        if (border_mask) {
            bm = border_mask[n];
            im = interior_mask[n];
        }
        for (int it = 0; it < iters; ++it) {
            const int f = c0 + it * FP;
            if (f >= F) continue;
            float yv = to_f32<TY>(y[pix * (int64_t)y_cs + f]);
            float p;
            float pv = 0.f;
            if (prev) {
                pv = prev[(int64_t)b * prev_bs + n * F + f];
                if (nan_to_num) pv = nan_to_zero(pv);
            }
            // reference order (lightning.py:605-610 / 623): ((prev*(1-ds)) + (y*std)) + mean
            if (std) {
                p = pv * keep_prev + yv * std[f];
                p = p + mean[f];
            } else {
                p = pv * keep_prev + yv;
            }
            if (border_mask) {  // lightning.py:628-631
                float bs = border_state[(int64_t)b * border_bs + n * F + f];
                if (nan_to_num) bs = nan_to_zero(bs);
                p = bm * bs + im * p;
            }
            new_state[(int64_t)b * new_bs + n * F + f] = p;
        }
    }
}

template <typename TY>
__global__ void __launch_bounds__(256)
    ar_update_bwd_kernel(const float* __restrict__ dnew, int64_t dnew_bs, const float* __restrict__ std,
                         const float* __restrict__ interior_mask, TY* __restrict__ dy, int y_cs,
                         float* __restrict__ dprev, int64_t dprev_bs, int B, int64_t N, int F, float keep_prev, int FP,
                         int iters) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int PP = 64 / FP;
    const int pp = lane / FP, c0 = lane % FP;
    const int64_t total = (int64_t)B * N;
    for (int64_t base = (int64_t)wave * PP; base < total; base += (int64_t)nwaves * PP) {
        const int64_t pix = base + pp;
        if (pix >= total) continue;
        const int b = (int)(pix / N);
        const int64_t n = pix - (int64_t)b * N;
        const float im = interior_mask ? interior_mask[n] : 1.0f;
        for (int it = 0; it < iters; ++it) {
            const int f = c0 + it * FP;
            if (f >= y_cs) continue;
            float gy = 0.f;
            if (f < F) {
                const float g = dnew[(int64_t)b * dnew_bs + n * F + f] * im;
                gy = std ? g * std[f] : g;
                if (dprev) dprev[(int64_t)b * dprev_bs + n * F + f] = g * keep_prev;
            }
            dy[pix * (int64_t)y_cs + f] = from_f32<TY>(gy);
        }
    }
}

// build_x for feature counts off the 16-byte grid (round 4; the shipped Titan configuration: 21 features, 4 statics, 21 forcings):
// the (N, F) / (N, Fs) / (N, Ff) fp32 sources are streamed FLAT, 16 bytes per lane -- a lane's 4 elements may straddle two grid points;
// (point, channel) of an element by an exact multiply-shift division -- into an LDS tile of 64 rows [c_pad] that goes out as whole rows
// in 16-byte slots (the scheme of the flat AR-step kernels, losses.hip).  The quad kernel above gathers such rows element by element
// (122 us at 2 x 512 x 640 x 46 -> 64: 1.6 TB/s).  Same values, bit for bit (a conversion per element, no arithmetic).
constexpr int BX_P = 64;   // grid points per tile
__device__ __forceinline__ void bx_pf(int e, int F, unsigned rcp, int& pl, int& f) {   // e < 4096, F <= 64
    pl = (int)(((unsigned)e * rcp) >> 20);
    f = e - pl * F;
}
typedef unsigned int bx_u32x4 __attribute__((ext_vector_type(4)));

template <typename TX>
__global__ void __launch_bounds__(256)
    build_x_flat_kernel(const float* __restrict__ prev, int64_t prev_bs, int64_t prev_ts, const float* __restrict__ statics,
                        int64_t statics_bs, const float* __restrict__ forcing, int64_t forcing_bs, TX* __restrict__ x, int c_pad,
                        int T_in, int64_t N, int F, int Fs, int Ff, int n_prev_ch) {
    extern __shared__ __attribute__((aligned(16))) char bsm[];
    TX* xtile = reinterpret_cast<TX*>(bsm);
    const int tid = threadIdx.x, b = blockIdx.y;
    const int c_in = n_prev_ch + Fs + Ff;
    // the zero padding: laid once, never written again
    for (int i = tid; i < BX_P * (c_pad - c_in); i += 256) {
        const int w = c_pad - c_in, pl = i / w;
        xtile[pl * c_pad + c_in + (i - pl * w)] = from_f32<TX>(0.f);
    }
    const unsigned rcp_p = n_prev_ch ? ((1u << 20) + F - 1) / F : 0u;
    const unsigned rcp_s = Fs ? ((1u << 20) + Fs - 1) / Fs : 0u, rcp_f = Ff ? ((1u << 20) + Ff - 1) / Ff : 0u;
    const int xslots = c_pad * (int)sizeof(TX) / 16;
    const int64_t ntiles = (N + BX_P - 1) / BX_P;
    auto stream_in = [&](const float* src, int C, unsigned rcp, int c_off, int np) __attribute__((always_inline)) {
        const int nel = np * C;   // a multiple of 4 (host)
        for (int e0 = 4 * tid; e0 < nel; e0 += 4 * 256) {
            const v4f v = *reinterpret_cast<const v4f*>(src + e0);
            int pl, f;
            bx_pf(e0, C, rcp, pl, f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xtile[pl * c_pad + c_off + f] = from_f32<TX>(v[j]);
                if (++f == C) { f = 0; ++pl; }
            }
        }
    };
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t n0 = t * BX_P;
        const int np = (int)((N - n0) < BX_P ? (N - n0) : BX_P);
        __syncthreads();   // (the previous tile's rows are out)
        for (int ti = 0; ti * F < n_prev_ch; ++ti)
            stream_in(prev + (int64_t)b * prev_bs + (int64_t)ti * prev_ts + n0 * F, F, rcp_p, ti * F, np);
        if (Fs) stream_in(statics + (int64_t)b * statics_bs + n0 * Fs, Fs, rcp_s, n_prev_ch, np);
        if (Ff) stream_in(forcing + (int64_t)b * forcing_bs + n0 * Ff, Ff, rcp_f, n_prev_ch + Fs, np);
        __syncthreads();
        for (int sl = tid; sl < np * xslots; sl += 256)
            reinterpret_cast<bx_u32x4*>(x + ((int64_t)b * N + n0) * c_pad)[sl] = reinterpret_cast<const bx_u32x4*>(xtile)[sl];
    }
}

static inline int stream_grid(int64_t total_pixels, int PP) {
    // memory-bound: cap at ~8 blocks of 256 threads per CU and grid-stride the rest
    int64_t waves = (total_pixels + PP - 1) / PP;
    int64_t blocks = (waves + 3) / 4;
    int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

}  // namespace p4c

using namespace p4c;

static int build_x_impl(const float* prev, int64_t prev_bs, int64_t prev_ts, const float* statics,
                        int64_t statics_bs, const float* forcing, int64_t forcing_bs, void* x, int x_dtype,
                        int c_pad, int B, int T_in, int64_t N, int F, int Fs, int Ff, int mask_on_nan,
                        int downscaling_only, BlockMask bm, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && statics && forcing, "p4c_build_x: null pointer");
    P4C_CHECK_ARG(B > 0 && N > 0 && F > 0 && Fs >= 0 && Ff >= 0 && T_in >= 0, "p4c_build_x: bad dims");
    const int n_prev_ch = downscaling_only ? 0 : T_in * F;
    P4C_CHECK_ARG(n_prev_ch == 0 || prev, "p4c_build_x: prev is null");
    const int c_in = n_prev_ch + Fs + Ff + (mask_on_nan ? 1 : 0);
    P4C_CHECK_ARG(c_pad >= c_in, "p4c_build_x: c_pad (%d) < C_in (%d)", c_pad, c_in);
    const int FP = pow2_ge(c_pad);
    const int iters = (c_pad + FP - 1) / FP;
    P4C_CHECK_ARG(iters <= K1_MAX_ITERS, "p4c_build_x: c_pad %d too large (max %d)", c_pad, 64 * K1_MAX_ITERS);
    const bool odd = c_pad % 4 != 0;   // the exact C_in of a generic model (e.g. 69): fp32 rows, unaligned vector stores
    // flat streams through an LDS tile of rows wherever its alignment conditions hold: any feature count, and faster than the quad
    // kernel where that one applies too (2 x 512 x 512, 69 -> 96 channels: 45 against 71 us)  (P4C_NO_FLAT_STEP=1: the quad kernel below)
    {
        const int esz = x_dtype == P4C_BF16 ? 2 : 4;
        const char* nf = diag_env("P4C_NO_FLAT_STEP");
        auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        const bool flat = !(nf && nf[0] == '1') && !mask_on_nan && !bm.selected && (x_dtype == P4C_F32 || x_dtype == P4C_BF16) && c_pad <= 256 &&
                          (c_pad * esz) % 16 == 0 && al16(x) && F <= 64 && Fs <= 64 && Ff <= 64 &&
                          (n_prev_ch > 0 ? ((N * F) % 4 == 0 && prev_bs % 4 == 0 && prev_ts % 4 == 0 && al16(prev)) : true) &&
                          (Fs == 0 || ((N * Fs) % 4 == 0 && statics_bs % 4 == 0 && al16(statics))) &&
                          (Ff == 0 || ((N * Ff) % 4 == 0 && forcing_bs % 4 == 0 && al16(forcing)));
        if (flat) {
            const int64_t ntiles = (N + BX_P - 1) / BX_P;
            int64_t blocks = (int64_t)num_cus() * 8 / (B > 0 ? B : 1);
            if (blocks > ntiles) blocks = ntiles;
            if (blocks < 1) blocks = 1;
            const size_t smem = (size_t)BX_P * c_pad * esz;
            if (x_dtype == P4C_F32) {
                P4C_TRY(ensure_dyn_smem((const void*)build_x_flat_kernel<float>, (int)smem));
                hipLaunchKernelGGL(build_x_flat_kernel<float>, dim3((unsigned)blocks, B), dim3(256), smem, as_stream(stream), prev, prev_bs, prev_ts,
                                   statics, statics_bs, forcing, forcing_bs, (float*)x, c_pad, T_in, N, F, Fs, Ff, n_prev_ch);
            } else {
                P4C_TRY(ensure_dyn_smem((const void*)build_x_flat_kernel<bf16>, (int)smem));
                hipLaunchKernelGGL(build_x_flat_kernel<bf16>, dim3((unsigned)blocks, B), dim3(256), smem, as_stream(stream), prev, prev_bs, prev_ts,
                                   statics, statics_bs, forcing, forcing_bs, (bf16*)x, c_pad, T_in, N, F, Fs, Ff, n_prev_ch);
            }
            P4C_CHECK_LAUNCH("p4c_build_x(flat)");
            return P4C_OK;
        }
    }
    if (!mask_on_nan && !bm.selected && c_pad <= 256 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
        ((x_dtype == P4C_F32) || (x_dtype == P4C_BF16 && !odd))) {
        const int FP4 = pow2_ge((c_pad + 3) / 4);
        const int vec_prev = n_prev_ch > 0 && F % 4 == 0 && prev_bs % 4 == 0 && prev_ts % 4 == 0 &&
                             (reinterpret_cast<uintptr_t>(prev) & 15) == 0;
        const int vec_stat = Fs % 4 == 0 && n_prev_ch % 4 == 0 && statics_bs % 4 == 0 &&
                             (reinterpret_cast<uintptr_t>(statics) & 15) == 0;
        const int grid4 = stream_grid(N, 64 / FP4);
        if (x_dtype == P4C_F32 && odd)
            hipLaunchKernelGGL((build_x_v4_kernel<float, true>), dim3(grid4, B), dim3(256), 0, as_stream(stream), prev, prev_bs,
                               prev_ts, statics, statics_bs, forcing, forcing_bs, (float*)x, c_pad, B, T_in, N, F, Fs, Ff,
                               n_prev_ch, FP4, vec_prev, vec_stat);
        else if (x_dtype == P4C_F32)
            hipLaunchKernelGGL(build_x_v4_kernel<float>, dim3(grid4, B), dim3(256), 0, as_stream(stream), prev, prev_bs, prev_ts,
                               statics, statics_bs, forcing, forcing_bs, (float*)x, c_pad, B, T_in, N, F, Fs, Ff, n_prev_ch, FP4,
                               vec_prev, vec_stat);
        else
            hipLaunchKernelGGL(build_x_v4_kernel<bf16>, dim3(grid4, B), dim3(256), 0, as_stream(stream), prev, prev_bs, prev_ts,
                               statics, statics_bs, forcing, forcing_bs, (bf16*)x, c_pad, B, T_in, N, F, Fs, Ff, n_prev_ch, FP4,
                               vec_prev, vec_stat);
        P4C_CHECK_LAUNCH("p4c_build_x(v4)");
        return P4C_OK;
    }
    const int grid = stream_grid((int64_t)B * N, 64 / FP);
    if (x_dtype == P4C_F32)
        hipLaunchKernelGGL(build_x_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), prev, prev_bs, prev_ts,
                           statics, statics_bs, forcing, forcing_bs, (float*)x, c_pad, B, T_in, N, F, Fs, Ff,
                           mask_on_nan, n_prev_ch, FP, iters, bm);
    else if (x_dtype == P4C_BF16)
        hipLaunchKernelGGL(build_x_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), prev, prev_bs, prev_ts,
                           statics, statics_bs, forcing, forcing_bs, (bf16*)x, c_pad, B, T_in, N, F, Fs, Ff,
                           mask_on_nan, n_prev_ch, FP, iters, bm);
    else
        return fail(P4C_ERR_INVALID, "p4c_build_x: bad dtype %d", x_dtype);
    P4C_CHECK_LAUNCH("p4c_build_x");
    return P4C_OK;
}

extern "C" int p4c_build_x(const float* prev, int64_t prev_bs, int64_t prev_ts, const float* statics,
                           int64_t statics_bs, const float* forcing, int64_t forcing_bs, void* x, int x_dtype,
                           int c_pad, int B, int T_in, int64_t N, int F, int Fs, int Ff, int mask_on_nan,
                           int downscaling_only, p4c_stream_t stream) {
    return build_x_impl(prev, prev_bs, prev_ts, statics, statics_bs, forcing, forcing_bs, x, x_dtype, c_pad, B, T_in, N, F, Fs,
                        Ff, mask_on_nan, downscaling_only, BlockMask{nullptr, 1, 1, 1}, stream);
}

static int check_block_mask(const uint8_t* selected, int H, int W, int block_h, int block_w, int64_t N, const char* who) {
    P4C_CHECK_ARG(selected, "%s: block table is null", who);
    P4C_CHECK_ARG(H > 0 && W > 0 && (int64_t)H * W == N, "%s: H*W (%d*%d) != N (%lld)", who, H, W, (long long)N);
    P4C_CHECK_ARG(block_h > 0 && block_w > 0, "%s: block size must be positive", who);
    return P4C_OK;
}

extern "C" int p4c_build_x_masked(const float* prev, int64_t prev_bs, int64_t prev_ts, const float* statics,
                                  int64_t statics_bs, const float* forcing, int64_t forcing_bs, void* x, int x_dtype,
                                  int c_pad, int B, int T_in, int64_t N, int F, int Fs, int Ff, int mask_on_nan,
                                  int downscaling_only, const uint8_t* block_selected, int H, int W, int block_h,
                                  int block_w, p4c_stream_t stream) {
    if (int rc = check_block_mask(block_selected, H, W, block_h, block_w, N, "p4c_build_x_masked")) return rc;
    return build_x_impl(prev, prev_bs, prev_ts, statics, statics_bs, forcing, forcing_bs, x, x_dtype, c_pad, B, T_in, N, F, Fs,
                        Ff, mask_on_nan, downscaling_only, BlockMask{block_selected, W, block_h, block_w}, stream);
}

// ---------------------------------------------------------------------------- rows next to the path
// un-normalise a prediction per feature: out = x*std[f] + mean[f], evaluated as the reference's two in-place passes
// (`*= std` then `+= mean`, lightning.py:1162-1169): two roundings, no FMA (this file is built with -ffp-contract=off).
__global__ void __launch_bounds__(256) unnormalize_kernel(const float* __restrict__ x, const float* __restrict__ std,
                                                          const float* __restrict__ mean, float* __restrict__ out,
                                                          int64_t total, int F) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int f = (int)(i % F);
        const float v = x[i] * std[f];
        out[i] = v + mean[f];
    }
}

// standardise + pack: per-parameter planes raw[f][r] (what a loader reading one .npy per parameter produces, r over
// batch x timestep x grid) -> features-last rows out[r][f] = (raw[f][r] - mean[f]) / std[f]
// (datasets/base.py:448-452 followed by NamedTensor.concat + collate_fn :173-195).  A block transposes 256 rows x all
// features through LDS: every plane read is a contiguous 1 KB run and the block's output (256*F floats) is one
// contiguous run written in order.
constexpr int PACK_ROWS = 256;
__global__ void __launch_bounds__(256) pack_standardize_kernel(const float* __restrict__ raw, int64_t plane_stride,
                                                               const float* __restrict__ mean, const float* __restrict__ std,
                                                               float* __restrict__ out, int64_t R, int F) {
    extern __shared__ float tile[];  // [F][PACK_ROWS + 1]
    const int64_t r0 = (int64_t)blockIdx.x * PACK_ROWS;
    const int64_t r = r0 + threadIdx.x;
    // 8 planes per trip: the 8 loads of a thread are independent and in flight together
    for (int f0 = 0; f0 < F; f0 += 8) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (f0 + k < F && r < R) ? raw[(int64_t)(f0 + k) * plane_stride + r] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (f0 + k < F) {
                const float d = v[k] - mean[f0 + k];
                tile[(f0 + k) * (PACK_ROWS + 1) + threadIdx.x] = (r < R) ? d / std[f0 + k] : 0.f;
            }
    }
    __syncthreads();
    const int64_t rows_here = (R - r0 < PACK_ROWS) ? (R - r0) : PACK_ROWS;
    const int64_t total = rows_here * F;
    float* dst = out + r0 * F;
    int row = threadIdx.x / F, f = threadIdx.x - row * F;
    const int drow = 256 / F, df = 256 - drow * F;
    for (int64_t j = threadIdx.x; j < total; j += 256) {
        dst[j] = tile[f * (PACK_ROWS + 1) + row];
        row += drow;
        f += df;
        if (f >= F) { f -= F; ++row; }
    }
}

// AdamW over one flat fp32 parameter / gradient buffer (configure_optimizers, lightning.py:442-467 -> torch.optim.AdamW):
// the update order of torch's implementation (decoupled weight decay, lerp of the first moment, mul + addcmul of the
// second, bias corrections folded into step_size and the denominator), one launch instead of ~20 multi-tensor passes.
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float decay, float w1, float beta2,
                                                    float w2, float bc2_sqrt, float eps, float step_size) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float grad = g[i];
        float param = p[i] * decay;                       // param.mul_(1 - lr * weight_decay)
        const float mi = m[i] + w1 * (grad - m[i]);       // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * beta2 + w2 * grad * grad; // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        param = param - step_size * (mi / denom);         // param.addcdiv_(exp_avg, denom, value=-step_size)
        p[i] = param; m[i] = mi; v[i] = vi;
    }
}

extern "C" int p4c_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                              double beta1, double beta2, double eps, double weight_decay, int64_t step,
                              p4c_stream_t stream) {
    P4C_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, "p4c_adamw_step: null pointer");
    P4C_CHECK_ARG(n > 0 && step >= 1, "p4c_adamw_step: n and step must be positive");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    int64_t blocks = (n + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), params, grads, exp_avg, exp_avg_sq, n,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)sqrt(bc2), (float)eps, (float)(lr / bc1));
    P4C_CHECK_LAUNCH("p4c_adamw_step");
    return P4C_OK;
}

extern "C" int p4c_unnormalize(const float* x, const float* std, const float* mean, float* out, int64_t rows, int F,
                               p4c_stream_t stream) {
    P4C_CHECK_ARG(x && std && mean && out, "p4c_unnormalize: null pointer");
    P4C_CHECK_ARG(rows > 0 && F > 0, "p4c_unnormalize: bad dims");
    const int64_t total = rows * F;
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(unnormalize_kernel, dim3((int)blocks), dim3(256), 0, as_stream(stream), x, std, mean, out, total, F);
    P4C_CHECK_LAUNCH("p4c_unnormalize");
    return P4C_OK;
}

// un-normalise + features-last -> one plane per feature, through an LDS tile of 64 grid points x F features: global reads are
// rows of F floats (contiguous), global writes runs of 64 floats per feature (contiguous): both sides coalesced.  The
// arithmetic is the reference's two rounded steps (x * std, then + mean; contraction is off in this translation unit).
constexpr int PLANE_PTS = 64;
__global__ void __launch_bounds__(256)
    unnormalize_planes_kernel(const float* __restrict__ x, const float* __restrict__ std, const float* __restrict__ mean,
                              float* __restrict__ out, int64_t N, int F) {
    extern __shared__ float tile[];   // [PLANE_PTS][F + 1]
    const int64_t img = blockIdx.y;   // (b, t) index
    const int64_t n0 = (int64_t)blockIdx.x * PLANE_PTS;
    const int npts = (int)((N - n0) < PLANE_PTS ? (N - n0) : PLANE_PTS);
    const float* src = x + (img * N + n0) * F;
    for (int i = threadIdx.x; i < npts * F; i += 256) {
        const int pt = i / F, f = i - pt * F;
        float v = src[i] * std[f];
        v = v + mean[f];
        tile[pt * (F + 1) + f] = v;
    }
    __syncthreads();
    float* dst = out + img * F * N + n0;
    for (int i = threadIdx.x; i < F * PLANE_PTS; i += 256) {
        const int f = i / PLANE_PTS, pt = i - f * PLANE_PTS;
        if (pt < npts) dst[(int64_t)f * N + pt] = tile[pt * (F + 1) + f];
    }
}

extern "C" int p4c_unnormalize_planes(const float* x, const float* std, const float* mean, float* out, int64_t images,
                                      int64_t N, int F, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && std && mean && out, "p4c_unnormalize_planes: null pointer");
    P4C_CHECK_ARG(images > 0 && images < 65536 && N > 0 && F > 0 && F <= 512, "p4c_unnormalize_planes: bad dims");
    const size_t smem = (size_t)PLANE_PTS * (F + 1) * sizeof(float);
    P4C_TRY(ensure_dyn_smem((const void*)unnormalize_planes_kernel, PLANE_PTS * 513 * 4));
    hipLaunchKernelGGL(unnormalize_planes_kernel, dim3((unsigned)((N + PLANE_PTS - 1) / PLANE_PTS), (unsigned)images), dim3(256),
                       smem, as_stream(stream), x, std, mean, out, N, F);
    P4C_CHECK_LAUNCH("p4c_unnormalize_planes");
    return P4C_OK;
}

extern "C" int p4c_pack_standardize(const float* raw, int64_t plane_stride, const float* mean, const float* std, float* out,
                                    int64_t rows, int F, p4c_stream_t stream) {
    P4C_CHECK_ARG(raw && mean && std && out, "p4c_pack_standardize: null pointer");
    P4C_CHECK_ARG(rows > 0 && F > 0 && plane_stride >= rows, "p4c_pack_standardize: bad dims");
    P4C_CHECK_ARG((rows + PACK_ROWS - 1) / PACK_ROWS < ((int64_t)1 << 31), "p4c_pack_standardize: too many rows");
    P4C_CHECK_ARG(F <= 144, "p4c_pack_standardize: at most 144 features per call (LDS tile)");
    const size_t smem = (size_t)F * (PACK_ROWS + 1) * sizeof(float);
    P4C_TRY(ensure_dyn_smem((const void*)pack_standardize_kernel, 144 * (PACK_ROWS + 1) * 4));
    hipLaunchKernelGGL(pack_standardize_kernel, dim3((unsigned)((rows + PACK_ROWS - 1) / PACK_ROWS)), dim3(256), smem,
                       as_stream(stream), raw, plane_stride, mean, std, out, rows, F);
    P4C_CHECK_LAUNCH("p4c_pack_standardize");
    return P4C_OK;
}

static int build_x_bwd_impl(const void* dx, int dx_dtype, int c_pad, float* dprev, int B, int T_in, int64_t N,
                            int F, BlockMask bm, p4c_stream_t stream) {
    P4C_CHECK_ARG(dx && dprev, "p4c_build_x_bwd: null pointer");
    P4C_CHECK_ARG(c_pad >= T_in * F, "p4c_build_x_bwd: c_pad < T_in*F");
    const int nch = T_in * F;
    const int FP = pow2_ge(nch);
    const int iters = (nch + FP - 1) / FP;
    const int grid = stream_grid((int64_t)B * N, 64 / FP);
    if (dx_dtype == P4C_F32)
        hipLaunchKernelGGL(build_x_bwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)dx,
                           c_pad, dprev, B, T_in, N, F, FP, iters, bm);
    else if (dx_dtype == P4C_BF16)
        hipLaunchKernelGGL(build_x_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16*)dx,
                           c_pad, dprev, B, T_in, N, F, FP, iters, bm);
    else
        return fail(P4C_ERR_INVALID, "p4c_build_x_bwd: bad dtype %d", dx_dtype);
    P4C_CHECK_LAUNCH("p4c_build_x_bwd");
    return P4C_OK;
}

extern "C" int p4c_build_x_bwd(const void* dx, int dx_dtype, int c_pad, float* dprev, int B, int T_in, int64_t N,
                               int F, p4c_stream_t stream) {
    return build_x_bwd_impl(dx, dx_dtype, c_pad, dprev, B, T_in, N, F, BlockMask{nullptr, 1, 1, 1}, stream);
}

extern "C" int p4c_build_x_bwd_masked(const void* dx, int dx_dtype, int c_pad, float* dprev, int B, int T_in, int64_t N,
                                      int F, const uint8_t* block_selected, int H, int W, int block_h, int block_w,
                                      p4c_stream_t stream) {
    if (int rc = check_block_mask(block_selected, H, W, block_h, block_w, N, "p4c_build_x_bwd_masked")) return rc;
    return build_x_bwd_impl(dx, dx_dtype, c_pad, dprev, B, T_in, N, F, BlockMask{block_selected, W, block_h, block_w}, stream);
}

extern "C" int p4c_ar_update_fwd(const float* prev, int64_t prev_bs, const void* y, int y_dtype, int y_cs,
                                 const float* border_state, int64_t border_bs, const float* std, const float* mean,
                                 const float* border_mask, const float* interior_mask, float* new_state,
                                 int64_t new_bs, int B, int64_t N, int F, float keep_prev, int nan_to_num,
                                 p4c_stream_t stream) {
    P4C_CHECK_ARG(y && new_state, "p4c_ar_update_fwd: null pointer");
    P4C_CHECK_ARG(prev || keep_prev == 0.0f, "p4c_ar_update_fwd: prev is null but keep_prev != 0");
    P4C_CHECK_ARG((std == nullptr) == (mean == nullptr), "p4c_ar_update_fwd: std and mean go together");
    P4C_CHECK_ARG(!border_mask || (interior_mask && border_state), "p4c_ar_update_fwd: border forcing needs masks+state");
    P4C_CHECK_ARG(y_cs >= F && B > 0 && N > 0 && F > 0, "p4c_ar_update_fwd: bad dims");
    const int FP = pow2_ge(F);
    const int iters = (F + FP - 1) / FP;
    const int grid = stream_grid((int64_t)B * N, 64 / FP);
    if (y_dtype == P4C_F32)
        hipLaunchKernelGGL(ar_update_fwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), prev, prev_bs,
                           (const float*)y, y_cs, border_state, border_bs, std, mean, border_mask, interior_mask,
                           new_state, new_bs, B, N, F, keep_prev, nan_to_num, FP, iters);
    else if (y_dtype == P4C_BF16)
        hipLaunchKernelGGL(ar_update_fwd_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), prev, prev_bs,
                           (const bf16*)y, y_cs, border_state, border_bs, std, mean, border_mask, interior_mask,
                           new_state, new_bs, B, N, F, keep_prev, nan_to_num, FP, iters);
    else
        return fail(P4C_ERR_INVALID, "p4c_ar_update_fwd: bad dtype %d", y_dtype);
    P4C_CHECK_LAUNCH("p4c_ar_update_fwd");
    return P4C_OK;
}

extern "C" int p4c_ar_update_bwd(const float* dnew, int64_t dnew_bs, const float* std, const float* interior_mask,
                                 void* dy, int dy_dtype, int y_cs, float* dprev, int64_t dprev_bs, int B, int64_t N,
                                 int F, float keep_prev, p4c_stream_t stream) {
    P4C_CHECK_ARG(dnew && dy, "p4c_ar_update_bwd: null pointer");
    P4C_CHECK_ARG(y_cs >= F && B > 0 && N > 0 && F > 0, "p4c_ar_update_bwd: bad dims");
    const int FP = pow2_ge(y_cs);
    const int iters = (y_cs + FP - 1) / FP;
    const int grid = stream_grid((int64_t)B * N, 64 / FP);
    if (dy_dtype == P4C_F32)
        hipLaunchKernelGGL(ar_update_bwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), dnew, dnew_bs, std,
                           interior_mask, (float*)dy, y_cs, dprev, dprev_bs, B, N, F, keep_prev, FP, iters);
    else if (dy_dtype == P4C_BF16)
        hipLaunchKernelGGL(ar_update_bwd_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), dnew, dnew_bs, std,
                           interior_mask, (bf16*)dy, y_cs, dprev, dprev_bs, B, N, F, keep_prev, FP, iters);
    else
        return fail(P4C_ERR_INVALID, "p4c_ar_update_bwd: bad dtype %d", dy_dtype);
    P4C_CHECK_LAUNCH("p4c_ar_update_bwd");
    return P4C_OK;
}
