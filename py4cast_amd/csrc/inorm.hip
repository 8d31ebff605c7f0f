// InstanceNorm2d(affine) + LeakyReLU (+ residual) on features-last tensors (B, N = H*W, C), C a multiple of 4 up to 1024 -- the
// normalisation of MONAI's UnetResBlock (SwinUNetR decoder, config/CLI/model/swinunetr.yaml:23 `norm_name: instance`; UNETR++'s
// full-resolution blocks).  Statistics are per (sample, channel) over the pixels: two streaming passes each way,
//   forward : sums of x, x^2                        -> (tiny finalize by the caller) -> y = lrelu(x*scale + shift [+ res])
//   backward: sums of dz, dz*xhat (dz = dy*lrelu')  -> (tiny finalize)              -> dx = scale*(dz - m1 - xhat*m2), dres = dz
// HBM-bound; 16 B (fp32) / 8 B (bf16) per lane, channel quads across lanes so that a pixel row is read in whole lines.
#include "common.hpp"

namespace p4c {
namespace inorm {

__device__ __forceinline__ p4c_f32x4 ld4(const float* p) { return *reinterpret_cast<const p4c_f32x4*>(p); }
__device__ __forceinline__ p4c_f32x4 ld4(const bf16* p) { return load4f(p); }

// A multiplier per (row group, channel) behind the activation (round 6): out = lrelu((x * scale + shift + res) * m[p / rows][c] * factor).
// The channel dropout in front of UNETR++'s conv8 (mfai's `Sequential(Dropout2d(0.1), Conv2d)`: m = the Bernoulli draw per (sample,
// channel), factor = 1 / (1 - p)) rides in the batch norm + LeakyReLU passes that produce the convolution's input -- a full-size
// multiplication each way before.  m >= 0, so lrelu(z) m = lrelu(z m) and the saved output still tells the sign of z where m > 0;
// where m = 0 every gradient through the element is 0.  base == nullptr: no multiplier.
struct PostMul {
    const float* base;
    int64_t rows;         // rows of a (batch-as-one-sample) map that share a multiplier row: H * W of the real samples
    float factor;
    __device__ __forceinline__ p4c_f32x4 at(int64_t p, int C, int c) const {
        if (!base) return p4c_f32x4{1.f, 1.f, 1.f, 1.f};
        const p4c_f32x4 v = *reinterpret_cast<const p4c_f32x4*>(base + (p / rows) * C + c);
        return p4c_f32x4{v[0] * factor, v[1] * factor, v[2] * factor, v[3] * factor};
    }
};

// Finalize INSIDE the reduce launch (round 6): the workgroup that draws the last ticket turns the partials of all samples into the
// statistics -- what p4c_inorm_finalize_fwd / _bwd do in a launch of their own (a dependent 5-8 us launch per normalisation each way:
// ~160 per SwinUNETR training step, the finalize launches of UNETR++'s full-resolution blocks and batch norms).  Instance form only
// (one channel per statistics group).  The partials go out as device-scope (write-through) stores, the ticket follows once they are
// acknowledged, the last workgroup invalidates before it reads them (the pattern of the row kernel's BatchFin); it sums them with the
// same slice_sums() and the same per-channel expressions as the finalize kernels: bit-identical statistics.
struct InFin {
    unsigned int* ticket;     // zero between launches (the last workgroup resets it); nullptr: no in-kernel finalize
    const float* gamma;       // forward
    const float* beta;
    float eps;
    float* o0;                // forward: mean, rstd, scale, shift (B, C);  backward: c1, c2 (B, C), dgamma, dbeta (C)
    float* o1;
    float* o2;
    float* o3;
    double n_group;           // pixels per statistics group
    int CB, SL;               // finalize_geometry(C, 1)
};
__device__ void finalize_in_kernel(const InFin& f, bool bwd, const float* partial, int nb, int B, int C, float* red0, float* red1);

// partial[b][blk][0][c] = sum a, [1][c] = sum a*b over the block's pixels, with
//   MODE 0 (forward):  a = x,  b = x
//   MODE 1 (backward): a = dz = dy * (y > 0 ? 1 : slope),  b = xhat = (x - mean) * rstd
template <typename T, int MODE>
__global__ void __launch_bounds__(256) reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, float slope,
                                                     float* __restrict__ partial, int64_t N, int C, InFin fin, PostMul pm) {
    extern __shared__ float red[];              // [rows][2][C]  (>= 2048 floats: the in-kernel finalize reuses it)
    const int b = blockIdx.y, cqn = C >> 2;
    const int rows = 256 / cqn > 0 ? 256 / cqn : 1;
    const T* xb = x + (int64_t)b * N * C;
    const T* dyb = MODE ? dy + (int64_t)b * N * C : nullptr;
    const T* yb = MODE ? y + (int64_t)b * N * C : nullptr;
    for (int q0 = 0; q0 < cqn; q0 += 256) {     // (C > 1024 never happens; one trip)
        const int q = q0 + (int)(threadIdx.x % (cqn < 256 ? cqn : 256)), pl = threadIdx.x / (cqn < 256 ? cqn : 256);
        p4c_f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, mu = s1, rs = s1;
        const bool live = pl < rows && q < cqn;
        if (live && MODE) { mu = ld4(mean + b * C + 4 * q); rs = ld4(rstd + b * C + 4 * q); }
        if (live)
            for (int64_t p = (int64_t)blockIdx.x * rows + pl; p < N; p += (int64_t)gridDim.x * rows) {
                const p4c_f32x4 xv = ld4(xb + p * C + 4 * q);
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { s1[k] += xv[k]; s2[k] = __builtin_fmaf(xv[k], xv[k], s2[k]); }
                } else {
                    const p4c_f32x4 g = ld4(dyb + p * C + 4 * q), yv = ld4(yb + p * C + 4 * q);
                    const p4c_f32x4 pv = pm.at(p, C, 4 * q);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dz = (yv[k] > 0.f ? g[k] : g[k] * slope) * pv[k];
                        s1[k] += dz;
                        s2[k] = __builtin_fmaf(dz, (xv[k] - mu[k]) * rs[k], s2[k]);
                    }
                }
            }
        if (live) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { red[(pl * 2 + 0) * C + 4 * q + k] = s1[k]; red[(pl * 2 + 1) * C + 4 * q + k] = s2[k]; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += red[r * 2 * C + i];
        __hip_atomic_store(partial + ((int64_t)b * gridDim.x + blockIdx.x) * 2 * C + i, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!fin.ticket) return;
    __shared__ unsigned int lflag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's partial stores are acknowledged
    __syncthreads();                                        // ... and every wave's
    if (threadIdx.x == 0) lflag = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (lflag != gridDim.x * gridDim.y - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // invalidate only: the other workgroups' partials are read from memory
    if (threadIdx.x == 0) __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    finalize_in_kernel(fin, MODE == 1, partial, (int)gridDim.x, (int)gridDim.y, C, red, red + 1024);
}

// forward: y = lrelu(x * scale[b,c] + shift[b,c] (+ res));  backward (MODE 1): dx = scale * (dz - m1 - xhat * m2), dres = dz (+ res);
// MODE 2 (rstd == NULL): dx = scale * dz - m1 - (x - mean) * m2
template <typename T, int MODE>
__global__ void __launch_bounds__(256) apply_kernel(const T* __restrict__ x, const T* __restrict__ res, const T* __restrict__ dy,
                                                    const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ m1,
                                                    const float* __restrict__ m2, float slope, T* __restrict__ out, T* __restrict__ dres,
                                                    int64_t N, int C, PostMul pm) {
    const int b = blockIdx.y, cqn = C >> 2;
    const int64_t total = N * cqn;
    const int64_t base = (int64_t)b * N * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % cqn);
        const int64_t off = base + (i / cqn) * C + 4 * q;
        const p4c_f32x4 xv = ld4(x + off);
        const p4c_f32x4 pv = pm.at(i / cqn, C, 4 * q);
        p4c_f32x4 o;
        if (MODE == 0) {
            const p4c_f32x4 sc = ld4(scale + b * C + 4 * q), sh = ld4(shift + b * C + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = __builtin_fmaf(xv[k], sc[k], sh[k]);
            if (res) {
                const p4c_f32x4 rv = ld4(res + off);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] += rv[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] *= pv[k];
                o[k] = o[k] > 0.f ? o[k] : o[k] * slope;
            }
            store4f(out + off, o);
        } else {
            const p4c_f32x4 g = ld4(dy + off), yv = ld4(y + off);
            const p4c_f32x4 sc = ld4(scale + b * C + 4 * q), mu = ld4(mean + b * C + 4 * q);
            const p4c_f32x4 a1 = ld4(m1 + b * C + 4 * q), a2 = ld4(m2 + b * C + 4 * q);
            p4c_f32x4 dz;
            if (MODE == 1) {
                const p4c_f32x4 rs = ld4(rstd + b * C + 4 * q);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    dz[k] = (yv[k] > 0.f ? g[k] : g[k] * slope) * pv[k];
                    o[k] = sc[k] * (dz[k] - a1[k] - (xv[k] - mu[k]) * rs[k] * a2[k]);
                }
            } else {   // MODE 2: the caller's coefficients as they are (group norm: statistics shared by the channels of a group)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    dz[k] = (yv[k] > 0.f ? g[k] : g[k] * slope) * pv[k];
                    o[k] = sc[k] * dz[k] - a1[k] - (xv[k] - mu[k]) * a2[k];
                }
            }
            store4f(out + off, o);
            if (dres) {
                if (res) {      // backward: `res` is a gradient that reached the residual operand by another path -- added here, not by a launch
                    const p4c_f32x4 rv = ld4(res + off);
#pragma unroll
                    for (int k = 0; k < 4; ++k) dz[k] += rv[k];
                }
                store4f(dres + off, dz);
            }
        }
    }
}

static int blocks_for(int64_t N, int rows) {
    int64_t nb = (N + (int64_t)rows * 8 - 1) / ((int64_t)rows * 8);    // >= 8 pixels per thread row
    if (nb > 256) nb = 256;
    return (int)(nb < 1 ? 1 : nb);
}

// ---------------------------------------------------------------------------------------------
// Statistics finalize: what ops_inorm did with a dozen tiny torch launches per normalisation (partial.sum, /N, mean^2, clamp, rsqrt,
// casts, muls, contiguous ...: ~1 500 launches per SwinUNetR training step) as ONE launch each way.
// cpg = channels per statistics group (1: instance norm; C / groups: group norm); a workgroup owns whole groups: CB channels,
// CB = 64 rounded down to a multiple of cpg (the host guarantees cpg <= 64).  Threads = CB channels x SL slices over the partial
// blocks; sums in a fixed order (slice-strided, then a tree over the slices), the per-group math in double.
//   forward : mean, rstd, scale = rstd * gamma, shift = beta - mean * scale            -- all (B, C)
//   backward: mode 0 (instance): c1 = S0 / N, c2 = S1 / N            (p4c_inorm_apply forms dx = scale (dz - c1 - xhat c2))
//             mode 1 (group)   : c1 = rstd M1, c2 = rstd^2 M2 with M1 / M2 = mean over the group of gamma S0 / gamma S1
//             and dgamma[c] = sum_b S1[b, c], dbeta[c] = sum_b S0[b, c] (written, not accumulated), b in order.
// per-channel sums of one sample's partial blocks into red0 / red1[0 .. CB): threads = (CB / 4 channel quads) x SL slices over the
// blocks (16-byte loads; CB is a multiple of 4 -- finalize_geometry), slices combined by a tree, every sum in a fixed order
__device__ __forceinline__ void slice_sums(const float* __restrict__ part, int nb, int C, int c_lo, int CB, int SL, float* red0, float* red1) {
    const int nq = CB >> 2;
    const int q = threadIdx.x % nq, sl = threadIdx.x / nq;
    const bool active = sl < SL;            // (256 threads need not be a multiple of the quads: the rest only meet the barriers)
    const int c = c_lo + 4 * q;
    if (active) {
        p4c_f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        if (c < C)
            for (int k = sl; k < nb; k += SL) {
                s0 += *reinterpret_cast<const p4c_f32x4*>(part + ((int64_t)k * 2 + 0) * C + c);
                s1 += *reinterpret_cast<const p4c_f32x4*>(part + ((int64_t)k * 2 + 1) * C + c);
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) { red0[sl * CB + 4 * q + j] = s0[j]; red1[sl * CB + 4 * q + j] = s1[j]; }
    }
    __syncthreads();
    for (int off = SL >> 1; off > 0; off >>= 1) {
        if (active && sl < off) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red0[sl * CB + 4 * q + j] += red0[(sl + off) * CB + 4 * q + j];
                red1[sl * CB + 4 * q + j] += red1[(sl + off) * CB + 4 * q + j];
            }
        }
        __syncthreads();
    }
}

// instance form (one channel per group) of both finalizes for ALL samples, by one workgroup of 256 threads (see InFin)
__device__ void finalize_in_kernel(const InFin& f, bool bwd, const float* partial, int nb, int B, int C, float* red0, float* red1) {
    const int tid_c = threadIdx.x;
    for (int c_lo = 0; c_lo < C; c_lo += f.CB) {
        const int c = c_lo + tid_c;
        const bool mine = tid_c < f.CB && c < C;
        float dg = 0.f, db = 0.f;
        for (int b = 0; b < B; ++b) {
            __syncthreads();
            slice_sums(partial + (int64_t)b * nb * 2 * C, nb, C, c_lo, f.CB, f.SL, red0, red1);
            if (mine) {
                if (!bwd) {
                    const double mu = (double)red0[tid_c] / f.n_group;
                    double var = (double)red1[tid_c] / f.n_group - mu * mu;
                    var = var > 0.0 ? var : 0.0;
                    const float r = (float)(1.0 / sqrt(var + (double)f.eps)), m = (float)mu;
                    const float sc = r * f.gamma[c];
                    f.o0[(int64_t)b * C + c] = m;
                    f.o1[(int64_t)b * C + c] = r;
                    f.o2[(int64_t)b * C + c] = sc;
                    f.o3[(int64_t)b * C + c] = f.beta[c] - m * sc;
                } else {
                    const float S0 = red0[tid_c], S1 = red1[tid_c];
                    db += S0;
                    dg += S1;
                    f.o0[(int64_t)b * C + c] = (float)((double)S0 / f.n_group);
                    f.o1[(int64_t)b * C + c] = (float)((double)S1 / f.n_group);
                }
            }
        }
        if (bwd && mine) {
            f.o2[c] = dg;
            f.o3[c] = db;
        }
    }
}

__global__ void __launch_bounds__(256) finalize_fwd_kernel(const float* __restrict__ part, int nb, int C, int cpg, double n_group,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                           float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ scale,
                                                           float* __restrict__ shift, int CB, int SL) {
    __shared__ float red0[1024], red1[1024];
    const int b = blockIdx.y, c_lo = blockIdx.x * CB;
    slice_sums(part + (int64_t)b * nb * 2 * C, nb, C, c_lo, CB, SL, red0, red1);
    const int tid_c = threadIdx.x, c = c_lo + tid_c;
    if (tid_c < CB && c < C) {
        const int g0 = tid_c - tid_c % cpg;
        double s0 = 0.0, s1 = 0.0;
        for (int j = 0; j < cpg; ++j) { s0 += (double)red0[g0 + j]; s1 += (double)red1[g0 + j]; }
        const double mu = s0 / n_group;
        double var = s1 / n_group - mu * mu;
        var = var > 0.0 ? var : 0.0;
        const float r = (float)(1.0 / sqrt(var + (double)eps)), m = (float)mu;
        const float sc = r * gamma[c];
        mean[(int64_t)b * C + c] = m;
        rstd[(int64_t)b * C + c] = r;
        scale[(int64_t)b * C + c] = sc;
        shift[(int64_t)b * C + c] = beta[c] - m * sc;
    }
}

__global__ void __launch_bounds__(256) finalize_bwd_kernel(const float* __restrict__ part, int nb, int B, int C, int cpg, double n_group,
                                                           int mode, const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                           float* __restrict__ c1, float* __restrict__ c2, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int CB, int SL) {
    __shared__ float red0[1024], red1[1024];
    const int c_lo = blockIdx.x * CB;
    const int tid_c = threadIdx.x, c = c_lo + tid_c;
    float dg = 0.f, db = 0.f;
    for (int b = 0; b < B; ++b) {
        slice_sums(part + (int64_t)b * nb * 2 * C, nb, C, c_lo, CB, SL, red0, red1);
        if (tid_c < CB && c < C) {
            const float S0 = red0[tid_c], S1 = red1[tid_c];
            db += S0;
            dg += S1;
            if (mode == 0) {
                c1[(int64_t)b * C + c] = (float)((double)S0 / n_group);
                c2[(int64_t)b * C + c] = (float)((double)S1 / n_group);
            } else {
                const int g0 = tid_c - tid_c % cpg, cg0 = c - tid_c % cpg;
                double m1 = 0.0, m2 = 0.0;
                for (int j = 0; j < cpg; ++j) {
                    m1 += (double)gamma[cg0 + j] * (double)red0[g0 + j];
                    m2 += (double)gamma[cg0 + j] * (double)red1[g0 + j];
                }
                const double r = (double)rstd[(int64_t)b * C + c];
                c1[(int64_t)b * C + c] = (float)(r * m1 / n_group);
                c2[(int64_t)b * C + c] = (float)(r * r * m2 / n_group);
            }
        }
        __syncthreads();
    }
    if (tid_c < CB && c < C) {
        dgamma[c] = dg;
        dbeta[c] = db;
    }
}

static void finalize_geometry(int C, int cpg, int* CB, int* SL, int* blocks) {
    // whole groups per workgroup, a multiple of 4 channels (16-byte loads of the partials): lcm(cpg, 4) divides CB
    int unit = cpg;
    while (unit % 4) unit += cpg;
    int cb = unit <= 64 ? 64 - 64 % unit : unit;
    if (cb > C) cb = C;                   // (C is a multiple of 4 and of cpg: so is C itself)
    int sl = 1;
    while (sl * 2 * (cb / 4) <= 256 && sl * 2 * cb <= 1024) sl *= 2;   // power of two slices (tree), red0 / red1 hold SL x CB floats
    *CB = cb;
    *SL = sl;
    *blocks = (C + cb - 1) / cb;
}

}  // namespace inorm
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_inorm_blocks(int64_t N, int C) {
    const int cqn = C / 4, rows = 256 / cqn > 0 ? 256 / cqn : 1;
    return inorm::blocks_for(N, rows);
}

static int inorm_reduce_launch(const void* x, const void* dy, const void* y, const float* mean, const float* rstd, float slope, float* partial,
                               int dtype, int B, int64_t N, int C, const inorm::InFin& fin, p4c_stream_t stream,
                               const inorm::PostMul& pm = inorm::PostMul{nullptr, 1, 1.f}) {
    P4C_CHECK_ARG(x && partial && B > 0 && N > 0 && C > 0 && C % 4 == 0 && C <= 1024, "p4c_inorm_reduce: C must be a multiple of 4 up to 1024");
    const bool bwd = dy != nullptr;
    P4C_CHECK_ARG(!bwd || (y && mean && rstd), "p4c_inorm_reduce: the backward sums need y, mean and rstd");
    const int cqn = C / 4, rows = 256 / cqn > 0 ? 256 / cqn : 1;
    const dim3 grid(inorm::blocks_for(N, rows), B);
    size_t smem = (size_t)rows * 2 * C * sizeof(float);
    if (smem < 2048 * sizeof(float)) smem = 2048 * sizeof(float);
    hipStream_t st = as_stream(stream);
#define P4C_IN_RED(T, M)                                                                                             \
    do {                                                                                                             \
        P4C_TRY(ensure_dyn_smem((const void*)inorm::reduce_kernel<T, M>, 64 * 1024));                                \
        hipLaunchKernelGGL((inorm::reduce_kernel<T, M>), grid, dim3(256), smem, st, (const T*)x, (const T*)dy, (const T*)y, mean, \
                           rstd, slope, partial, N, C, fin, pm);                                                     \
    } while (0)
    if (dtype == P4C_F32) { if (bwd) P4C_IN_RED(float, 1); else P4C_IN_RED(float, 0); }
    else if (dtype == P4C_BF16) { if (bwd) P4C_IN_RED(bf16, 1); else P4C_IN_RED(bf16, 0); }
    else return fail(P4C_ERR_INVALID, "p4c_inorm_reduce: bad dtype");
#undef P4C_IN_RED
    P4C_CHECK_LAUNCH("p4c_inorm_reduce");
    return P4C_OK;
}

extern "C" int p4c_inorm_reduce(const void* x, const void* dy, const void* y, const float* mean, const float* rstd, float slope,
                                float* partial, int dtype, int B, int64_t N, int C, p4c_stream_t stream) {
    return inorm_reduce_launch(x, dy, y, mean, rstd, slope, partial, dtype, B, N, C, inorm::InFin{}, stream);
}

extern "C" int p4c_inorm_reduce_finalize_fwd(const void* x, float* partial, unsigned int* ticket, const float* gamma, const float* beta, float eps,
                                             float* mean, float* rstd, float* scale, float* shift, int dtype, int B, int64_t N, int C,
                                             p4c_stream_t stream) {
    P4C_CHECK_ARG(ticket && gamma && beta && mean && rstd && scale && shift, "p4c_inorm_reduce_finalize_fwd: NULL pointer");
    inorm::InFin fin{ticket, gamma, beta, eps, mean, rstd, scale, shift, (double)N, 0, 0};
    int blocks;
    inorm::finalize_geometry(C > 0 ? C : 4, 1, &fin.CB, &fin.SL, &blocks);
    return inorm_reduce_launch(x, nullptr, nullptr, nullptr, nullptr, 1.f, partial, dtype, B, N, C, fin, stream);
}

extern "C" int p4c_inorm_reduce_finalize_bwd(const void* x, const void* dy, const void* y, const float* mean, const float* rstd, float slope,
                                             float* partial, unsigned int* ticket, float* c1, float* c2, float* dgamma, float* dbeta, int dtype,
                                             int B, int64_t N, int C, p4c_stream_t stream) {
    P4C_CHECK_ARG(ticket && dy && c1 && c2 && dgamma && dbeta, "p4c_inorm_reduce_finalize_bwd: NULL pointer");
    inorm::InFin fin{ticket, nullptr, nullptr, 0.f, c1, c2, dgamma, dbeta, (double)N, 0, 0};
    int blocks;
    inorm::finalize_geometry(C > 0 ? C : 4, 1, &fin.CB, &fin.SL, &blocks);
    return inorm_reduce_launch(x, dy, y, mean, rstd, slope, partial, dtype, B, N, C, fin, stream);
}

static int inorm_apply_launch(const void* x, const void* res, const void* dy, const void* y, const float* scale, const float* shift,
                              const float* mean, const float* rstd, const float* m1, const float* m2, float slope, void* out, void* dres,
                              int dtype, int B, int64_t N, int C, p4c_stream_t stream, const inorm::PostMul& pm) {
    P4C_CHECK_ARG(x && out && scale && B > 0 && N > 0 && C > 0 && C % 4 == 0, "p4c_inorm_apply: bad arguments");
    const bool bwd = dy != nullptr;
    P4C_CHECK_ARG(bwd ? (y && mean && m1 && m2) : (shift != nullptr), "p4c_inorm_apply: missing operands");
    int64_t blocks = (N * (C / 4) + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 8 / B + 1;
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks, B);
    hipStream_t st = as_stream(stream);
#define P4C_IN_APP(T, M)                                                                                                        \
    hipLaunchKernelGGL((inorm::apply_kernel<T, M>), grid, dim3(256), 0, st, (const T*)x, (const T*)res, (const T*)dy, (const T*)y, scale, \
                       shift, mean, rstd, m1, m2, slope, (T*)out, (T*)dres, N, C, pm)
    if (dtype == P4C_F32) { if (!bwd) P4C_IN_APP(float, 0); else if (rstd) P4C_IN_APP(float, 1); else P4C_IN_APP(float, 2); }
    else if (dtype == P4C_BF16) { if (!bwd) P4C_IN_APP(bf16, 0); else if (rstd) P4C_IN_APP(bf16, 1); else P4C_IN_APP(bf16, 2); }
    else return fail(P4C_ERR_INVALID, "p4c_inorm_apply: bad dtype");
#undef P4C_IN_APP
    P4C_CHECK_LAUNCH("p4c_inorm_apply");
    return P4C_OK;
}

extern "C" int p4c_inorm_apply(const void* x, const void* res, const void* dy, const void* y, const float* scale, const float* shift,
                               const float* mean, const float* rstd, const float* m1, const float* m2, float slope, void* out, void* dres,
                               int dtype, int B, int64_t N, int C, p4c_stream_t stream) {
    return inorm_apply_launch(x, res, dy, y, scale, shift, mean, rstd, m1, m2, slope, out, dres, dtype, B, N, C, stream,
                              inorm::PostMul{nullptr, 1, 1.f});
}

static int check_postmul(const char* name, const float* mul, int64_t mul_rows, int64_t N) {
    P4C_CHECK_ARG(mul && mul_rows > 0 && N % mul_rows == 0 && (reinterpret_cast<uintptr_t>(mul) & 15) == 0,
                  "%s: the multiplier needs a 16-byte aligned table and a row count that divides the map's rows", name);
    return P4C_OK;
}

extern "C" int p4c_inorm_apply_mul(const void* x, const void* res, const void* dy, const void* y, const float* scale, const float* shift,
                                   const float* mean, const float* rstd, const float* m1, const float* m2, float slope, void* out, void* dres,
                                   int dtype, int B, int64_t N, int C, const float* mul, int64_t mul_rows, float mul_factor, p4c_stream_t stream) {
    P4C_TRY(check_postmul("p4c_inorm_apply_mul", mul, mul_rows, N));
    return inorm_apply_launch(x, res, dy, y, scale, shift, mean, rstd, m1, m2, slope, out, dres, dtype, B, N, C, stream,
                              inorm::PostMul{mul, mul_rows, mul_factor});
}

extern "C" int p4c_inorm_reduce_mul(const void* x, const void* dy, const void* y, const float* mean, const float* rstd, float slope,
                                    float* partial, int dtype, int B, int64_t N, int C, const float* mul, int64_t mul_rows, float mul_factor,
                                    p4c_stream_t stream) {
    P4C_TRY(check_postmul("p4c_inorm_reduce_mul", mul, mul_rows, N));
    P4C_CHECK_ARG(dy != nullptr, "p4c_inorm_reduce_mul: the backward sums only (the forward statistics see no multiplier)");
    return inorm_reduce_launch(x, dy, y, mean, rstd, slope, partial, dtype, B, N, C, inorm::InFin{}, stream, inorm::PostMul{mul, mul_rows, mul_factor});
}

extern "C" int p4c_inorm_finalize_fwd(const float* partial, int nblk, int B, int64_t N, int C, int groups, const float* gamma, const float* beta,
                                      float eps, float* mean, float* rstd, float* scale, float* shift, p4c_stream_t stream) {
    P4C_CHECK_ARG(partial && gamma && beta && mean && rstd && scale && shift && B > 0 && N > 0 && nblk > 0, "p4c_inorm_finalize_fwd: bad arguments");
    const int cpg = groups > 0 ? C / groups : 1;
    P4C_CHECK_ARG(C > 0 && (groups <= 0 || C % groups == 0) && cpg <= 64, "p4c_inorm_finalize_fwd: at most 64 channels per group (C %d, groups %d)", C, groups);
    int CB, SL, blocks;
    inorm::finalize_geometry(C, cpg, &CB, &SL, &blocks);
    hipLaunchKernelGGL(inorm::finalize_fwd_kernel, dim3(blocks, B), dim3(256), 0, as_stream(stream), partial, nblk, C, cpg, (double)N * cpg, gamma,
                       beta, eps, mean, rstd, scale, shift, CB, SL);
    P4C_CHECK_LAUNCH("p4c_inorm_finalize_fwd");
    return P4C_OK;
}

extern "C" int p4c_inorm_finalize_bwd(const float* partial, int nblk, int B, int64_t N, int C, int groups, const float* gamma, const float* rstd,
                                      float* c1, float* c2, float* dgamma, float* dbeta, p4c_stream_t stream) {
    P4C_CHECK_ARG(partial && c1 && c2 && dgamma && dbeta && B > 0 && N > 0 && nblk > 0, "p4c_inorm_finalize_bwd: bad arguments");
    const int cpg = groups > 0 ? C / groups : 1;
    P4C_CHECK_ARG(C > 0 && (groups <= 0 || (C % groups == 0 && gamma && rstd)) && cpg <= 64,
                  "p4c_inorm_finalize_bwd: at most 64 channels per group; the group form needs gamma and rstd");
    int CB, SL, blocks;
    inorm::finalize_geometry(C, cpg, &CB, &SL, &blocks);
    hipLaunchKernelGGL(inorm::finalize_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), partial, nblk, B, C, cpg, (double)N * cpg,
                       groups > 0 ? 1 : 0, gamma, rstd, c1, c2, dgamma, dbeta, CB, SL);
    P4C_CHECK_LAUNCH("p4c_inorm_finalize_bwd");
    return P4C_OK;
}
