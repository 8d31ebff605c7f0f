// Normalisation (BatchNorm2d / GroupNorm), 2x2 max-pool and bilinear up-sample-and-sum kernels of the
// HalfUNet path, forward and backward.  All HBM-bound streaming passes over NHWC tensors with
// C = 64 channels: a thread owns 4 consecutive channels (16-byte accesses), 16 threads per pixel.
//
// "Deferred normalisation": a conv writes its RAW output y plus per-tile channel sums; the
// normalised activation a = relu(y*scale[b,c] + shift[b,c]) is applied by whoever reads y next
// (next conv's tile staging, the pool, the up-sample-and-sum, the backward passes) and is never
// written to HBM.  scale/shift/mean/rstd are (B,64) arrays for both norm types (BatchNorm repeats
// the same row for every b), so consumers are norm-agnostic.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

namespace p4c {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C = 64;

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// 4 consecutive channels of one pixel, activations stored as fp32 (16 B) or bf16 (8 B); arithmetic is fp32
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ f32x4 ld4(const __bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void st4(__bf16* p, f32x4 v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(p) = o;
}
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}

// ------------------------------------------------------------------------------------------
// norm_finalize: conv epilogue partials [B*tiles_per_sample][2][64] -> scale/shift/mean/rstd (B,64)
//   mode 0 (BatchNorm2d, training): statistics over (B,H,W) per channel, biased variance for the
//          normalisation, running stats updated with the unbiased one (torch semantics).
//   mode 1 (GroupNorm): statistics over (H,W,channels of the group) per sample.
// grid: mode 0 -> 64 blocks (one per channel); mode 1 -> B*groups blocks.  256 threads.
__global__ void __launch_bounds__(256)
    norm_finalize_kernel(const float* __restrict__ partial, int tiles_per_sample, int B, int64_t hw, int mode, int groups,
                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                         float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ scale,
                         float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ double red[2][256];
    double s1 = 0.0, s2 = 0.0;
    int c_lo, c_hi, b_lo, b_hi;
    // BatchNorm: the channel's parameters are requested before the partial sums are (one memory round trip less on the critical
    // path of a kernel that is nothing but dependent round trips)
    float pf_gamma = 0.f, pf_beta = 0.f, pf_rm = 0.f, pf_rv = 0.f;
    if (mode == 0) {
        pf_gamma = gamma[blockIdx.x];
        pf_beta = beta[blockIdx.x];
        if (running_mean) { pf_rm = running_mean[blockIdx.x]; pf_rv = running_var[blockIdx.x]; }
    }
    if (mode == 0) {
        c_lo = blockIdx.x; c_hi = c_lo + 1; b_lo = 0; b_hi = B;
    } else {
        const int cpg = C / groups;
        const int b = blockIdx.x / groups, g = blockIdx.x - b * groups;
        c_lo = g * cpg; c_hi = c_lo + cpg; b_lo = b; b_hi = b + 1;
    }
    const int nc = c_hi - c_lo;
    const int64_t ntile = (int64_t)(b_hi - b_lo) * tiles_per_sample;
    for (int64_t i = threadIdx.x; i < ntile * nc; i += 256) {
        const int64_t t = i / nc;
        const int c = c_lo + (int)(i - t * nc);
        const int64_t tile = (int64_t)b_lo * tiles_per_sample + t;
        s1 += (double)partial[tile * 128 + c];
        s2 += (double)partial[tile * 128 + 64 + c];
    }
    // wave-level butterfly (no LDS round trips), then one 4-way combine through LDS: the result lands in red[.][0]
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s1 += __shfl_xor(s1, off);
        s2 += __shfl_xor(s2, off);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s1;
        red[1][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        red[0][0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        red[1][0] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
    __syncthreads();
    const double n = (double)(b_hi - b_lo) * (double)hw * (double)nc;
    const double mean = red[0][0] / n;
    double var = red[1][0] / n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    if (mode == 0) {
        const int c = c_lo;
        if (threadIdx.x == 0 && running_mean) {
            running_mean[c] = (1.f - momentum) * pf_rm + momentum * (float)mean;
            const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
            running_var[c] = (1.f - momentum) * pf_rv + momentum * (float)unbiased;
        }
        const float sc = pf_gamma * rstd, sh = pf_beta - (float)mean * sc;
        for (int b = threadIdx.x; b < B; b += 256) {
            scale[b * C + c] = sc; shift[b * C + c] = sh; mean_out[b * C + c] = (float)mean; rstd_out[b * C + c] = rstd;
        }
    } else {
        for (int c = c_lo + threadIdx.x; c < c_hi; c += 256) {
            const float sc = gamma[c] * rstd;
            scale[b_lo * C + c] = sc; shift[b_lo * C + c] = beta[c] - (float)mean * sc;
            mean_out[b_lo * C + c] = (float)mean; rstd_out[b_lo * C + c] = rstd;
        }
    }
}

// BatchNorm2d in eval mode: scale/shift from the running statistics.
__global__ void norm_eval_kernel(int B, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                 const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                 float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                 float* __restrict__ rstd_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    const int c = i & 63;
    const float rstd = 1.0f / sqrtf(running_var[c] + eps);
    const float sc = gamma[c] * rstd;
    scale[i] = sc; shift[i] = beta[c] - running_mean[c] * sc; mean_out[i] = running_mean[c]; rstd_out[i] = rstd;
}

// ------------------------------------------------------------------------------------------
// norm backward, pass 1: per-(b,c) sums of g and g*xhat, g = dA * (y*scale+shift > 0),
// xhat = (y-mean)*rstd.  grid (nblk, B); partial [b][blk][2][64].
template <typename T>
__global__ void __launch_bounds__(256)
    norm_bwd_reduce_kernel(const T* __restrict__ dA, const T* __restrict__ y, const float* __restrict__ scale,
                           const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                           int relu, int64_t hw, float* __restrict__ partial) {
    __shared__ float red[2][16][64];
    const int b = blockIdx.y;
    const int c4 = threadIdx.x & 15, pl = threadIdx.x >> 4;  // 16 pixels per block iteration
    const f32x4 sc = ld4(scale + b * C + 4 * c4), sh = ld4(shift + b * C + 4 * c4);
    const f32x4 mu = ld4(mean + b * C + 4 * c4), rs = ld4(rstd + b * C + 4 * c4);
    f32x4 a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
    const T* yb = y + (int64_t)b * hw * C;
    const T* gb = dA + (int64_t)b * hw * C;
    for (int64_t p = (int64_t)blockIdx.x * 16 + pl; p < hw; p += (int64_t)gridDim.x * 16) {
        const f32x4 yv = ld4(yb + p * C + 4 * c4);
        f32x4 g = ld4(gb + p * C + 4 * c4);
        if (relu) {
            const f32x4 pre = yv * sc + sh;
            g.x = pre.x > 0.f ? g.x : 0.f; g.y = pre.y > 0.f ? g.y : 0.f;
            g.z = pre.z > 0.f ? g.z : 0.f; g.w = pre.w > 0.f ? g.w : 0.f;
        }
        a1 += g;
        a2 += g * ((yv - mu) * rs);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[0][pl][4 * c4 + j] = a1[j];
        red[1][pl][4 * c4 + j] = a2[j];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int st = threadIdx.x >> 6, c = threadIdx.x & 63;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[st][k][c];
        partial[(((int64_t)b * gridDim.x + blockIdx.x) * 2 + st) * 64 + c] = s;
    }
}

// Epilogue shared by the kernels that take pass 1 of a normalisation backward (grid (nblk, B), 256 threads): red[st][pl][c] holds the
// 32 pixel lanes' sums of g (st 0) and g * xhat (st 1); the workgroup's slot goes to partial[b][blockIdx.x][st][c].  With fin.ticket
// set the LAST workgroup of the launch then finishes the pass (kernels.hpp: BwdFin) -- the slots summed in a fixed order whichever
// workgroup comes last: bitwise reproducible.
__device__ __forceinline__ void bwd_slot_finish(float (&red)[2][32][65], float* __restrict__ partial, const BwdFin& fin) {
    const int b = blockIdx.y, nblk = gridDim.x;
    __syncthreads();
    if (threadIdx.x < 128) {
        const int st = threadIdx.x >> 6, c = threadIdx.x & 63;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s += red[st][k][c];
        float* dst = partial + (((int64_t)b * nblk + blockIdx.x) * 2 + st) * 64 + c;
        if (fin.ticket) __hip_atomic_store(dst, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (write-through: see conv_rows.hip, BatchFin)
        else *dst = s;
    }
    if (!fin.ticket) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // every slot store of this workgroup is acknowledged (and red[] is free)
    unsigned int* lflag = reinterpret_cast<unsigned int*>(&red[0][0][0]);
    float* lsum = &red[1][0][0];   // [8][128] (16-byte aligned: the kernels declare red so)
    const int nslots = nblk * gridDim.y;
    if (threadIdx.x == 0) *lflag = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*lflag != (unsigned)nslots - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // invalidate only: the other workgroups' slots are read from memory
    // 256 threads: 32 column quads of the 128 sums x 8 slot groups; slots in increasing order within a group, groups in order
    const int cq = threadIdx.x & 31, sg = threadIdx.x >> 5;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = sg; s0 < nslots; s0 += 64) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int sl = s0 + 8 * u;
            v[u] = sl < nslots ? *(reinterpret_cast<const f32x4*>(partial + (int64_t)sl * 128) + cq) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    __syncthreads();   // (lflag has been read by everyone)
    *reinterpret_cast<f32x4*>(lsum + sg * 128 + 4 * cq) = acc;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        float tot1 = 0.f, tot2 = 0.f;
#pragma unroll
        for (int g8 = 0; g8 < 8; ++g8) { tot1 += lsum[g8 * 128 + c]; tot2 += lsum[g8 * 128 + 64 + c]; }
        fin.dgamma[c] += tot2;
        fin.dbeta[c] += tot1;
        const float ga = fin.gamma[c];
        const float a = fin.training ? ga * tot1 / fin.count : 0.f, bb = fin.training ? ga * tot2 / fin.count : 0.f;
        for (int bi = 0; bi < fin.B; ++bi) { fin.k1[bi * C + c] = a; fin.k2[bi * C + c] = bb; }
    }
    if (threadIdx.x == 0) __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The same pass for bf16 storage with 16-byte loads (8 channels per thread, 8 threads per pixel, 32 pixels per block iteration)
// and four iterations of both tensors in flight per thread: the 8-byte-per-lane form above streams the full-resolution maps at
// 3.6 TB/s (37 us per launch, rocprofv3), this one is what the row kernels of rows.hip reach (>5 TB/s).  Same partial layout.
typedef unsigned int norm_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void unpack8(const norm_u32x4& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__global__ void __launch_bounds__(256)
    norm_bwd_reduce_bf16x8_kernel(const __bf16* __restrict__ dA, const __bf16* __restrict__ y, const float* __restrict__ scale,
                                  const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                                  int relu, int64_t hw, float* __restrict__ partial, BwdFin fin) {
    __shared__ __attribute__((aligned(16))) float red[2][32][65];
    const int b = blockIdx.y;
    const int c8 = threadIdx.x & 7, pl = threadIdx.x >> 3;   // 32 pixels per block iteration
    float sc[8], sh[8], mu[8], rs[8], a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale[b * C + 8 * c8 + j]; sh[j] = shift[b * C + 8 * c8 + j];
        mu[j] = mean[b * C + 8 * c8 + j]; rs[j] = rstd[b * C + 8 * c8 + j];
        a1[j] = a2[j] = 0.f;
    }
    const norm_u32x4* yb = reinterpret_cast<const norm_u32x4*>(y + (int64_t)b * hw * C);
    const norm_u32x4* gb = reinterpret_cast<const norm_u32x4*>(dA + (int64_t)b * hw * C);
    const int64_t stride = (int64_t)gridDim.x * 32;
    for (int64_t p0 = (int64_t)blockIdx.x * 32 + pl; p0 < hw; p0 += 4 * stride) {
        norm_u32x4 vy[4], vg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = p0 + u * stride;
            vy[u] = vg[u] = norm_u32x4{0u, 0u, 0u, 0u};
            if (p < hw) {
                vy[u] = yb[p * 8 + c8];
                vg[u] = gb[p * 8 + c8];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float fy[8], fg[8];
            unpack8(vy[u], fy);
            unpack8(vg[u], fg);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float g = fg[j];
                if (relu) g = (fy[j] * sc[j] + sh[j]) > 0.f ? g : 0.f;
                a1[j] += g;
                a2[j] += g * ((fy[j] - mu[j]) * rs[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[0][pl][8 * c8 + j] = a1[j];
        red[1][pl][8 * c8 + j] = a2[j];
    }
    bwd_slot_finish(red, partial, fin);
}

// pass 1b: reduce partials; accumulate dgamma/dbeta; emit per-(b,c) k1,k2 so that
//   dY = rstd * (gamma*g - k1 - xhat*k2)          (training statistics)
//   dY = scale * g  (k1 = k2 = 0)                 (eval-mode BatchNorm: statistics are constants)
// One block per statistics group: BatchNorm -> one channel (64 blocks), GroupNorm -> the C/groups channels
// of one group (`groups` blocks).  256 threads = channels of the group x slices over the partial blocks;
// samples are walked in order, so every sum has a fixed order (deterministic).
__global__ void __launch_bounds__(256)
    norm_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int B, int64_t hw, int mode, int groups,
                             int training, const float* __restrict__ gamma, float* __restrict__ dgamma,
                             float* __restrict__ dbeta, float* __restrict__ k1, float* __restrict__ k2) {
    __shared__ float S[2][256];
    const int cpg = mode == 0 ? 1 : C / groups;   // channels per block
    const int c_lo = blockIdx.x * cpg;
    const int cl = threadIdx.x % cpg, sl = threadIdx.x / cpg, nsl = 256 / cpg;
    const int c = c_lo + cl;
    float tot1 = 0.f, tot2 = 0.f;
    if (mode == 0) {
        // BatchNorm: only the totals over the batch are needed -> one pass, wave butterflies, one LDS combine
        const float pf_gamma = gamma[c], pf_dg = dgamma[c], pf_db = dbeta[c];   // requested before the partials (see norm_finalize)
        float s1 = 0.f, s2 = 0.f;
        for (int b = 0; b < B; ++b)
            for (int k = threadIdx.x; k < nblk; k += 256) {
                s1 += partial[(((int64_t)b * nblk + k) * 2 + 0) * 64 + c];
                s2 += partial[(((int64_t)b * nblk + k) * 2 + 1) * 64 + c];
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off);
            s2 += __shfl_xor(s2, off);
        }
        if ((threadIdx.x & 63) == 0) { S[0][threadIdx.x >> 6] = s1; S[1][threadIdx.x >> 6] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            tot1 = (S[0][0] + S[0][1]) + (S[0][2] + S[0][3]);
            tot2 = (S[1][0] + S[1][1]) + (S[1][2] + S[1][3]);
            dgamma[c] = pf_dg + tot2;
            dbeta[c] = pf_db + tot1;
            const float n = (float)B * (float)hw;
            const float a = training ? pf_gamma * tot1 / n : 0.f, bb = training ? pf_gamma * tot2 / n : 0.f;
            for (int b = 0; b < B; ++b) { k1[b * C + c] = a; k2[b * C + c] = bb; }
        }
        return;
    }
    for (int b = 0; b < B; ++b) {
        float s1 = 0.f, s2 = 0.f;
        for (int k = sl; k < nblk; k += nsl) {
            s1 += partial[(((int64_t)b * nblk + k) * 2 + 0) * 64 + c];
            s2 += partial[(((int64_t)b * nblk + k) * 2 + 1) * 64 + c];
        }
        S[0][threadIdx.x] = s1;
        S[1][threadIdx.x] = s2;
        __syncthreads();
        for (int off = nsl >> 1; off > 0; off >>= 1) {  // tree over the slices of each channel
            if (sl < off) {
                S[0][threadIdx.x] += S[0][threadIdx.x + off * cpg];
                S[1][threadIdx.x] += S[1][threadIdx.x + off * cpg];
            }
            __syncthreads();
        }
        // S[.][cl] (sl == 0 rows) now hold the per-channel sums of sample b
        if (sl == 0) { tot1 += S[0][cl]; tot2 += S[1][cl]; }
        if (mode == 1 && sl == 0) {
            float m1 = 0.f, m2 = 0.f;
            for (int j = 0; j < cpg; ++j) {
                m1 += gamma[c_lo + j] * S[0][j];
                m2 += gamma[c_lo + j] * S[1][j];
            }
            const float n = (float)hw * (float)cpg;
            k1[b * C + c] = m1 / n;
            k2[b * C + c] = m2 / n;
        }
        __syncthreads();
    }
    if (sl == 0) {
        dgamma[c] += tot2;
        dbeta[c] += tot1;
        if (mode == 0) {
            const float n = (float)B * (float)hw;
            const float a = training ? gamma[c] * tot1 / n : 0.f, bb = training ? gamma[c] * tot2 / n : 0.f;
            for (int b = 0; b < B; ++b) { k1[b * C + c] = a; k2[b * C + c] = bb; }
        }
    }
}

// pass 2: dY = rstd*(gamma*g - k1 - xhat*k2), written over dA (in place allowed).
template <typename T>
__global__ void __launch_bounds__(256)
    norm_bwd_apply_kernel(const T* __restrict__ dA, const T* __restrict__ y, const float* __restrict__ scale,
                          const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                          const float* __restrict__ gamma, const float* __restrict__ k1, const float* __restrict__ k2,
                          int relu, int64_t hw, int B, T* __restrict__ dY) {
    const int64_t total = (int64_t)B * hw * 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i & 15);
        const int64_t pix = i >> 4;
        const int b = (int)(pix / hw);
        const f32x4 yv = ld4(y + pix * C + 4 * c4);
        f32x4 g = ld4(dA + pix * C + 4 * c4);
        const f32x4 mu = ld4(mean + b * C + 4 * c4), rs = ld4(rstd + b * C + 4 * c4);
        if (relu) {
            const f32x4 pre = yv * ld4(scale + b * C + 4 * c4) + ld4(shift + b * C + 4 * c4);
            g.x = pre.x > 0.f ? g.x : 0.f; g.y = pre.y > 0.f ? g.y : 0.f;
            g.z = pre.z > 0.f ? g.z : 0.f; g.w = pre.w > 0.f ? g.w : 0.f;
        }
        const f32x4 xh = (yv - mu) * rs;
        const f32x4 r = rs * (ld4(gamma + 4 * c4) * g - ld4(k1 + b * C + 4 * c4) - xh * ld4(k2 + b * C + 4 * c4));
        st4(dY + pix * C + 4 * c4, r);
    }
}

// ------------------------------------------------------------------------------------------
// pool_fwd: P[b,Y,X,:] = max over the 2x2 window of relu(y*scale+shift)      (H,W even)
template <typename T>
__global__ void __launch_bounds__(256)
    pool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int B,
                    int H, int W, T* __restrict__ P) {
    const int Ho = H / 2, Wo = W / 2;
    const int64_t total = (int64_t)B * Ho * Wo * 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i & 15);
        int64_t pix = i >> 4;
        const int X = (int)(pix % Wo); pix /= Wo;
        const int Y = (int)(pix % Ho);
        const int b = (int)(pix / Ho);
        const f32x4 sc = ld4(scale + b * C + 4 * c4), sh = ld4(shift + b * C + 4 * c4);
        const T* base = y + (((int64_t)b * H + 2 * Y) * W + 2 * X) * C + 4 * c4;
        f32x4 m = relu4(ld4(base) * sc + sh);
        f32x4 v = relu4(ld4(base + C) * sc + sh);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        v = relu4(ld4(base + (int64_t)W * C) * sc + sh);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        v = relu4(ld4(base + (int64_t)W * C + C) * sc + sh);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        st4(P + (i >> 4) * C + 4 * c4, m);
    }
}

// bilinear source index, torch semantics (align_corners=False, scale_factor given): src = (dst+0.5)/s - 0.5, clamped >= 0
__device__ __forceinline__ void bilin(int dst, int s, int n_in, int& i0, int& i1, float& l1) {
    float src = ((float)dst + 0.5f) / (float)s - 0.5f;
    if (src < 0.f) src = 0.f;
    i0 = (int)src;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

template <typename T>
struct UpLevels {
    const T* y[5];          // raw conv outputs of enc1..enc5 (level k at H/2^k)
    const float* scale[5];  // (B,64)
    const float* shift[5];
};

// One level of the up-sample-and-sum for a run of 16 output pixels x 4 channels held by one thread.
// The 16 outputs of a level-K run (scale s = 2^K) interpolate between 16/s + 2 source columns: those are
// normalised (+ReLU) and blended along y ONCE, then each output is one x-blend with compile-time weights.
// Source columns are clamped to the image, which reproduces torch's border rule (src < 0 -> 0, i1 <= n-1)
// because both blended values then coincide.
template <typename T, int K>
__device__ __forceinline__ void up_level_run(const T* __restrict__ yk, const float* __restrict__ scale,
                                             const float* __restrict__ shift, int b, int yy, int x0, int H, int W, int c4,
                                             f32x4 (&acc)[16]) {
    constexpr int s = 1 << K;
    constexpr int NC = 16 / s + 2;
    const int Hk = H >> K, Wk = W >> K;
    int y0, y1;
    float ly;
    bilin(yy, s, Hk, y0, y1, ly);
    const float hy = 1.f - ly;
    const f32x4 sc = ld4(scale + b * C + 4 * c4), sh = ld4(shift + b * C + 4 * c4);
    const T* r0 = yk + ((int64_t)b * Hk + y0) * Wk * C + 4 * c4;
    const T* r1 = yk + ((int64_t)b * Hk + y1) * Wk * C + 4 * c4;
    const int cbase = x0 / s - 1;
    f32x4 v[NC];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        int col = cbase + j;
        col = col < 0 ? 0 : (col > Wk - 1 ? Wk - 1 : col);
        const f32x4 a = relu4(ld4(r0 + (int64_t)col * C) * sc + sh);
        const f32x4 q = relu4(ld4(r1 + (int64_t)col * C) * sc + sh);
        v[j] = hy * a + ly * q;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float t = ((float)i + 0.5f) / (float)s - 0.5f;  // constant-folded after unrolling
        const float fl = t < 0.f ? -1.f : (float)(int)t;
        const int j0 = (int)fl + 1;
        const float lx = t - fl;
        acc[i] += (1.f - lx) * v[j0] + lx * v[j0 + 1];
    }
}

// upsum_fwd: S[b,y,x,:] = sum_k up_{2^k}( relu(y_k*scale_k+shift_k) )   (k = 0..4), bilinear, align_corners=False.
// Thread = 4 channels x 16 consecutive pixels of a row (W % 16 == 0).
template <typename T>
__global__ void __launch_bounds__(256) upsum_fwd_kernel(UpLevels<T> lv, int B, int H, int W, T* __restrict__ S) {
    const int segs = W / 16;
    const int64_t total = (int64_t)B * H * segs * 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i & 15);
        int64_t seg = i >> 4;
        const int x0 = (int)(seg % segs) * 16; seg /= segs;
        const int yy = (int)(seg % H);
        const int b = (int)(seg / H);
        const int64_t pix0 = ((int64_t)b * H + yy) * W + x0;
        f32x4 acc[16];
        {
            const f32x4 sc = ld4(lv.scale[0] + b * C + 4 * c4), sh = ld4(lv.shift[0] + b * C + 4 * c4);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = relu4(ld4(lv.y[0] + (pix0 + j) * C + 4 * c4) * sc + sh);
        }
        up_level_run<T, 1>(lv.y[1], lv.scale[1], lv.shift[1], b, yy, x0, H, W, c4, acc);
        up_level_run<T, 2>(lv.y[2], lv.scale[2], lv.shift[2], b, yy, x0, H, W, c4, acc);
        up_level_run<T, 3>(lv.y[3], lv.scale[3], lv.shift[3], b, yy, x0, H, W, c4, acc);
        up_level_run<T, 4>(lv.y[4], lv.scale[4], lv.shift[4], b, yy, x0, H, W, c4, acc);
#pragma unroll
        for (int j = 0; j < 16; ++j) st4(S + (pix0 + j) * C + 4 * c4, acc[j]);
    }
}

// up_bwd_x4: x pass of the adjoint of the four bilinear up-samplings, all levels from ONE read of dS:
//   Tx_k[b,y,X,:] = sum_x wx_k(x,X) dS[b,y,x,:]   (k = 1..4, s = 2^k), Tx_k is (B,H,W/s,64).
// wx_k(x,X) = max(0, 1 - |src - X|), src = (x+0.5)/s - 0.5, except at the clamped borders (src < 0 -> all weight
// on X = 0; src > Wk-1 -> all weight on X = Wk-1).  One workgroup = one row x 64 pixels (+8 halo each side) staged
// in LDS as fp32; the 2s taps of an output are split over thread groups so every thread does 8 taps per level.
template <typename T>
struct UpBwdOut {
    T* tx[4];
};

template <typename T, int K>
__device__ __forceinline__ void up_bwd_level(const f32x4 (*tile)[16], int64_t row, int x0, int W, int c4, int g, T* __restrict__ out) {
    constexpr int s = 1 << K;
    constexpr int NOUT = 64 / s;                    // outputs of this level in the strip
    constexpr int PARTS = NOUT >= 16 ? 1 : 16 / NOUT;  // thread groups sharing one output
    constexpr int PER = NOUT >= 16 ? NOUT / 16 : 1;    // outputs per thread group
    constexpr int TAPS = 2 * s / PARTS;
    const int Wk = W >> K;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int Xl = PARTS == 1 ? g * PER + q : g / PARTS;
        const int part = PARTS == 1 ? 0 : g % PARTS;
        const int X = x0 / s + Xl;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jj = 0; jj < TAPS; ++jj) {
            const int j = part * TAPS + jj;
            float w = 1.f - fabsf(((float)j + 0.5f) / (float)s - 1.f);
            if ((X == 0 && j < s) || (X == Wk - 1 && j >= s)) w = 1.f;
            acc += w * tile[s * Xl - s / 2 + j + 8][c4];
        }
        if (PARTS >= 2) {
            acc.x += __shfl_xor(acc.x, 16); acc.y += __shfl_xor(acc.y, 16);
            acc.z += __shfl_xor(acc.z, 16); acc.w += __shfl_xor(acc.w, 16);
        }
        if (PARTS >= 4) {
            acc.x += __shfl_xor(acc.x, 32); acc.y += __shfl_xor(acc.y, 32);
            acc.z += __shfl_xor(acc.z, 32); acc.w += __shfl_xor(acc.w, 32);
        }
        if (part == 0 && X < Wk) st4(out + (row * Wk + X) * C + 4 * c4, acc);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) up_bwd_x4_kernel(const T* __restrict__ dS, int H, int W, UpBwdOut<T> o) {
    __shared__ f32x4 tile[80][16];
    const int strips = (W + 63) / 64;
    const int64_t row = blockIdx.x / strips;  // b*H + y
    const int x0 = (int)(blockIdx.x % strips) * 64;
    const int c4 = threadIdx.x & 15, g = threadIdx.x >> 4;
#pragma unroll
    for (int p = g; p < 80; p += 16) {
        const int x = x0 - 8 + p;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (x >= 0 && x < W) v = ld4(dS + (row * W + x) * C + 4 * c4);
        tile[p][c4] = v;
    }
    __syncthreads();
    up_bwd_level<T, 1>(tile, row, x0, W, c4, g, o.tx[0]);
    up_bwd_level<T, 2>(tile, row, x0, W, c4, g, o.tx[1]);
    up_bwd_level<T, 3>(tile, row, x0, W, c4, g, o.tx[2]);
    up_bwd_level<T, 4>(tile, row, x0, W, c4, g, o.tx[3]);
}

// norm backward pass 2 for bf16 storage: 16-byte accesses (8 channels per thread), the seven per-channel constants of the sample in
// registers (grid.y = sample), four pixel rows in flight.
__device__ __forceinline__ norm_u32x4 pack8(const float* f) {
    norm_u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned int lo = __float_as_uint(f[2 * i]), hi = __float_as_uint(f[2 * i + 1]);
        lo = ((lo & 0x7fffffffu) > 0x7f800000u) ? ((lo >> 16) | 0x40u) : ((lo + 0x7fffu + ((lo >> 16) & 1u)) >> 16);
        hi = ((hi & 0x7fffffffu) > 0x7f800000u) ? ((hi >> 16) | 0x40u) : ((hi + 0x7fffu + ((hi >> 16) & 1u)) >> 16);
        v[i] = lo | (hi << 16);
    }
    return v;
}
__global__ void __launch_bounds__(256)
    norm_bwd_apply_bf16x8_kernel(const __bf16* __restrict__ dA, const __bf16* __restrict__ y, const float* __restrict__ scale,
                                 const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
                                 const float* __restrict__ gamma, const float* __restrict__ k1, const float* __restrict__ k2, int relu,
                                 int64_t hw, __bf16* __restrict__ dY) {
    const int b = blockIdx.y;
    const int c8 = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float sc[8], sh[8], mu[8], rs[8], ga[8], a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = b * C + 8 * c8 + j;
        sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; rs[j] = rstd[c];
        ga[j] = gamma[8 * c8 + j]; a1[j] = k1[c]; a2[j] = k2[c];
    }
    const norm_u32x4* yb = reinterpret_cast<const norm_u32x4*>(y + (int64_t)b * hw * C);
    const norm_u32x4* gb = reinterpret_cast<const norm_u32x4*>(dA + (int64_t)b * hw * C);
    norm_u32x4* ob = reinterpret_cast<norm_u32x4*>(dY + (int64_t)b * hw * C);
    const int64_t stride = (int64_t)gridDim.x * 32;
    for (int64_t p0 = (int64_t)blockIdx.x * 32 + pl; p0 < hw; p0 += 4 * stride) {
        norm_u32x4 vy[4], vg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = p0 + u * stride;
            vy[u] = vg[u] = norm_u32x4{0u, 0u, 0u, 0u};
            if (p < hw) {
                vy[u] = yb[p * 8 + c8];
                vg[u] = gb[p * 8 + c8];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t p = p0 + u * stride;
            float fy[8], fg[8], r[8];
            unpack8(vy[u], fy);
            unpack8(vg[u], fg);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float g = fg[j];
                if (relu) g = (fy[j] * sc[j] + sh[j]) > 0.f ? g : 0.f;
                const float xh = (fy[j] - mu[j]) * rs[j];
                r[j] = rs[j] * (ga[j] * g - a1[j] - xh * a2[j]);
            }
            if (p < hw) ob[p * 8 + c8] = pack8(r);
        }
    }
}

// enc_out_bwd: gradient wrt the (post-ReLU) output of encoder level k (at Hk x Wk):
//   dA = [T given]   sum_y wy(y,Y) T[b,y,X,:]          (adjoint of the bilinear up-sample, y pass)
//      + [dS given]  dS[b,Y,X,:]                        (level 1: identity)
//      + [dP given]  dP[b,Y/2,X/2,:] if (Y,X) is the arg-max of its 2x2 window of relu(y*scale+shift)
//                    (first maximum in row-major order, as torch's max_pool2d)
template <typename T>
__global__ void __launch_bounds__(256)
    enc_out_bwd_kernel(const T* __restrict__ Tx, int Hfull, int s, const T* __restrict__ dS,
                       const T* __restrict__ dP, const T* __restrict__ y, const float* __restrict__ scale,
                       const float* __restrict__ shift, int B, int Hk, int Wk, T* __restrict__ dA) {
    const int64_t total = (int64_t)B * Hk * Wk * 16;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i & 15);
        int64_t pix = i >> 4;
        const int X = (int)(pix % Wk); pix /= Wk;
        const int Y = (int)(pix % Hk);
        const int b = (int)(pix / Hk);
        f32x4 acc = {0, 0, 0, 0};
        if (dS) acc = ld4(dS + (i >> 4) * C + 4 * c4);
        if (Tx) {
            int ya = s * Y - s / 2, yb = s * Y + 3 * s / 2 - 1;
            if (ya < 0) ya = 0;
            if (yb > Hfull - 1) yb = Hfull - 1;
            for (int yy = ya; yy <= yb; ++yy) {
                int y0, y1; float ly;
                bilin(yy, s, Hk, y0, y1, ly);
                const float w = (y0 == Y ? 1.f - ly : 0.f) + (y1 == Y ? ly : 0.f);
                if (w != 0.f) acc += w * ld4(Tx + (((int64_t)b * Hfull + yy) * Wk + X) * C + 4 * c4);
            }
        }
        if (dP) {
            const f32x4 sc = ld4(scale + b * C + 4 * c4), sh = ld4(shift + b * C + 4 * c4);
            const int Y0 = Y & ~1, X0 = X & ~1;
            const T* base = y + (((int64_t)b * Hk + Y0) * Wk + X0) * C + 4 * c4;
            f32x4 v[4];
            v[0] = relu4(ld4(base) * sc + sh);
            v[1] = relu4(ld4(base + C) * sc + sh);
            v[2] = relu4(ld4(base + (int64_t)Wk * C) * sc + sh);
            v[3] = relu4(ld4(base + (int64_t)Wk * C + C) * sc + sh);
            const int me = (Y & 1) * 2 + (X & 1);
            const f32x4 g = ld4(dP + (((int64_t)b * (Hk / 2) + Y / 2) * (Wk / 2) + X / 2) * C + 4 * c4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int arg = 0;
                float m = v[0][j];
#pragma unroll
                for (int q = 1; q < 4; ++q)
                    if (v[q][j] > m) { m = v[q][j]; arg = q; }
                if (arg == me) acc[j] += g[j];
            }
        }
        st4(dA + (i >> 4) * C + 4 * c4, acc);
    }
}

// enc_out_bwd for bf16 storage: 16-byte accesses (8 channels per thread, 32 pixels per block iteration), grid.y = sample; the same
// operations in the same order per element as the generic kernel above (bit-identical results).
__global__ void __launch_bounds__(256)
    enc_out_bwd_bf16x8_kernel(const __bf16* __restrict__ Tx, int Hfull, int s, const __bf16* __restrict__ dS,
                              const __bf16* __restrict__ dP, const __bf16* __restrict__ y, const float* __restrict__ scale,
                              const float* __restrict__ shift, int Hk, int Wk, __bf16* __restrict__ dA,
                              const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ partial) {
    // partial != null: pass 1 of the normalisation backward of THIS level's second convolution (norm_bwd_reduce: the sums of
    // g = dA * [relu alive] and g * xhat over the pixels of this workgroup) is taken here, on the dA just formed (its
    // bf16-rounded value, what a separate pass would read back) -- one launch and one read of dA and y less per encoder level
    __shared__ __attribute__((aligned(16))) float red[2][32][65];
    const int b = blockIdx.y;
    const int c8 = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float sc[8], sh[8], mu[8], rs[8], a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = (dP || partial) ? scale[b * C + 8 * c8 + j] : 0.f;
        sh[j] = (dP || partial) ? shift[b * C + 8 * c8 + j] : 0.f;
        mu[j] = partial ? mean[b * C + 8 * c8 + j] : 0.f;
        rs[j] = partial ? rstd[b * C + 8 * c8 + j] : 0.f;
        a1[j] = a2[j] = 0.f;
    }
    const int hw = Hk * Wk;
    const norm_u32x4* dSb = dS ? reinterpret_cast<const norm_u32x4*>(dS + (int64_t)b * hw * C) : nullptr;
    const norm_u32x4* Txb = Tx ? reinterpret_cast<const norm_u32x4*>(Tx + (int64_t)b * Hfull * Wk * C) : nullptr;
    const norm_u32x4* yb = reinterpret_cast<const norm_u32x4*>(y + (int64_t)b * hw * C);
    const norm_u32x4* dPb = dP ? reinterpret_cast<const norm_u32x4*>(dP + (int64_t)b * (Hk / 2) * (Wk / 2) * C) : nullptr;
    norm_u32x4* ob = reinterpret_cast<norm_u32x4*>(dA + (int64_t)b * hw * C);
    for (int p = blockIdx.x * 32 + pl; p < hw; p += gridDim.x * 32) {
        const int Y = p / Wk, X = p - Y * Wk;
        float acc[8], ymine[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        if (partial && !dPb) unpack8(yb[(int64_t)p * 8 + c8], ymine);
        if (dSb) unpack8(dSb[(int64_t)p * 8 + c8], acc);
        if (Txb) {
            int ya = s * Y - s / 2, ye = s * Y + 3 * s / 2 - 1;
            if (ya < 0) ya = 0;
            if (ye > Hfull - 1) ye = Hfull - 1;
            for (int yy = ya; yy <= ye; ++yy) {
                int y0, y1; float ly;
                bilin(yy, s, Hk, y0, y1, ly);
                const float w = (y0 == Y ? 1.f - ly : 0.f) + (y1 == Y ? ly : 0.f);
                if (w != 0.f) {
                    float t[8];
                    unpack8(Txb[((int64_t)yy * Wk + X) * 8 + c8], t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += w * t[j];
                }
            }
        }
        if (dPb) {
            const int Y0 = Y & ~1, X0 = X & ~1;
            const int64_t base = ((int64_t)Y0 * Wk + X0) * 8 + c8;
            float v[4][8], g[8];
            unpack8(yb[base], v[0]);
            unpack8(yb[base + 8], v[1]);
            unpack8(yb[base + (int64_t)Wk * 8], v[2]);
            unpack8(yb[base + (int64_t)Wk * 8 + 8], v[3]);
            unpack8(dPb[((int64_t)(Y / 2) * (Wk / 2) + X / 2) * 8 + c8], g);
            const int me = (Y & 1) * 2 + (X & 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int arg = 0;
                float m = fmaxf(v[0][j] * sc[j] + sh[j], 0.f);
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float vq = fmaxf(v[q][j] * sc[j] + sh[j], 0.f);
                    if (vq > m) { m = vq; arg = q; }
                }
                if (arg == me) acc[j] += g[j];
                if (partial) ymine[j] = me == 0 ? v[0][j] : (me == 1 ? v[1][j] : (me == 2 ? v[2][j] : v[3][j]));
            }
        }
        const norm_u32x4 packed = pack8(acc);
        ob[(int64_t)p * 8 + c8] = packed;
        if (partial) {
            float gr[8];
            unpack8(packed, gr);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = (ymine[j] * sc[j] + sh[j]) > 0.f ? gr[j] : 0.f;
                a1[j] += g;
                a2[j] += g * ((ymine[j] - mu[j]) * rs[j]);
            }
        }
    }
    if (partial) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[0][pl][8 * c8 + j] = a1[j];
            red[1][pl][8 * c8 + j] = a2[j];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int st = threadIdx.x >> 6, c = threadIdx.x & 63;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) sum += red[st][k][c];
            partial[(((int64_t)b * gridDim.x + blockIdx.x) * 2 + st) * 64 + c] = sum;
        }
    }
}

static inline int ew_grid(int64_t total_threads) {
    int64_t blocks = (total_threads + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

// ---------------------------------------------------------------------------------------------- host wrappers
int norm_finalize(const float* partial, int tiles_per_sample, int B, int64_t hw, int mode, int groups, const float* gamma,
                  const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                  float* shift, float* mean, float* rstd, hipStream_t stream) {
    const int grid = mode == 0 ? C : B * groups;
    if (diag_skip(1)) return P4C_OK;
    hipLaunchKernelGGL(norm_finalize_kernel, dim3(grid), dim3(256), 0, stream, partial, tiles_per_sample, B, hw, mode,
                       groups, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, rstd);
    P4C_CHECK_LAUNCH("norm_finalize");
    return P4C_OK;
}

int norm_eval(int B, const float* gamma, const float* beta, float eps, const float* running_mean,
              const float* running_var, float* scale, float* shift, float* mean, float* rstd, hipStream_t stream) {
    hipLaunchKernelGGL(norm_eval_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, B, gamma, beta, eps,
                       running_mean, running_var, scale, shift, mean, rstd);
    P4C_CHECK_LAUNCH("norm_eval");
    return P4C_OK;
}

int bwd_fin_max_slots() {
    if (const char* e = diag_env("P4C_BWD_INFIN_MAX")) return atoi(e);   // (read per call: A/B scripts and the parity tests switch it)
    return 256;
}

int norm_bwd_blocks(int64_t hw) {
    int64_t nblk = (hw + 16 * 8 - 1) / (16 * 8);  // >= 8 pixels per thread row
    if (nblk > 512) nblk = 512;
    if (nblk < 1) nblk = 1;
    return (int)nblk;
}

// `storage` = element type of the activation / gradient tensors in HBM (P4C_F32 or P4C_BF16); statistics,
// scale/shift and partial sums are always fp32.
template <typename T>
static int norm_bwd_t(const T* dA, const T* y, const float* scale, const float* shift, const float* mean,
                      const float* rstd, const float* gamma, int relu, int B, int64_t hw, int mode, int groups,
                      int training, float* partial, float* k1, float* k2, float* dgamma, float* dbeta, T* dY,
                      hipStream_t stream, int pre_nblk, unsigned int* fin_ticket, bool pre_finalized) {
    // pre_nblk > 0: pass 1 was taken by the kernel that produced dA (enc_out_bwd / the data-gradient convolution): `partial`
    // already holds pre_nblk slots per sample
    const int nblk = pre_nblk > 0 ? pre_nblk : norm_bwd_blocks(hw);
    const bool skip_reduce = diag_skip(4), skip_apply = diag_skip(16);
    bool skip_fin = diag_skip(2) || pre_finalized;
    if (skip_reduce || pre_nblk > 0) {
    } else if (std::is_same<T, __bf16>::value && diag_env("P4C_NORM_REDUCE_V1") == nullptr) {
        // BatchNorm with few slots: the launch's last workgroup finishes the pass (no norm_bwd_finalize launch)
        BwdFin fin{};
        if (fin_ticket && mode == 0 && (int64_t)B * nblk <= bwd_fin_max_slots()) {
            fin = BwdFin{fin_ticket, gamma, dgamma, dbeta, k1, k2, (float)B * (float)hw, B, training};
            skip_fin = true;
        }
        hipLaunchKernelGGL(norm_bwd_reduce_bf16x8_kernel, dim3(nblk, B), dim3(256), 0, stream, (const __bf16*)dA, (const __bf16*)y, scale,
                           shift, mean, rstd, relu, hw, partial, fin);
    } else
        hipLaunchKernelGGL(norm_bwd_reduce_kernel<T>, dim3(nblk, B), dim3(256), 0, stream, dA, y, scale, shift, mean, rstd,
                           relu, hw, partial);
    P4C_CHECK_LAUNCH("norm_bwd_reduce");
    if (!skip_fin && !(skip_reduce && pre_nblk <= 0))
        hipLaunchKernelGGL(norm_bwd_finalize_kernel, dim3(mode == 0 ? C : groups), dim3(256), 0, stream, partial, nblk, B, hw,
                           mode, groups, training, gamma, dgamma, dbeta, k1, k2);
    P4C_CHECK_LAUNCH("norm_bwd_finalize");
    if (skip_apply || dY == nullptr) {   // (dY == nullptr: the consumers apply pass 2 while they load dA and y -- NormBwdCoef)
    } else if (std::is_same<T, __bf16>::value && diag_env("P4C_NORM_APPLY_V1") == nullptr) {
        int64_t blocks = (hw + 127) / 128;                       // >= 4 pixel rows of 32 per workgroup
        const int64_t cap = (int64_t)num_cus() * 8 / (B > 0 ? B : 1);
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(norm_bwd_apply_bf16x8_kernel, dim3((unsigned)blocks, B), dim3(256), 0, stream, (const __bf16*)dA,
                           (const __bf16*)y, scale, shift, mean, rstd, gamma, k1, k2, relu, hw, (__bf16*)dY);
    } else {
        hipLaunchKernelGGL(norm_bwd_apply_kernel<T>, dim3(ew_grid((int64_t)B * hw * 16)), dim3(256), 0, stream, dA, y, scale,
                           shift, mean, rstd, gamma, k1, k2, relu, hw, B, dY);
    }
    P4C_CHECK_LAUNCH("norm_bwd_apply");
    return P4C_OK;
}

int norm_bwd(int storage, const void* dA, const void* y, const float* scale, const float* shift, const float* mean,
             const float* rstd, const float* gamma, int relu, int B, int64_t hw, int mode, int groups, int training,
             float* partial, float* k1, float* k2, float* dgamma, float* dbeta, void* dY, hipStream_t stream, int pre_nblk,
             unsigned int* fin_ticket, bool pre_finalized) {
    if (storage == P4C_BF16)
        return norm_bwd_t<__bf16>((const __bf16*)dA, (const __bf16*)y, scale, shift, mean, rstd, gamma, relu, B, hw, mode,
                                  groups, training, partial, k1, k2, dgamma, dbeta, (__bf16*)dY, stream, pre_nblk, fin_ticket, pre_finalized);
    return norm_bwd_t<float>((const float*)dA, (const float*)y, scale, shift, mean, rstd, gamma, relu, B, hw, mode, groups,
                             training, partial, k1, k2, dgamma, dbeta, (float*)dY, stream, pre_nblk, fin_ticket, pre_finalized);
}

// pool_fwd for bf16 storage: 16-byte accesses (8 channels per thread), the sample's scale / shift in registers (grid.y = sample),
// two output pixels in flight, 32-bit index arithmetic -- the same fp32 arithmetic per element as pool_fwd_kernel
__global__ void __launch_bounds__(256)
    pool_fwd_bf16x8_kernel(const __bf16* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift, int H, int W,
                           __bf16* __restrict__ P) {
    const int b = blockIdx.y;
    const int c8 = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = scale[b * C + 8 * c8 + j]; sh[j] = shift[b * C + 8 * c8 + j]; }
    const int Ho = H / 2, Wo = W / 2, npx = Ho * Wo;
    const norm_u32x4* yb = reinterpret_cast<const norm_u32x4*>(y + (int64_t)b * H * W * C);
    norm_u32x4* pb = reinterpret_cast<norm_u32x4*>(P + (int64_t)b * npx * C);
    const int stride = gridDim.x * 32;
    for (int p0 = blockIdx.x * 32 + pl; p0 < npx; p0 += 2 * stride) {
        norm_u32x4 v[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p0 + u * stride;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = norm_u32x4{0u, 0u, 0u, 0u};
            if (p < npx) {
                const int Y = p / Wo, X = p - Y * Wo;
                const int64_t base = ((int64_t)(2 * Y) * W + 2 * X) * 8 + c8;
                v[u][0] = yb[base];
                v[u][1] = yb[base + 8];
                v[u][2] = yb[base + (int64_t)W * 8];
                v[u][3] = yb[base + (int64_t)W * 8 + 8];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int p = p0 + u * stride;
            if (p >= npx) continue;
            float m[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float f[8];
                unpack8(v[u][k], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = fmaxf(f[j] * sc[j] + sh[j], 0.f);
                    m[j] = k == 0 ? a : fmaxf(m[j], a);
                }
            }
            pb[(int64_t)p * 8 + c8] = pack8(m);
        }
    }
}

template <typename T>
static int pool_fwd_t(const T* y, const float* scale, const float* shift, int B, int H, int W, T* P, hipStream_t stream) {
    if (std::is_same<T, __bf16>::value && diag_env("P4C_POOL_V1") == nullptr && (int64_t)(H / 2) * (W / 2) < ((int64_t)1 << 30)) {
        int64_t blocks = ((int64_t)(H / 2) * (W / 2) + 63) / 64;   // two pixel rows of 32 per workgroup and trip
        const int64_t cap = (int64_t)num_cus() * 8 / (B > 0 ? B : 1) + 1;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(pool_fwd_bf16x8_kernel, dim3((unsigned)blocks, B), dim3(256), 0, stream, (const __bf16*)y, scale, shift, H, W,
                           (__bf16*)P);
        P4C_CHECK_LAUNCH("pool_fwd");
        return P4C_OK;
    }
    hipLaunchKernelGGL(pool_fwd_kernel<T>, dim3(ew_grid((int64_t)B * (H / 2) * (W / 2) * 16)), dim3(256), 0, stream, y,
                       scale, shift, B, H, W, P);
    P4C_CHECK_LAUNCH("pool_fwd");
    return P4C_OK;
}
int pool_fwd(int storage, const void* y, const float* scale, const float* shift, int B, int H, int W, void* P,
             hipStream_t stream) {
    return storage == P4C_BF16 ? pool_fwd_t<__bf16>((const __bf16*)y, scale, shift, B, H, W, (__bf16*)P, stream)
                               : pool_fwd_t<float>((const float*)y, scale, shift, B, H, W, (float*)P, stream);
}

template <typename T>
static int upsum_fwd_t(const void* const* y, const float* const* scale, const float* const* shift, int B, int H, int W,
                       T* S, hipStream_t stream) {
    UpLevels<T> lv;
    for (int k = 0; k < 5; ++k) { lv.y[k] = (const T*)y[k]; lv.scale[k] = scale[k]; lv.shift[k] = shift[k]; }
    hipLaunchKernelGGL(upsum_fwd_kernel<T>, dim3(ew_grid((int64_t)B * H * (W / 16) * 16)), dim3(256), 0, stream, lv, B, H, W, S);
    P4C_CHECK_LAUNCH("upsum_fwd");
    return P4C_OK;
}
int upsum_fwd(int storage, const void* const* y, const float* const* scale, const float* const* shift, int B, int H, int W,
              void* S, hipStream_t stream) {
    return storage == P4C_BF16 ? upsum_fwd_t<__bf16>(y, scale, shift, B, H, W, (__bf16*)S, stream)
                               : upsum_fwd_t<float>(y, scale, shift, B, H, W, (float*)S, stream);
}

template <typename T>
static int up_bwd_x4_t(const T* dS, int B, int H, int W, void* const* tx, hipStream_t stream) {
    UpBwdOut<T> o;
    for (int k = 0; k < 4; ++k) o.tx[k] = (T*)tx[k];
    // bf16 rows in whole 64-pixel strips: the banded interpolation matrix on the matrix cores (upbwd_mfma.hip)
    if (std::is_same<T, __bf16>::value && up_bwd_x4_mfma_ok(B, H, W) && diag_env("P4C_UPBWD_VALU") == nullptr)
        return launch_up_bwd_x4_mfma((const void*)dS, (int64_t)B * H, W, tx, stream);
    hipLaunchKernelGGL(up_bwd_x4_kernel<T>, dim3(B * H * ((W + 63) / 64)), dim3(256), 0, stream, dS, H, W, o);
    P4C_CHECK_LAUNCH("up_bwd_x4");
    return P4C_OK;
}
int up_bwd_x4(int storage, const void* dS, int B, int H, int W, void* const* tx, hipStream_t stream) {
    return storage == P4C_BF16 ? up_bwd_x4_t<__bf16>((const __bf16*)dS, B, H, W, tx, stream)
                               : up_bwd_x4_t<float>((const float*)dS, B, H, W, tx, stream);
}

// ---- round 3: the same arithmetic (same order per element, bit-identical dA) with the memory latency taken out ------------
// enc_out_bwd_px: one thread per (pixel, channel octet) like enc_out_bwd_bf16x8_kernel, but the up-sampling adjoint's rows of Tx are
// loaded EIGHT AT A TIME without a branch (weights that are zero are skipped by a select): at level 4 the old loop walked 32
// dependent loads per pixel (18-20 us per launch for 4 MB of input).
// enc_out_bwd_blk: one thread per (2x2 pixel block, channel octet) for the fine levels (stride <= 2), where the kernel was bound by
// vector instructions: the arg-max of a pooling window is evaluated once instead of once per pixel of the window, each y value is
// loaded once, and a row of Tx serves both pixel rows of the block.
template <int CHUNK>
__device__ __forceinline__ void up_adjoint_rows(const norm_u32x4* __restrict__ Txb, int Hfull, int Wk, int Hk, int s, int X, int c8, int Y,
                                                float (&acc)[8]) {
    int ya = s * Y - s / 2, ye = s * Y + 3 * s / 2 - 1;
    if (ya < 0) ya = 0;
    if (ye > Hfull - 1) ye = Hfull - 1;
    for (int y0c = ya; y0c <= ye; y0c += CHUNK) {
        norm_u32x4 t[CHUNK];
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            const int yy = y0c + u <= ye ? y0c + u : ye;
            t[u] = Txb[((int64_t)yy * Wk + X) * 8 + c8];
        }
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
            const int yy = y0c + u;
            int i0, i1; float ly;
            bilin(yy <= ye ? yy : ye, s, Hk, i0, i1, ly);
            float w = (i0 == Y ? 1.f - ly : 0.f) + (i1 == Y ? ly : 0.f);
            if (yy > ye) w = 0.f;
            float f[8];
            unpack8(t[u], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = w != 0.f ? acc[j] + w * f[j] : acc[j];
        }
    }
}

__global__ void __launch_bounds__(256)
    enc_out_bwd_px_kernel(const __bf16* __restrict__ Tx, int Hfull, int s, const __bf16* __restrict__ dS,
                          const __bf16* __restrict__ dP, const __bf16* __restrict__ y, const float* __restrict__ scale,
                          const float* __restrict__ shift, int Hk, int Wk, __bf16* __restrict__ dA,
                          const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ partial, BwdFin fin) {
    __shared__ __attribute__((aligned(16))) float red[2][32][65];
    const int b = blockIdx.y;
    const int c8 = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float sc[8], sh[8], mu[8], rs[8], a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = (dP || partial) ? scale[b * C + 8 * c8 + j] : 0.f;
        sh[j] = (dP || partial) ? shift[b * C + 8 * c8 + j] : 0.f;
        mu[j] = partial ? mean[b * C + 8 * c8 + j] : 0.f;
        rs[j] = partial ? rstd[b * C + 8 * c8 + j] : 0.f;
        a1[j] = a2[j] = 0.f;
    }
    const int hw = Hk * Wk;
    const norm_u32x4* dSb = dS ? reinterpret_cast<const norm_u32x4*>(dS + (int64_t)b * hw * C) : nullptr;
    const norm_u32x4* Txb = Tx ? reinterpret_cast<const norm_u32x4*>(Tx + (int64_t)b * Hfull * Wk * C) : nullptr;
    const norm_u32x4* yb = reinterpret_cast<const norm_u32x4*>(y + (int64_t)b * hw * C);
    const norm_u32x4* dPb = dP ? reinterpret_cast<const norm_u32x4*>(dP + (int64_t)b * (Hk / 2) * (Wk / 2) * C) : nullptr;
    norm_u32x4* ob = reinterpret_cast<norm_u32x4*>(dA + (int64_t)b * hw * C);
    for (int p = blockIdx.x * 32 + pl; p < hw; p += gridDim.x * 32) {
        const int Y = p / Wk, X = p - Y * Wk;
        float acc[8], ymine[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        // every load that does not depend on the adjoint's rows goes out first
        norm_u32x4 v4[4], gq = norm_u32x4{0u, 0u, 0u, 0u}, ym = norm_u32x4{0u, 0u, 0u, 0u};
        if (dPb) {
            const int64_t base = ((int64_t)(Y & ~1) * Wk + (X & ~1)) * 8 + c8;
            v4[0] = yb[base]; v4[1] = yb[base + 8]; v4[2] = yb[base + (int64_t)Wk * 8]; v4[3] = yb[base + (int64_t)Wk * 8 + 8];
            gq = dPb[((int64_t)(Y / 2) * (Wk / 2) + X / 2) * 8 + c8];
        } else if (partial) {
            ym = yb[(int64_t)p * 8 + c8];
        }
        if (dSb) unpack8(dSb[(int64_t)p * 8 + c8], acc);
        if (Txb) up_adjoint_rows<8>(Txb, Hfull, Wk, Hk, s, X, c8, Y, acc);
        if (partial && !dPb) unpack8(ym, ymine);
        if (dPb) {
            float v[4][8], g[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) unpack8(v4[q], v[q]);
            unpack8(gq, g);
            const int me = (Y & 1) * 2 + (X & 1);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int arg = 0;
                float m = fmaxf(v[0][j] * sc[j] + sh[j], 0.f);
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    const float vq = fmaxf(v[q][j] * sc[j] + sh[j], 0.f);
                    if (vq > m) { m = vq; arg = q; }
                }
                if (arg == me) acc[j] += g[j];
                if (partial) ymine[j] = me == 0 ? v[0][j] : (me == 1 ? v[1][j] : (me == 2 ? v[2][j] : v[3][j]));
            }
        }
        const norm_u32x4 packed = pack8(acc);
        ob[(int64_t)p * 8 + c8] = packed;
        if (partial) {
            float gr[8];
            unpack8(packed, gr);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = (ymine[j] * sc[j] + sh[j]) > 0.f ? gr[j] : 0.f;
                a1[j] += g;
                a2[j] += g * ((ymine[j] - mu[j]) * rs[j]);
            }
        }
    }
    if (partial) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[0][pl][8 * c8 + j] = a1[j];
            red[1][pl][8 * c8 + j] = a2[j];
        }
        bwd_slot_finish(red, partial, fin);
    }
}

__global__ void __launch_bounds__(256)
    enc_out_bwd_blk_kernel(const __bf16* __restrict__ Tx, int Hfull, int s, const __bf16* __restrict__ dS,
                           const __bf16* __restrict__ dP, const __bf16* __restrict__ y, const float* __restrict__ scale,
                           const float* __restrict__ shift, int Hk, int Wk, __bf16* __restrict__ dA,
                           const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ partial, BwdFin fin) {
    // requires dP (a pooled level below) and Hk, Wk even: levels 0 and 1 of the plan
    __shared__ __attribute__((aligned(16))) float red[2][32][65];
    const int b = blockIdx.y;
    const int c8 = threadIdx.x & 7, pl = threadIdx.x >> 3;
    float sc[8], sh[8], mu[8], rs[8], a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale[b * C + 8 * c8 + j];
        sh[j] = shift[b * C + 8 * c8 + j];
        mu[j] = partial ? mean[b * C + 8 * c8 + j] : 0.f;
        rs[j] = partial ? rstd[b * C + 8 * c8 + j] : 0.f;
        a1[j] = a2[j] = 0.f;
    }
    const int hw = Hk * Wk, Wb = Wk / 2, nblk = (Hk / 2) * Wb;
    const norm_u32x4* dSb = dS ? reinterpret_cast<const norm_u32x4*>(dS + (int64_t)b * hw * C) : nullptr;
    const norm_u32x4* Txb = Tx ? reinterpret_cast<const norm_u32x4*>(Tx + (int64_t)b * Hfull * Wk * C) : nullptr;
    const norm_u32x4* yb = reinterpret_cast<const norm_u32x4*>(y + (int64_t)b * hw * C);
    const norm_u32x4* dPb = reinterpret_cast<const norm_u32x4*>(dP + (int64_t)b * nblk * C);
    norm_u32x4* ob = reinterpret_cast<norm_u32x4*>(dA + (int64_t)b * hw * C);
    for (int blk = blockIdx.x * 32 + pl; blk < nblk; blk += gridDim.x * 32) {
        const int by = blk / Wb, bx = blk - by * Wb;
        const int64_t base = ((int64_t)(2 * by) * Wk + 2 * bx) * 8 + c8;   // pixel q = 2 dy + dx: base + dy * Wk * 8 + dx * 8
        norm_u32x4 v4[4], d4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v4[q] = yb[base + (int64_t)(q >> 1) * Wk * 8 + (q & 1) * 8];
            d4[q] = dSb ? dSb[base + (int64_t)(q >> 1) * Wk * 8 + (q & 1) * 8] : norm_u32x4{0u, 0u, 0u, 0u};
        }
        const norm_u32x4 gq = dPb[(int64_t)blk * 8 + c8];
        float acc[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q) unpack8(d4[q], acc[q]);
        if (Txb) {
#pragma unroll
            for (int q = 0; q < 4; ++q) up_adjoint_rows<4>(Txb, Hfull, Wk, Hk, s, 2 * bx + (q & 1), c8, 2 * by + (q >> 1), acc[q]);
        }
        float v[4][8], g[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) unpack8(v4[q], v[q]);
        unpack8(gq, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int arg = 0;
            float m = fmaxf(v[0][j] * sc[j] + sh[j], 0.f);
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const float vq = fmaxf(v[q][j] * sc[j] + sh[j], 0.f);
                if (vq > m) { m = vq; arg = q; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][j] = arg == q ? acc[q][j] + g[j] : acc[q][j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const norm_u32x4 packed = pack8(acc[q]);
            ob[base + (int64_t)(q >> 1) * Wk * 8 + (q & 1) * 8] = packed;
            if (partial) {
                float gr[8];
                unpack8(packed, gr);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gg = (v[q][j] * sc[j] + sh[j]) > 0.f ? gr[j] : 0.f;
                    a1[j] += gg;
                    a2[j] += gg * ((v[q][j] - mu[j]) * rs[j]);
                }
            }
        }
    }
    if (partial) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[0][pl][8 * c8 + j] = a1[j];
            red[1][pl][8 * c8 + j] = a2[j];
        }
        bwd_slot_finish(red, partial, fin);
    }
}

template <typename T>
static int enc_out_bwd_t(const T* Tx, int Hfull, int s, const T* dS, const T* dP, const T* y, const float* scale,
                         const float* shift, int B, int Hk, int Wk, T* dA, const float* mean, const float* rstd, float* partial,
                         int* nblk_out, hipStream_t stream, const BwdFin* finp, bool* finalized_out) {
    if (nblk_out) *nblk_out = 0;
    if (finalized_out) *finalized_out = false;
    if (std::is_same<T, __bf16>::value && diag_env("P4C_ENC_OUT_V1") == nullptr) {
        const char* v2 = diag_env("P4C_ENC_OUT_V2");   // 0: the round-2 kernel (one thread per pixel, dependent row loads)
        const bool old = v2 && v2[0] == '0';
        const bool blk = !old && dP && s <= 2 && Hk % 2 == 0 && Wk % 2 == 0;   // one thread per 2x2 block on the fine levels
        int64_t blocks = blk ? ((int64_t)(Hk / 2) * (Wk / 2) + 31) / 32 : ((int64_t)Hk * Wk + 31) / 32;
        int64_t cap = (int64_t)num_cus() * 8 / (B > 0 ? B : 1);
        const bool fuse = partial && nblk_out && mean && rstd;
        if (fuse && cap > NORM_BWD_MAX_BLOCKS) cap = NORM_BWD_MAX_BLOCKS;   // one partial slot per workgroup
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        // few slots (the coarse levels): the last workgroup also finishes the pass (BwdFin) -- no norm_bwd_finalize launch
        const bool infin = fuse && !old && finp && finp->ticket && finalized_out && blocks * B <= bwd_fin_max_slots();
        const BwdFin fin = infin ? *finp : BwdFin{};
        if (old)
            hipLaunchKernelGGL(enc_out_bwd_bf16x8_kernel, dim3((unsigned)blocks, B), dim3(256), 0, stream, (const __bf16*)Tx, Hfull, s,
                               (const __bf16*)dS, (const __bf16*)dP, (const __bf16*)y, scale, shift, Hk, Wk, (__bf16*)dA, mean, rstd,
                               fuse ? partial : nullptr);
        else
            hipLaunchKernelGGL(blk ? enc_out_bwd_blk_kernel : enc_out_bwd_px_kernel, dim3((unsigned)blocks, B), dim3(256), 0, stream,
                               (const __bf16*)Tx, Hfull, s, (const __bf16*)dS, (const __bf16*)dP, (const __bf16*)y, scale, shift, Hk, Wk,
                               (__bf16*)dA, mean, rstd, fuse ? partial : nullptr, fin);
        if (fuse) *nblk_out = (int)blocks;
        if (infin) *finalized_out = true;
    } else {
        hipLaunchKernelGGL(enc_out_bwd_kernel<T>, dim3(ew_grid((int64_t)B * Hk * Wk * 16)), dim3(256), 0, stream, Tx, Hfull, s,
                           dS, dP, y, scale, shift, B, Hk, Wk, dA);
    }
    P4C_CHECK_LAUNCH("enc_out_bwd");
    return P4C_OK;
}
int enc_out_bwd(int storage, const void* Tx, int Hfull, int s, const void* dS, const void* dP, const void* y,
                const float* scale, const float* shift, int B, int Hk, int Wk, void* dA, hipStream_t stream, const float* mean,
                const float* rstd, float* partial, int* nblk_out, const BwdFin* fin, bool* finalized_out) {
    if (storage == P4C_BF16)
        return enc_out_bwd_t<__bf16>((const __bf16*)Tx, Hfull, s, (const __bf16*)dS, (const __bf16*)dP, (const __bf16*)y, scale,
                                     shift, B, Hk, Wk, (__bf16*)dA, mean, rstd, partial, nblk_out, stream, fin, finalized_out);
    return enc_out_bwd_t<float>((const float*)Tx, Hfull, s, (const float*)dS, (const float*)dP, (const float*)y, scale, shift,
                                B, Hk, Wk, (float*)dA, mean, rstd, partial, nblk_out, stream, fin, finalized_out);
}

}  // namespace p4c
