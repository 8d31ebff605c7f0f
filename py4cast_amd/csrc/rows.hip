// Row-wise passes of the node / edge MLPs of the mesh-GNN models (GraphLAM / HiLAM: every MLP is Linear - SiLU - Linear -
// LayerNorm on rows of 64 features, config/CLI/model/graphlam.yaml:19-26 hidden_dims 64 hidden_layers 1; the networks come from
// mfai, py4cast/models.py:10-20).  With 0.5 M grid nodes and 1-2 M edges per sample every such pass is a stream of 128-byte rows:
// HBM-bound, and what the library path (torch LayerNorm + hipBLASLt) does badly -- a rocprofv3 trace of the library-only GraphLAM
// step on MI355X spends 43 % in the weight-gradient GEMMs (64 x 64 outputs, K = 2 M rows: 600 us per call) and 30 % in
// LayerNorm forward / backward (fp32, 300-430 us per call).
//   * row_layernorm_fwd / bwd : LayerNorm over the C features of a row (+ optional residual), one 16-byte vector per lane, row
//     statistics by butterflies over the lanes of the row, gamma / beta gradients accumulated in registers over a persistent
//     wave's rows and reduced in a fixed order (no atomics).  Statistics are recomputed in the backward: nothing is saved.
//   * row_linear_wgrad        : dW[o][k] = sum_r dY[r][o] X[r][k], db[o] = sum_r dY[r][o] for R >> 64: the reduction index is
//     the ROW, so both MFMA operands are transposed reads (ds_read_b64_tr_b16) of the [row][feature] tiles staged in LDS;
//     persistent waves keep the 64 x K accumulator in registers, per-workgroup partials are reduced in a fixed order.
#include "kernels.hpp"

namespace p4c {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        return u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
    }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(v[i] << 16);
            f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ unsigned int rne(float x) {
        unsigned int u = __float_as_uint(x);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
        return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = rne(f[2 * i]) | (rne(f[2 * i + 1]) << 16);
        return v;
    }
};

// sum over the lpr lanes of a row (lpr = power of two, rows aligned to lpr lanes)
__device__ __forceinline__ float row_sum(float v, int lpr) {
    for (int o = 1; o < lpr; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// rows = tokens of (B, Hp, Wp) maps of which only [0,H) x [0,W) are real (Swin pads its maps to multiples of the window): Hp = 0 -> no mask
struct RowMask {
    int Hp, Wp, H, W;
    __device__ __forceinline__ bool masked(int64_t r) const {
        if (Hp == 0) return false;
        const unsigned rem = (unsigned)(r % ((int64_t)Hp * Wp));
        const unsigned y = rem / (unsigned)Wp, x = rem - y * (unsigned)Wp;
        return y >= (unsigned)H || x >= (unsigned)W;
    }
};

// ---------------------------------------------------------------- LayerNorm forward: out = LN(x) * gamma + beta (+ res)
template <typename T>
__global__ void __launch_bounds__(256)
    row_layernorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gamma,
                             const float* __restrict__ beta, float eps, T* __restrict__ out, int64_t R, int C, int lpr_log2,
                             RowMask mk) {
    constexpr int NV = Vec16<T>::N;
    const int lane = threadIdx.x & 63, lpr = 1 << lpr_log2, rpw = 64 >> lpr_log2;
    const int chunk = lane & (lpr - 1), sub = lane >> lpr_log2;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int chunks = C / NV;
    const bool cok = chunk < chunks;
    float g[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = cok ? gamma[chunk * NV + i] : 0.f;
        b[i] = cok ? beta[chunk * NV + i] : 0.f;
    }
    const float inv_c = 1.f / (float)C;
    for (int64_t r0 = wave * rpw * 2; r0 < R; r0 += nwaves * rpw * 2) {
        u32x4 vx[2], vr[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t r = r0 + u * rpw + sub;
            ok[u] = r < R && cok;
            vx[u] = vr[u] = u32x4{0, 0, 0, 0};
            if (ok[u]) {
                vx[u] = reinterpret_cast<const u32x4*>(x)[r * chunks + chunk];
                if (res) vr[u] = reinterpret_cast<const u32x4*>(res)[r * chunks + chunk];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float f[NV], fr[NV];
            Vec16<T>::unpack(vx[u], f);
            Vec16<T>::unpack(vr[u], fr);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) s += f[i];
            const float mean = row_sum(s, lpr) * inv_c;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f[i] = cok ? f[i] - mean : 0.f;
                q += f[i] * f[i];
            }
            const float rstd = rsqrtf(row_sum(q, lpr) * inv_c + eps);
            const bool pad = mk.masked(r0 + u * rpw + sub);   // a padding token: the row is ZERO (F.pad of the normalised map)
#pragma unroll
            for (int i = 0; i < NV; ++i) f[i] = pad ? 0.f : f[i] * rstd * g[i] + b[i] + fr[i];
            if (ok[u]) reinterpret_cast<u32x4*>(out)[(r0 + u * rpw + sub) * chunks + chunk] = Vec16<T>::pack(f);
        }
    }
}

// ---------------------------------------------------------------- LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dgamma = sum_r dy * xhat,  dbeta = sum_r dy
// partial: [workgroup][2][C] floats (dgamma, dbeta of the rows the workgroup processed), reduced by row_param_reduce_kernel.
template <typename T>
__global__ void __launch_bounds__(256)
    row_layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ gamma, float eps,
                             T* __restrict__ dx, float* __restrict__ partial, int64_t R, int C, int lpr_log2, RowMask mk) {
    constexpr int NV = Vec16<T>::N;
    const int lane = threadIdx.x & 63, lpr = 1 << lpr_log2, rpw = 64 >> lpr_log2;
    const int chunk = lane & (lpr - 1), sub = lane >> lpr_log2;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int chunks = C / NV;
    const bool cok = chunk < chunks;
    float g[NV], dg[NV], db[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = cok ? gamma[chunk * NV + i] : 0.f;
        dg[i] = db[i] = 0.f;
    }
    const float inv_c = 1.f / (float)C;
    for (int64_t r0 = wave * rpw * 2; r0 < R; r0 += nwaves * rpw * 2) {
        u32x4 vx[2], vd[2];
        bool ok[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t r = r0 + u * rpw + sub;
            ok[u] = r < R && cok;
            vx[u] = vd[u] = u32x4{0, 0, 0, 0};
            if (ok[u]) {
                vx[u] = reinterpret_cast<const u32x4*>(x)[r * chunks + chunk];
                if (!mk.masked(r)) vd[u] = reinterpret_cast<const u32x4*>(dy)[r * chunks + chunk];   // (padding token: no gradient)
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float f[NV], d[NV];
            Vec16<T>::unpack(vx[u], f);
            Vec16<T>::unpack(vd[u], d);
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) s += f[i];
            const float mean = row_sum(s, lpr) * inv_c;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f[i] = cok ? f[i] - mean : 0.f;
                q += f[i] * f[i];
            }
            const float rstd = rsqrtf(row_sum(q, lpr) * inv_c + eps);
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f[i] *= rstd;                 // xhat
                dg[i] += d[i] * f[i];
                db[i] += d[i];
                d[i] *= g[i];                 // g
                m1 += d[i];
                m2 += d[i] * f[i];
            }
            m1 = row_sum(m1, lpr) * inv_c;
            m2 = row_sum(m2, lpr) * inv_c;
#pragma unroll
            for (int i = 0; i < NV; ++i) d[i] = rstd * (d[i] - m1 - f[i] * m2);
            if (ok[u]) reinterpret_cast<u32x4*>(dx)[(r0 + u * rpw + sub) * chunks + chunk] = Vec16<T>::pack(d);
        }
    }
    // the rows of one wave: lanes with the same feature chunk; then the four waves of the workgroup in wave order (fixed order)
    for (int o = lpr; o < 64; o <<= 1)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            dg[i] += __shfl_xor(dg[i], o, 64);
            db[i] += __shfl_xor(db[i], o, 64);
        }
    __shared__ float wred[4][2 * 512];   // C <= 512 (1 KiB rows: 256 fp32 or 512 bf16 features)
    const int wv = threadIdx.x >> 6;
    if (sub == 0 && cok) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            wred[wv][chunk * NV + i] = dg[i];
            wred[wv][C + chunk * NV + i] = db[i];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * C; j += blockDim.x)
        partial[(int64_t)blockIdx.x * 2 * C + j] = ((wred[0][j] + wred[1][j]) + wred[2][j]) + wred[3][j];
}

// ---------------------------------------------------------------- (x + add) -> LayerNorm, rows up to 2 KiB
// t = x + add[r % add_rows] (add == NULL: t = x), sum_out = t (optional), out = LN(t) * gamma + beta.  One row per wave (64 lanes x NCH
// 16-byte chunks): UNETR++'s token rows (128 ... 1024 bf16 features) with the positional embedding (one (N, C) table for every sample)
// added on the way in -- the block's residual t and its normalised copy from ONE read of x.
template <typename T, int NCH>
__global__ void __launch_bounds__(256)
    row_add_layernorm_fwd_kernel(const T* __restrict__ x, const T* __restrict__ add, int64_t add_rows, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float eps, T* __restrict__ sum_out, T* __restrict__ out, int64_t R, int C) {
    constexpr int NV = Vec16<T>::N;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int chunks = C / NV;
    const float inv_c = 1.f / (float)C;
    for (int64_t r = wave; r < R; r += nwaves) {
        float f[NCH][NV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + 64 * k;
            const bool ok = ch < chunks;
            u32x4 v = u32x4{0, 0, 0, 0};
            if (ok) v = reinterpret_cast<const u32x4*>(x)[r * chunks + ch];
            Vec16<T>::unpack(v, f[k]);
            if (add && ok) {
                float a[NV];
                Vec16<T>::unpack(reinterpret_cast<const u32x4*>(add)[(r % add_rows) * chunks + ch], a);
#pragma unroll
                for (int i = 0; i < NV; ++i) f[k][i] += a[i];
                const u32x4 tv = Vec16<T>::pack(f[k]);
                Vec16<T>::unpack(tv, f[k]);          // the statistics are those of the STORED sum (what the backward re-reads)
                if (sum_out) reinterpret_cast<u32x4*>(sum_out)[r * chunks + ch] = tv;
            } else if (sum_out && ok) {
                reinterpret_cast<u32x4*>(sum_out)[r * chunks + ch] = v;     // add == NULL: t = x, written all the same (the header's contract)
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) s += f[k][i];
        }
        const float mean = row_sum(s, 64) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const bool ok = lane + 64 * k < chunks;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f[k][i] = ok ? f[k][i] - mean : 0.f;
                q += f[k][i] * f[k][i];
            }
        }
        const float rstd = rsqrtf(row_sum(q, 64) * inv_c + eps);
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + 64 * k;
            if (ch < chunks) {
#pragma unroll
                for (int i = 0; i < NV; ++i) f[k][i] = f[k][i] * rstd * gamma[ch * NV + i] + beta[ch * NV + i];
                reinterpret_cast<u32x4*>(out)[r * chunks + ch] = Vec16<T>::pack(f[k]);
            }
        }
    }
}

// dt = LN_backward(dy; t) (+ extra: the gradient that reaches t from its other consumers), partial dgamma / dbeta per workgroup
template <typename T, int NCH>
__global__ void __launch_bounds__(256)
    row_add_layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ t, const T* __restrict__ extra, const float* __restrict__ gamma,
                                 float eps, T* __restrict__ dt, float* __restrict__ partial, int64_t R, int C) {
    constexpr int NV = Vec16<T>::N;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    const int chunks = C / NV;
    const float inv_c = 1.f / (float)C;
    float g[NCH][NV], dg[NCH][NV], db[NCH][NV];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            g[k][i] = lane + 64 * k < chunks ? gamma[(lane + 64 * k) * NV + i] : 0.f;
            dg[k][i] = db[k][i] = 0.f;
        }
    for (int64_t r = wave; r < R; r += nwaves) {
        float f[NCH][NV], d[NCH][NV];
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + 64 * k;
            u32x4 vx = u32x4{0, 0, 0, 0}, vd = vx;
            if (ch < chunks) {
                vx = reinterpret_cast<const u32x4*>(t)[r * chunks + ch];
                vd = reinterpret_cast<const u32x4*>(dy)[r * chunks + ch];
            }
            Vec16<T>::unpack(vx, f[k]);
            Vec16<T>::unpack(vd, d[k]);
#pragma unroll
            for (int i = 0; i < NV; ++i) s += f[k][i];
        }
        const float mean = row_sum(s, 64) * inv_c;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const bool ok = lane + 64 * k < chunks;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f[k][i] = ok ? f[k][i] - mean : 0.f;
                q += f[k][i] * f[k][i];
            }
        }
        const float rstd = rsqrtf(row_sum(q, 64) * inv_c + eps);
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f[k][i] *= rstd;
                dg[k][i] += d[k][i] * f[k][i];
                db[k][i] += d[k][i];
                d[k][i] *= g[k][i];
                m1 += d[k][i];
                m2 += d[k][i] * f[k][i];
            }
        m1 = row_sum(m1, 64) * inv_c;
        m2 = row_sum(m2, 64) * inv_c;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            const int ch = lane + 64 * k;
            if (ch < chunks) {
                float e[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) e[i] = 0.f;
                if (extra) Vec16<T>::unpack(reinterpret_cast<const u32x4*>(extra)[r * chunks + ch], e);
#pragma unroll
                for (int i = 0; i < NV; ++i) d[k][i] = rstd * (d[k][i] - m1 - f[k][i] * m2) + e[i];
                reinterpret_cast<u32x4*>(dt)[r * chunks + ch] = Vec16<T>::pack(d[k]);
            }
        }
    }
    __shared__ float wred[4][2 * 1024];   // C <= 1024 (2 KiB bf16 rows) / 512 (fp32)
    const int Cp = C;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int ch = lane + 64 * k;
        if (ch < chunks) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                wred[wv][ch * NV + i] = dg[k][i];
                wred[wv][Cp + ch * NV + i] = db[k][i];
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 2 * C; j += blockDim.x)
        partial[(int64_t)blockIdx.x * 2 * C + j] = ((wred[0][j] + wred[1][j]) + wred[2][j]) + wred[3][j];
}

// out[j] = sum_s partial[s][j], j < n, in a fixed order: a block owns 32 outputs, its 8 thread rows take the slots s = sg (mod 8)
// with four independent partial sums each, then the 8 rows are added in order through LDS.
__global__ void __launch_bounds__(256) row_param_reduce_kernel(const float* __restrict__ partial, int slots, int n, float* __restrict__ out) {
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
        for (int k = 1; k < 8; ++k) t += red[k][jj];
        out[j] = t;
    }
}

// The same sum for n % 4 == 0 with more loads in flight (round 6): with 2 048 slots (the add + LayerNorm backward of 32 768 token rows)
// a thread of the kernel above walks 256 slots in 64 dependent rounds -- 15 us for 2 MB of L2-resident partials, 126 times per UNETR++
// step.  Here a block owns 32 outputs as 8 quads x 32 slot lanes, a lane takes the slots s = lane (mod 32) eight 16-byte loads at a
// time, the 32 lanes' sums are added in lane order through LDS (fixed order: bit-identical reruns).
__global__ void __launch_bounds__(256) row_param_reduce4_kernel(const float* __restrict__ partial, int slots, int n, float* __restrict__ out) {
    __shared__ p4c_f32x4 red[32][8];
    const int q = threadIdx.x & 7, sl = threadIdx.x >> 3;
    const int j = blockIdx.x * 32 + 4 * q;
    p4c_f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    if (j < n) {
        const float* p = partial + j;
        int s = sl;
        for (; s + 7 * 32 < slots; s += 8 * 32) {
            p4c_f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const p4c_f32x4*>(p + (int64_t)(s + 32 * u) * n);
#pragma unroll
            for (int u = 0; u < 8; u += 2)
#pragma unroll
                for (int k = 0; k < 4; ++k) { a0[k] += v[u][k]; a1[k] += v[u + 1][k]; }
        }
        for (; s < slots; s += 32) {
            const p4c_f32x4 v = *reinterpret_cast<const p4c_f32x4*>(p + (int64_t)s * n);
#pragma unroll
            for (int k = 0; k < 4; ++k) a0[k] += v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) a0[k] += a1[k];
    red[sl][q] = a0;
    __syncthreads();
    if (sl == 0 && j < n) {
        p4c_f32x4 t = red[0][q];
        for (int r = 1; r < 32; ++r) {
            const p4c_f32x4 v = red[r][q];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += v[k];
        }
        *reinterpret_cast<p4c_f32x4*>(out + j) = t;
    }
}

static void launch_row_param_reduce(const float* partial, int slots, int n, float* out, hipStream_t s) {
    if (n % 4 == 0 && slots >= 64 && (reinterpret_cast<uintptr_t>(partial) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0)
        hipLaunchKernelGGL(row_param_reduce4_kernel, dim3((n + 31) / 32), dim3(256), 0, s, partial, slots, n, out);
    else
        hipLaunchKernelGGL(row_param_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, s, partial, slots, n, out);
}

// ---------------------------------------------------------------- tall-skinny weight gradient
// dW[o][k] = sum_r dY[r][o] X[r][k]  (O = 64, K = 16 * KS16 <= 128), db[o] = sum_r dY[r][o];  bf16 rows.
// A wave stages tiles of 64 rows of dY ([row][64]) and X ([row][K]) into its own LDS images (16-byte coalesced loads, rows
// padded by 16 B so that the transposed reads spread over the banks), then A = dY^T and B = X are both ds_read_b64_tr_b16
// operands: lane (channel, h) gets 8 consecutive rows of its channel.
constexpr int WG_ROWS = 64;
template <int KS16>
__global__ void __launch_bounds__(256, 2)
    row_linear_wgrad_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, float* __restrict__ partial, int64_t R) {
    constexpr int K = 16 * KS16, NT = (K + 31) / 32;
    constexpr int DROW = 64 * 2 + 16, XROW = K * 2 + 16;          // padded LDS row strides in bytes
    constexpr int WAVE_LDS = WG_ROWS * (DROW + XROW);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* imgD = smem + wv * WAVE_LDS;
    char* imgX = imgD + WG_ROWS * DROW;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (R + WG_ROWS - 1) / WG_ROWS;

    f32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    float dbias[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dbias[i] = 0.f;

    // staging maps: dY rows are 8 vectors of 16 B, X rows K/8 vectors
    constexpr int XV = K / 8;                       // 16-byte vectors per X row
    constexpr int DY_IT = WG_ROWS * 8 / 64;         // 8 vector loads per lane
    constexpr int X_IT = (WG_ROWS * XV + 63) / 64;
    const int i16 = lane & 15, tg = (lane >> 4) & 1, h = lane >> 5;

    for (int64_t t = wave; t < ntiles; t += nwaves) {
        const int64_t row0 = t * WG_ROWS;
        u32x4 vd[DY_IT], vxx[X_IT];
#pragma unroll
        for (int it = 0; it < DY_IT; ++it) {
            const int v = lane + it * 64, rr = v >> 3, c = v & 7;
            vd[it] = (row0 + rr < R) ? reinterpret_cast<const u32x4*>(dy)[(row0 + rr) * 8 + c] : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int v = lane + it * 64, rr = v / XV, c = v - rr * XV;
            vxx[it] = (rr < WG_ROWS && row0 + rr < R) ? reinterpret_cast<const u32x4*>(x)[(row0 + rr) * XV + c] : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int it = 0; it < DY_IT; ++it) {
            const int v = lane + it * 64, rr = v >> 3, c = v & 7;
            *reinterpret_cast<u32x4*>(imgD + rr * DROW + c * 16) = vd[it];
            float f[8];
            Vec16<bf16>::unpack(vd[it], f);        // the lane's feature chunk c = lane & 7 is the same for every `it`
#pragma unroll
            for (int i = 0; i < 8; ++i) dbias[i] += f[i];
        }
#pragma unroll
        for (int it = 0; it < X_IT; ++it) {
            const int v = lane + it * 64, rr = v / XV, c = v - rr * XV;
            if (rr < WG_ROWS) *reinterpret_cast<u32x4*>(imgX + rr * XROW + c * 16) = vxx[it];
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < WG_ROWS / 16; ++ks) {
            // operand of k-step ks (rows 16 ks .. +15): lane (ch = 32 tile + (lane & 31), h): rows 8h + j (natural order)
            const int rbase = 16 * ks + 8 * h + (i16 >> 2);
            bf16x8 a[2], b[NT];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const char* p = imgD + rbase * DROW + (32 * m + tg * 16 + (i16 & 3) * 4) * 2;
                union { s16x4 s[2]; bf16x8 v; } u;
                u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
                u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * DROW));
                a[m] = u.v;
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                int ch = 32 * n + tg * 16 + (i16 & 3) * 4;
                if (ch >= K) ch = 0;   // K = 16 (mod 32): the upper half-tile reads valid memory, its columns are discarded
                const char* p = imgX + rbase * XROW + ch * 2;
                union { s16x4 s[2]; bf16x8 v; } u;
                u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
                u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * XROW));
                b[n] = u.v;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }

    // per-workgroup partial: the four waves add in wave order through LDS (fixed order)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                 // [64][K] + [64]
    const int r = lane & 31;
    for (int turn = 0; turn < 4; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int o = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * h, k = 32 * n + r;
                        if (k < K) {
                            if (turn == 0) red[o * K + k] = acc[m][n][i];
                            else red[o * K + k] += acc[m][n][i];
                        }
                    }
            // bias: lanes with the same feature chunk (lane & 7) hold partial sums of 8 features
            float s[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s[i] = dbias[i];
                for (int o = 8; o < 64; o <<= 1) s[i] += __shfl_xor(s[i], o, 64);
            }
            if (lane < 8)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (turn == 0) red[64 * K + lane * 8 + i] = s[i];
                    else red[64 * K + lane * 8 + i] += s[i];
                }
        }
        __syncthreads();
    }
    float* dst = partial + (int64_t)blockIdx.x * (64 * K + 64);
    for (int i = threadIdx.x; i < 64 * K + 64; i += blockDim.x) dst[i] = red[i];
}

inline int ceil_log2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

int ln_grid(int64_t R, int rpw) {
    int64_t waves = (R + 2 * rpw - 1) / (2 * rpw);
    int64_t blocks = (waves + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int check_rows(const char* name, int64_t R, int C, int dtype, int* lpr_log2) {
    P4C_CHECK_ARG(R >= 0 && C > 0, "%s: bad sizes R=%lld C=%d", name, (long long)R, C);
    P4C_CHECK_ARG(dtype == P4C_F32 || dtype == P4C_BF16, "%s: dtype must be P4C_F32 or P4C_BF16", name);
    const int esz = dtype == P4C_F32 ? 4 : 2;
    P4C_CHECK_ARG((C * esz) % 16 == 0 && C * esz <= 1024, "%s: a row (C=%d x %d B) must be a multiple of 16 bytes, at most 1 KiB",
                  name, C, esz);
    *lpr_log2 = ceil_log2(C * esz / 16);
    return P4C_OK;
}

int wgrad_grid(int64_t R) {
    const int64_t tiles = (R + WG_ROWS - 1) / WG_ROWS;
    int64_t blocks = (tiles + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * 2;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int KS16>
int launch_wgrad(const void* dy, const void* x, float* partial, int64_t R, int G, hipStream_t stream) {
    constexpr int K = 16 * KS16;
    constexpr int smem_tiles = 4 * WG_ROWS * ((64 * 2 + 16) + (K * 2 + 16));
    constexpr int smem_red = (64 * K + 64) * 4;
    constexpr int smem = smem_tiles > smem_red ? smem_tiles : smem_red;
    P4C_TRY(ensure_dyn_smem((const void*)row_linear_wgrad_kernel<KS16>, smem));
    hipLaunchKernelGGL((row_linear_wgrad_kernel<KS16>), dim3(G), dim3(256), smem, stream, (const bf16*)dy, (const bf16*)x, partial, R);
    P4C_CHECK_LAUNCH("row_linear_wgrad");
    return P4C_OK;
}

// out (n fp32) (+)= sum over the nb slices of x (nb x n): the gradient of a table that was broadcast over the samples (UNETR++'s position
// embedding: t = x + pos), added straight into the parameter's .grad -- the tensor library summed into a fresh tensor that autograd then
// added to the gradient in a second pass.
template <typename T>
__global__ void __launch_bounds__(256) sum_leading_kernel(const T* __restrict__ x, int nb, int64_t n, float* __restrict__ out, int accumulate) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    p4c_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (accumulate) acc = load4f(out + i);
    for (int b = 0; b < nb; ++b) {
        const p4c_f32x4 v = load4f(x + (int64_t)b * n + i);
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    store4f(out + i, acc);
}

}  // namespace
}  // namespace p4c

using namespace p4c;

static int check_mask(const char* who, int64_t R, int Hp, int Wp, int H, int W, RowMask* mk) {
    *mk = RowMask{0, 0, 0, 0};
    if (Hp == 0 && Wp == 0) return P4C_OK;
    P4C_CHECK_ARG(Hp > 0 && Wp > 0 && H > 0 && W > 0 && H <= Hp && W <= Wp && R % ((int64_t)Hp * Wp) == 0,
                  "%s: rows (%lld) must be whole (Hp %d, Wp %d) maps holding the real (H %d, W %d) one", who, (long long)R, Hp, Wp, H, W);
    if (H < Hp || W < Wp) *mk = RowMask{Hp, Wp, H, W};
    return P4C_OK;
}

extern "C" int p4c_row_layernorm_fwd_masked(const void* x, const void* res, const float* gamma, const float* beta, float eps, void* out,
                                            int64_t R, int C, int dtype, int Hp, int Wp, int H, int W, p4c_stream_t stream) {
    int lpr_log2;
    int rc = check_rows("p4c_row_layernorm_fwd", R, C, dtype, &lpr_log2);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_ARG(x && gamma && beta && out, "p4c_row_layernorm_fwd: NULL pointer");
    RowMask mk;
    P4C_TRY(check_mask("p4c_row_layernorm_fwd_masked", R, Hp, Wp, H, W, &mk));
    if (R == 0) return P4C_OK;
    const int G = ln_grid(R, 64 >> lpr_log2);
    if (dtype == P4C_F32)
        hipLaunchKernelGGL(row_layernorm_fwd_kernel<float>, dim3(G), dim3(256), 0, as_stream(stream), (const float*)x, (const float*)res,
                           gamma, beta, eps, (float*)out, R, C, lpr_log2, mk);
    else
        hipLaunchKernelGGL(row_layernorm_fwd_kernel<bf16>, dim3(G), dim3(256), 0, as_stream(stream), (const bf16*)x, (const bf16*)res,
                           gamma, beta, eps, (bf16*)out, R, C, lpr_log2, mk);
    P4C_CHECK_LAUNCH("row_layernorm_fwd");
    return P4C_OK;
}

extern "C" int p4c_row_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta, float eps, void* out,
                                     int64_t R, int C, int dtype, p4c_stream_t stream) {
    return p4c_row_layernorm_fwd_masked(x, res, gamma, beta, eps, out, R, C, dtype, 0, 0, 0, 0, stream);
}

extern "C" size_t p4c_row_layernorm_bwd_workspace_bytes(int64_t R, int C, int dtype) {
    if (R <= 0 || C <= 0) return 0;
    const int esz = dtype == P4C_F32 ? 4 : 2;
    const int lpr_log2 = ceil_log2(C * esz / 16);
    return (size_t)ln_grid(R, 64 >> lpr_log2) * 2 * C * sizeof(float);
}

extern "C" int p4c_row_layernorm_bwd_masked(const void* dy, const void* x, const float* gamma, float eps, void* dx, float* dgamma,
                                            float* dbeta, void* workspace, int64_t R, int C, int dtype, int Hp, int Wp, int H, int W,
                                            p4c_stream_t stream) {
    int lpr_log2;
    int rc = check_rows("p4c_row_layernorm_bwd", R, C, dtype, &lpr_log2);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_ARG(dy && x && gamma && dx && dgamma && dbeta && workspace, "p4c_row_layernorm_bwd: NULL pointer");
    P4C_CHECK_ARG(dbeta == dgamma + C, "p4c_row_layernorm_bwd: dbeta must follow dgamma (one (2,C) buffer)");
    RowMask mk;
    P4C_TRY(check_mask("p4c_row_layernorm_bwd_masked", R, Hp, Wp, H, W, &mk));
    hipStream_t s = as_stream(stream);
    if (R == 0) {
        P4C_CHECK_HIP(zero_words_async(dgamma, 2 * C * sizeof(float), s));   // (not a memset: common.hpp)
        return P4C_OK;
    }
    const int G = ln_grid(R, 64 >> lpr_log2);
    float* partial = reinterpret_cast<float*>(workspace);
    if (dtype == P4C_F32)
        hipLaunchKernelGGL(row_layernorm_bwd_kernel<float>, dim3(G), dim3(256), 0, s, (const float*)dy, (const float*)x, gamma, eps,
                           (float*)dx, partial, R, C, lpr_log2, mk);
    else
        hipLaunchKernelGGL(row_layernorm_bwd_kernel<bf16>, dim3(G), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, gamma, eps,
                           (bf16*)dx, partial, R, C, lpr_log2, mk);
    P4C_CHECK_LAUNCH("row_layernorm_bwd");
    launch_row_param_reduce(partial, G, 2 * C, dgamma, s);
    P4C_CHECK_LAUNCH("row_param_reduce");
    return P4C_OK;
}

extern "C" int p4c_row_layernorm_bwd(const void* dy, const void* x, const float* gamma, float eps, void* dx, float* dgamma,
                                     float* dbeta, void* workspace, int64_t R, int C, int dtype, p4c_stream_t stream) {
    return p4c_row_layernorm_bwd_masked(dy, x, gamma, eps, dx, dgamma, dbeta, workspace, R, C, dtype, 0, 0, 0, 0, stream);
}

static int add_ln_grid(int64_t R) {
    int64_t blocks = (R + 3) / 4;
    const int64_t cap = (int64_t)p4c::num_cus() * 8;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}
static int add_ln_check(const char* name, int64_t R, int C, int dtype, int* nch) {
    P4C_CHECK_ARG(R >= 0 && C > 0 && (dtype == P4C_F32 || dtype == P4C_BF16), "%s: bad sizes / dtype", name);
    const int esz = dtype == P4C_F32 ? 4 : 2;
    P4C_CHECK_ARG((C * esz) % 16 == 0 && C * esz <= 2048, "%s: a row (C=%d x %d B) must be a multiple of 16 bytes, at most 2 KiB", name, C, esz);
    *nch = C * esz > 1024 ? 2 : 1;
    return P4C_OK;
}

extern "C" int p4c_row_add_layernorm_fwd(const void* x, const void* add, int64_t add_rows, const float* gamma, const float* beta, float eps,
                                         void* sum_out, void* out, int64_t R, int C, int dtype, p4c_stream_t stream) {
    int nch;
    P4C_TRY(add_ln_check("p4c_row_add_layernorm_fwd", R, C, dtype, &nch));
    P4C_CHECK_ARG(x && gamma && beta && out && (!add || add_rows > 0), "p4c_row_add_layernorm_fwd: NULL pointer");
    if (R == 0) return P4C_OK;
    hipStream_t s = as_stream(stream);
    const int G = add_ln_grid(R);
#define P4C_LAUNCH_ALN(T, N) hipLaunchKernelGGL((row_add_layernorm_fwd_kernel<T, N>), dim3(G), dim3(256), 0, s, (const T*)x, (const T*)add, add_rows, gamma, beta, eps, (T*)sum_out, (T*)out, R, C)
    if (dtype == P4C_F32) { if (nch == 2) P4C_LAUNCH_ALN(float, 2); else P4C_LAUNCH_ALN(float, 1); }
    else { if (nch == 2) P4C_LAUNCH_ALN(bf16, 2); else P4C_LAUNCH_ALN(bf16, 1); }
#undef P4C_LAUNCH_ALN
    P4C_CHECK_LAUNCH("row_add_layernorm_fwd");
    return P4C_OK;
}

extern "C" size_t p4c_row_add_layernorm_bwd_workspace_bytes(int64_t R, int C) {
    if (R <= 0 || C <= 0) return 0;
    return (size_t)add_ln_grid(R) * 2 * C * sizeof(float);
}

extern "C" int p4c_row_add_layernorm_bwd(const void* dy, const void* t, const void* extra, const float* gamma, float eps, void* dt, float* dgamma,
                                         float* dbeta, void* workspace, int64_t R, int C, int dtype, p4c_stream_t stream) {
    int nch;
    P4C_TRY(add_ln_check("p4c_row_add_layernorm_bwd", R, C, dtype, &nch));
    P4C_CHECK_ARG(dy && t && gamma && dt && dgamma && dbeta && workspace && R > 0, "p4c_row_add_layernorm_bwd: NULL pointer / no rows");
    P4C_CHECK_ARG(dbeta == dgamma + C, "p4c_row_add_layernorm_bwd: dbeta must follow dgamma (one (2,C) buffer)");
    hipStream_t s = as_stream(stream);
    const int G = add_ln_grid(R);
    float* partial = reinterpret_cast<float*>(workspace);
#define P4C_LAUNCH_ALNB(T, N) hipLaunchKernelGGL((row_add_layernorm_bwd_kernel<T, N>), dim3(G), dim3(256), 0, s, (const T*)dy, (const T*)t, (const T*)extra, gamma, eps, (T*)dt, partial, R, C)
    if (dtype == P4C_F32) { if (nch == 2) P4C_LAUNCH_ALNB(float, 2); else P4C_LAUNCH_ALNB(float, 1); }
    else { if (nch == 2) P4C_LAUNCH_ALNB(bf16, 2); else P4C_LAUNCH_ALNB(bf16, 1); }
#undef P4C_LAUNCH_ALNB
    P4C_CHECK_LAUNCH("row_add_layernorm_bwd");
    launch_row_param_reduce(partial, G, 2 * C, dgamma, s);
    P4C_CHECK_LAUNCH("row_param_reduce");
    return P4C_OK;
}

extern "C" size_t p4c_row_linear_wgrad_workspace_bytes(int64_t R, int K) {
    if (R <= 0 || K <= 0) return 0;
    return (size_t)wgrad_grid(R) * (64 * K + 64) * sizeof(float);
}

extern "C" int p4c_row_linear_wgrad(const void* dy, const void* x, float* dw_db, void* workspace, int64_t R, int O, int K, int dtype,
                                    p4c_stream_t stream) {
    P4C_CHECK_ARG(dy && x && dw_db && workspace, "p4c_row_linear_wgrad: NULL pointer");
    P4C_CHECK_ARG(O == 64, "p4c_row_linear_wgrad: O = %d output features (only 64 is implemented; pad)", O);
    P4C_CHECK_ARG(K % 16 == 0 && K >= 16 && K <= 128, "p4c_row_linear_wgrad: K = %d input features (multiples of 16 up to 128; pad)", K);
    P4C_CHECK_ARG(dtype == P4C_BF16, "p4c_row_linear_wgrad: bf16 rows only (the fp32 flavour keeps the library GEMM)");
    P4C_CHECK_ARG(R > 0, "p4c_row_linear_wgrad: R must be positive");
    hipStream_t s = as_stream(stream);
    const int G = wgrad_grid(R);
    float* partial = reinterpret_cast<float*>(workspace);
    int rc;
    switch (K / 16) {
        case 1: rc = launch_wgrad<1>(dy, x, partial, R, G, s); break;
        case 2: rc = launch_wgrad<2>(dy, x, partial, R, G, s); break;
        case 3: rc = launch_wgrad<3>(dy, x, partial, R, G, s); break;
        case 4: rc = launch_wgrad<4>(dy, x, partial, R, G, s); break;
        case 5: rc = launch_wgrad<5>(dy, x, partial, R, G, s); break;
        case 6: rc = launch_wgrad<6>(dy, x, partial, R, G, s); break;
        case 7: rc = launch_wgrad<7>(dy, x, partial, R, G, s); break;
        default: rc = launch_wgrad<8>(dy, x, partial, R, G, s); break;
    }
    if (rc != P4C_OK) return rc;
    const int n = 64 * K + 64;
    launch_row_param_reduce(partial, G, n, dw_db, s);
    P4C_CHECK_LAUNCH("row_param_reduce");
    return P4C_OK;
}


extern "C" int p4c_sum_leading(const void* x, int dtype, int nb, int64_t n, float* out, int accumulate, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && out && nb > 0 && n > 0 && n % 4 == 0 && (dtype == P4C_F32 || dtype == P4C_BF16), "p4c_sum_leading: bad arguments");
    P4C_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0, "p4c_sum_leading: unaligned buffers");
    const unsigned blocks = (unsigned)((n / 4 + 255) / 256);
    if (dtype == P4C_F32) hipLaunchKernelGGL(sum_leading_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float*)x, nb, n, out, accumulate);
    else hipLaunchKernelGGL(sum_leading_kernel<bf16>, dim3(blocks), dim3(256), 0, as_stream(stream), (const bf16*)x, nb, n, out, accumulate);
    P4C_CHECK_LAUNCH("p4c_sum_leading");
    return P4C_OK;
}
