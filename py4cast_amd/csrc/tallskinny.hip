// Tall-skinny products for the efficient paired attention (EPA) of UNETR++ (config/CLI/model/unetrpp.yaml; mfai v5.0.1's UNetRPP, absent
// from the reference checkout).  Per (sample, head) group the attention works on N x d token matrices with N = H*W/16 ... H*W/1024
// tokens and d, e <= 64 columns; every product is either
//   gram :  C[g] (d x e) = X[g]^T Y[g]                   -- a reduction over the N tokens (q^T k, k^T W^T, dA = dXca^T v, ...)
//   apply:  O[g] (N x e) = X[g] (N x d) M[g] (d x e)      -- a per-token small matrix product (v A^T, q KP, S VP^T, ...)
// and the two are each other's adjoints, so forward AND backward of the block are these two kernels (py4cast_amd/ops_ts.py).
// Both are HBM-streaming passes over the token matrices (arithmetic intensity <= 2*min(d,e)/esz flop per byte, far below the
// ridge for the stages that matter: d = 8..32 at N = 16 384..1 024), operands addressed in place inside the (B, N, 4, heads, d)
// output of the qkvv projection / the (B, N, C) token tensor through (group, row) strides: no permute / contiguous copies.
#include "common.hpp"

namespace p4c {
namespace ts {

constexpr int TOK = 64;      // tokens per LDS tile
constexpr int MAXD = 64;

template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldf<bf16>(const bf16* p) { return __bfloat162float(*p); }

struct Mat {             // token matrix of group g = (b, h): element (n, i) at base + b*bs + h*hs + n*rs + i
    const void* base;
    int64_t bs, hs, rs;
};
struct MatOut {
    void* base;
    int64_t bs, hs, rs;
};

// C[g][split] (d x e) partial = sum over the split's tokens of X[n,:]^T Y[n,:]; thread t owns a 4 x 4 block of C (d, e multiples of 4)
template <typename TX, typename TY>
__global__ void __launch_bounds__(256) gram_kernel(Mat X, Mat Y, float* __restrict__ part, int heads, int64_t N, int d, int e, int nsplit) {
    __shared__ float lx[TOK][MAXD + 1], ly[TOK][MAXD + 1];
    const int g = blockIdx.x, sp = blockIdx.y;
    const int b = g / heads, h = g - b * heads;
    const TX* xb = reinterpret_cast<const TX*>(X.base) + b * X.bs + h * X.hs;
    const TY* yb = reinterpret_cast<const TY*>(Y.base) + b * Y.bs + h * Y.hs;
    const int bi = (threadIdx.x >> 4) * 4, bj = (threadIdx.x & 15) * 4;      // 16 x 16 thread grid over up to 64 x 64 outputs
    const bool live = bi < d && bj < e;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
    const int64_t per = (N + nsplit - 1) / nsplit;
    const int64_t n0 = sp * per, n1 = (n0 + per < N) ? n0 + per : N;
    for (int64_t t0 = n0; t0 < n1; t0 += TOK) {
        const int nt = (int)((n1 - t0) < TOK ? (n1 - t0) : TOK);
        for (int i = threadIdx.x; i < TOK * d; i += 256) {
            const int r = i / d, c = i - r * d;
            lx[r][c] = r < nt ? ldf<TX>(xb + (t0 + r) * X.rs + c) : 0.f;
        }
        for (int i = threadIdx.x; i < TOK * e; i += 256) {
            const int r = i / e, c = i - r * e;
            ly[r][c] = r < nt ? ldf<TY>(yb + (t0 + r) * Y.rs + c) : 0.f;
        }
        __syncthreads();
        if (live) {
#pragma unroll 8
            for (int r = 0; r < TOK; ++r) {
                float xv[4], yv[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) { xv[a] = lx[r][bi + a]; yv[a] = ly[r][bj + a]; }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_fmaf(xv[a], yv[c], acc[a][c]);
            }
        }
        __syncthreads();
    }
    if (live) {
        float* dst = part + ((int64_t)g * nsplit + sp) * d * e;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) dst[(bi + a) * e + bj + c] = acc[a][c];
    }
}

// O[g] (N x e) = X[g] (N x d) M[g] (d x e) (+ O when accumulate): 64 tokens x 4 column groups per workgroup, M in LDS
template <typename TX, typename TO>
__global__ void __launch_bounds__(256) apply_kernel(Mat X, const float* __restrict__ M, int64_t m_gs, MatOut O, int heads, int64_t N,
                                                    int d, int e, int accumulate) {
    __shared__ float lm[MAXD][MAXD + 1];
    __shared__ float lx[TOK][MAXD + 1];
    const int g = blockIdx.x;
    const int b = g / heads, h = g - b * heads;
    const TX* xb = reinterpret_cast<const TX*>(X.base) + b * X.bs + h * X.hs;
    TO* ob = reinterpret_cast<TO*>(O.base) + b * O.bs + h * O.hs;
    const float* mg = M + (int64_t)g * m_gs;
    for (int i = threadIdx.x; i < d * e; i += 256) lm[i / e][i % e] = mg[i];
    const int tok = threadIdx.x >> 2, jg = threadIdx.x & 3;
    const int ew = (e + 3) / 4;                 // columns per thread (<= 16), column c = jg * ew + k
    for (int64_t t0 = (int64_t)blockIdx.y * TOK; t0 < N; t0 += (int64_t)gridDim.y * TOK) {
        __syncthreads();
        const int nt = (int)((N - t0) < TOK ? (N - t0) : TOK);
        for (int i = threadIdx.x; i < TOK * d; i += 256) {
            const int r = i / d, c = i - r * d;
            lx[r][c] = r < nt ? ldf<TX>(xb + (t0 + r) * X.rs + c) : 0.f;
        }
        __syncthreads();
        if (tok < nt) {
            float acc[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            for (int i = 0; i < d; ++i) {
                const float xv = lx[tok][i];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < ew) acc[k] = __builtin_fmaf(xv, lm[i][jg * ew + k < e ? jg * ew + k : 0], acc[k]);
            }
            TO* orow = ob + (t0 + tok) * O.rs;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int c = jg * ew + k;
                if (k < ew && c < e) {
                    float v = acc[k];
                    if (accumulate) v += to_f32<TO>(orow[c]);
                    orow[c] = from_f32<TO>(v);
                }
            }
        }
    }
}

}  // namespace ts
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_ts_gram_splits(int64_t N) {
    int64_t s = (N + 2047) / 2048;     // >= 2048 tokens per workgroup
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    return (int)s;
}

extern "C" int p4c_ts_gram(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const void* y, int y_dtype,
                           int64_t y_bs, int64_t y_hs, int64_t y_rs, float* partial, int B, int heads, int64_t N, int d, int e,
                           p4c_stream_t stream) {
    P4C_CHECK_ARG(x && y && partial, "p4c_ts_gram: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && N > 0 && d > 0 && e > 0 && d <= ts::MAXD && e <= ts::MAXD && d % 4 == 0 && e % 4 == 0,
                  "p4c_ts_gram: d, e must be multiples of 4 up to %d (got %d, %d)", ts::MAXD, d, e);
    const int ns = p4c_ts_gram_splits(N);
    const ts::Mat X{x, x_bs, x_hs, x_rs}, Y{y, y_bs, y_hs, y_rs};
    const dim3 grid(B * heads, ns);
    hipStream_t st = as_stream(stream);
#define P4C_GRAM(TX, TY) hipLaunchKernelGGL((ts::gram_kernel<TX, TY>), grid, dim3(256), 0, st, X, Y, partial, heads, N, d, e, ns)
    if (x_dtype == P4C_F32 && y_dtype == P4C_F32) P4C_GRAM(float, float);
    else if (x_dtype == P4C_BF16 && y_dtype == P4C_BF16) P4C_GRAM(bf16, bf16);
    else if (x_dtype == P4C_BF16 && y_dtype == P4C_F32) P4C_GRAM(bf16, float);
    else if (x_dtype == P4C_F32 && y_dtype == P4C_BF16) P4C_GRAM(float, bf16);
    else return fail(P4C_ERR_INVALID, "p4c_ts_gram: bad dtype");
#undef P4C_GRAM
    P4C_CHECK_LAUNCH("p4c_ts_gram");
    return P4C_OK;
}

extern "C" int p4c_ts_apply(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m, int64_t m_gs,
                            void* out, int out_dtype, int64_t o_bs, int64_t o_hs, int64_t o_rs, int B, int heads, int64_t N, int d, int e,
                            int accumulate, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && m && out, "p4c_ts_apply: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && N > 0 && d > 0 && e > 0 && d <= ts::MAXD && e <= ts::MAXD, "p4c_ts_apply: d, e <= %d", ts::MAXD);
    const ts::Mat X{x, x_bs, x_hs, x_rs};
    const ts::MatOut O{out, o_bs, o_hs, o_rs};
    int64_t chunks = (N + ts::TOK - 1) / ts::TOK;
    const int64_t cap = (int64_t)num_cus() * 8 / (B * heads) + 1;
    if (chunks > cap) chunks = cap;
    const dim3 grid(B * heads, (unsigned)chunks);
    hipStream_t st = as_stream(stream);
#define P4C_APPLY(TX, TO) hipLaunchKernelGGL((ts::apply_kernel<TX, TO>), grid, dim3(256), 0, st, X, m, m_gs, O, heads, N, d, e, accumulate)
    if (x_dtype == P4C_F32 && out_dtype == P4C_F32) P4C_APPLY(float, float);
    else if (x_dtype == P4C_BF16 && out_dtype == P4C_BF16) P4C_APPLY(bf16, bf16);
    else if (x_dtype == P4C_BF16 && out_dtype == P4C_F32) P4C_APPLY(bf16, float);
    else if (x_dtype == P4C_F32 && out_dtype == P4C_BF16) P4C_APPLY(float, bf16);
    else return fail(P4C_ERR_INVALID, "p4c_ts_apply: bad dtype");
#undef P4C_APPLY
    P4C_CHECK_LAUNCH("p4c_ts_apply");
    return P4C_OK;
}
