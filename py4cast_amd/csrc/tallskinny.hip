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

constexpr int MAXD = 64;
constexpr int TOKG = 32;      // tokens per LDS tile of the gram kernel
constexpr int TOKA = 16;      // tokens per LDS tile of the apply kernel
constexpr int MAXCOL = 128;   // columns of a token tile held in LDS (heads x width of the chunk of heads a workgroup serves)

struct Mat {             // token matrix of group g = (b, h): element (n, i) at base + b*bs + h*hs + n*rs + i
    const void* base;
    int64_t bs, hs, rs;
};
struct MatOut {
    void* base;
    int64_t bs, hs, rs;
};

// 4 consecutive elements as fp32 (offsets are multiples of 4 elements: 8-byte / 16-byte aligned accesses)
__device__ __forceinline__ p4c_f32x4 ld4(const float* p) { return *reinterpret_cast<const p4c_f32x4*>(p); }
__device__ __forceinline__ p4c_f32x4 ld4(const bf16* p) { return load4f(p); }

// Stage `nt` token rows x `hc` heads x `w` columns (w % 4 == 0) into LDS as fp32, row t at dst + t*ld, head h at column h*w.
// With heads adjacent in memory (hs == w) consecutive threads read consecutive quads of a row: whole cache lines per row.
template <typename T>
__device__ __forceinline__ void stage(float* dst, int ld, const T* src, int64_t rs, int64_t hs, int nt, int tile, int hc, int w) {
    const int qpr = hc * (w >> 2);                    // quads per row
    for (int i = threadIdx.x; i < tile * qpr; i += 256) {
        const int t = i / qpr, q = i - t * qpr;
        const int h = q / (w >> 2), c = (q - h * (w >> 2)) << 2;
        p4c_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (t < nt) v = ld4(src + (int64_t)t * rs + (int64_t)h * hs + c);
        *reinterpret_cast<p4c_f32x4*>(dst + t * ld + h * w + c) = v;
    }
}

// gram: part[b][split][h] (d x e) = sum over the split's tokens of X[b,h][n,:]^T Y[b,h][n,:].
// A workgroup serves `hc` heads (blockIdx.z) of one sample and one token split; a thread owns one 4 x 4 block of one head's d x e
// result (a "task"); when there are fewer tasks than threads the tile's tokens are dealt to `groups` thread groups whose sums
// are combined through LDS at the end (fixed order).
// NORMS: the partial of a (b, split, head) is d*e + d + e floats: X^T Y, then the column sums of squares of X and of Y (what EPA
// normalises q and k by) -- accumulated by the tasks of block column / block row 0 from the values they hold anyway.
template <typename TX, typename TY, bool NORMS = false>
__global__ void __launch_bounds__(256) gram_kernel(Mat X, Mat Y, float* __restrict__ part, int heads, int64_t N, int d, int e, int nsplit,
                                                   int hc, int tpg) {
    __shared__ __attribute__((aligned(16))) float lds[2 * TOKG * (MAXCOL + 4)];
    const int b = blockIdx.x, sp = blockIdx.y, h0 = blockIdx.z * hc;
    const int nh = (heads - h0) < hc ? (heads - h0) : hc;
    const int ldx = nh * d + 4, ldy = nh * e + 4;
    float* lx = lds;
    float* ly = lds + TOKG * (MAXCOL + 4);
    const TX* xb = reinterpret_cast<const TX*>(X.base) + b * X.bs + h0 * X.hs;
    const TY* yb = reinterpret_cast<const TY*>(Y.base) + b * Y.bs + h0 * Y.hs;
    const int db = d >> 2, eb = e >> 2;
    const int ntasks = nh * db * eb;
    const int groups = 256 / tpg, grp = threadIdx.x / tpg, task = threadIdx.x - grp * tpg;
    const bool live = task < ntasks;
    const int th = task / (db * eb), tr = task - th * db * eb;
    const int ci = th * d + (tr / eb) * 4, cj = th * e + (tr % eb) * 4;
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = 0.f;
    float nx[4] = {0.f, 0.f, 0.f, 0.f}, ny[4] = {0.f, 0.f, 0.f, 0.f};
    const bool own_x = NORMS && live && (tr % eb) == 0, own_y = NORMS && live && (tr / eb) == 0;
    const int64_t per = (N + nsplit - 1) / nsplit;
    const int64_t n0 = sp * per, n1 = (n0 + per < N) ? n0 + per : N;
    for (int64_t t0 = n0; t0 < n1; t0 += TOKG) {
        const int nt = (int)((n1 - t0) < TOKG ? (n1 - t0) : TOKG);
        __syncthreads();
        stage<TX>(lx, ldx, xb + t0 * X.rs, X.rs, X.hs, nt, TOKG, nh, d);
        stage<TY>(ly, ldy, yb + t0 * Y.rs, Y.rs, Y.hs, nt, TOKG, nh, e);
        __syncthreads();
        if (live) {
            for (int t = grp; t < TOKG; t += groups) {
                const p4c_f32x4 xv = *reinterpret_cast<const p4c_f32x4*>(lx + t * ldx + ci);
                const p4c_f32x4 yv = *reinterpret_cast<const p4c_f32x4*>(ly + t * ldy + cj);
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[a][c] = __builtin_fmaf(xv[a], yv[c], acc[a][c]);
                if (own_x) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) nx[a] = __builtin_fmaf(xv[a], xv[a], nx[a]);
                }
                if (own_y) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) ny[c] = __builtin_fmaf(yv[c], yv[c], ny[c]);
                }
            }
        }
    }
    // combine the token groups (group 0 first, then 1, ...): red[grp][task][16 (+ 8)]
    __syncthreads();
    float* red = lds;
    constexpr int RS = NORMS ? 24 : 16;
    const int pstride = NORMS ? d * e + d + e : d * e;
    if (live) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[(grp * tpg + task) * RS + a * 4 + c] = acc[a][c];
        if (NORMS) {
#pragma unroll
            for (int a = 0; a < 4; ++a) { red[(grp * tpg + task) * RS + 16 + a] = nx[a]; red[(grp * tpg + task) * RS + 20 + a] = ny[a]; }
        }
    }
    __syncthreads();
    float* pb = part + (((int64_t)b * nsplit + sp) * heads + h0) * pstride;
    for (int i = threadIdx.x; i < ntasks * RS; i += 256) {
        const int tk = i / RS, el = i - tk * RS;
        const int hh = tk / (db * eb), r2 = tk - hh * db * eb;
        if (el >= 16) {   // column sums of squares: owned by the tasks of block column 0 (X) / block row 0 (Y)
            const bool isx = el < 20;
            if (isx ? (r2 % eb) != 0 : (r2 / eb) != 0) continue;
            float s = 0.f;
            for (int g2 = 0; g2 < groups; ++g2) s += red[(g2 * tpg + tk) * RS + el];
            if (isx) pb[(int64_t)hh * pstride + d * e + (r2 / eb) * 4 + (el - 16)] = s;
            else pb[(int64_t)hh * pstride + d * e + d + (r2 % eb) * 4 + (el - 20)] = s;
            continue;
        }
        float s = 0.f;
        for (int g2 = 0; g2 < groups; ++g2) s += red[(g2 * tpg + tk) * RS + el];
        const int ii = (r2 / eb) * 4 + (el >> 2), jj = (r2 % eb) * 4 + (el & 3);
        pb[(int64_t)hh * pstride + ii * e + jj] = s;
    }
}

// apply: O[b,h] (N x e) = X[b,h] (N x d) M[b,h] (d x e) (+ O when accumulate).  A workgroup serves `hc` heads (blockIdx.z) and walks
// token tiles; a task = (token, head, 4 output columns): d x 4 FMAs from the staged token row and the heads' matrices in LDS.
template <typename TX, typename TO>
__global__ void __launch_bounds__(256) apply_kernel(Mat X, const float* __restrict__ M, int64_t m_bs, int64_t m_hs, MatOut O, int heads,
                                                    int64_t N, int d, int e, int accumulate, int hc) {
    __shared__ __attribute__((aligned(16))) float lm[8192];                       // hc * d * e floats
    __shared__ __attribute__((aligned(16))) float lx[TOKA * (MAXCOL + 4)];
    const int b = blockIdx.x, h0 = blockIdx.z * hc;
    const int nh = (heads - h0) < hc ? (heads - h0) : hc;
    const TX* xb = reinterpret_cast<const TX*>(X.base) + b * X.bs + h0 * X.hs;
    TO* ob = reinterpret_cast<TO*>(O.base) + b * O.bs + h0 * O.hs;
    for (int i = threadIdx.x; i < nh * d * e; i += 256) {
        const int hh = i / (d * e);
        lm[i] = M[b * m_bs + (int64_t)(h0 + hh) * m_hs + (i - hh * d * e)];
    }
    const int ldx = nh * d + 4, eb = e >> 2;
    const int per_tok = nh * eb;
    for (int64_t t0 = (int64_t)blockIdx.y * TOKA; t0 < N; t0 += (int64_t)gridDim.y * TOKA) {
        const int nt = (int)((N - t0) < TOKA ? (N - t0) : TOKA);
        __syncthreads();
        stage<TX>(lx, ldx, xb + t0 * X.rs, X.rs, X.hs, nt, TOKA, nh, d);
        __syncthreads();
        for (int task = threadIdx.x; task < nt * per_tok; task += 256) {
            const int t = task / per_tok, r = task - t * per_tok;
            const int hh = r / eb, j = (r - hh * eb) << 2;
            const float* xr = lx + t * ldx + hh * d;
            const float* mr = lm + hh * d * e + j;
            p4c_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < d; ++i) {
                const float xv = xr[i];
                const p4c_f32x4 mv = *reinterpret_cast<const p4c_f32x4*>(mr + i * e);
                acc[0] = __builtin_fmaf(xv, mv[0], acc[0]); acc[1] = __builtin_fmaf(xv, mv[1], acc[1]);
                acc[2] = __builtin_fmaf(xv, mv[2], acc[2]); acc[3] = __builtin_fmaf(xv, mv[3], acc[3]);
            }
            TO* op = ob + (t0 + t) * O.rs + (int64_t)hh * O.hs + j;
            if (accumulate) {
                const p4c_f32x4 old = ld4(op);
                acc[0] += old[0]; acc[1] += old[1]; acc[2] += old[2]; acc[3] += old[3];
            }
            store4f(op, acc);
        }
    }
}

}  // namespace ts
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_ts_gram_splits(int64_t N) {
    int64_t s = (N + 63) / 64;        // two 32-token tiles per workgroup: the launch needs >> 256 workgroups to hide its load ->
    if (s > 256) s = 256;             // barrier -> compute -> barrier rhythm behind other workgroups of the same CU
    if (s < 1) s = 1;
    return (int)s;
}

static int pow2_ge_i(int v) { int p = 1; while (p < v) p <<= 1; return p; }

extern "C" int p4c_ts_gram(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const void* y, int y_dtype,
                           int64_t y_bs, int64_t y_hs, int64_t y_rs, float* partial, int B, int heads, int64_t N, int d, int e,
                           p4c_stream_t stream) {
    P4C_CHECK_ARG(x && y && partial, "p4c_ts_gram: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && N > 0 && d > 0 && e > 0 && d <= ts::MAXD && e <= ts::MAXD && d % 4 == 0 && e % 4 == 0,
                  "p4c_ts_gram: d, e must be multiples of 4 up to %d (got %d, %d)", ts::MAXD, d, e);
    P4C_CHECK_ARG(x_bs % 4 == 0 && x_hs % 4 == 0 && x_rs % 4 == 0 && y_bs % 4 == 0 && y_hs % 4 == 0 && y_rs % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                  "p4c_ts_gram: strides must be multiples of 4 elements, bases 16-byte aligned");
    const int ns = p4c_ts_gram_splits(N);
    // heads per workgroup: at most 256 tasks (4 x 4 result blocks) and MAXCOL staged columns per operand
    int hc = 256 / ((d / 4) * (e / 4));
    if (hc > ts::MAXCOL / d) hc = ts::MAXCOL / d;
    if (hc > ts::MAXCOL / e) hc = ts::MAXCOL / e;
    if (hc > heads) hc = heads;
    if (hc < 1) hc = 1;
    int tpg = pow2_ge_i(hc * (d / 4) * (e / 4));       // threads per token group
    if (tpg > 256) tpg = 256;
    if (tpg < 8) tpg = 8;                               // <= 32 token groups: a tile has 32 tokens
    const ts::Mat X{x, x_bs, x_hs, x_rs}, Y{y, y_bs, y_hs, y_rs};
    const dim3 grid(B, ns, (heads + hc - 1) / hc);
    hipStream_t st = as_stream(stream);
#define P4C_GRAM(TX, TY) hipLaunchKernelGGL((ts::gram_kernel<TX, TY>), grid, dim3(256), 0, st, X, Y, partial, heads, N, d, e, ns, hc, tpg)
    if (x_dtype == P4C_F32 && y_dtype == P4C_F32) P4C_GRAM(float, float);
    else if (x_dtype == P4C_BF16 && y_dtype == P4C_BF16) P4C_GRAM(bf16, bf16);
    else if (x_dtype == P4C_BF16 && y_dtype == P4C_F32) P4C_GRAM(bf16, float);
    else if (x_dtype == P4C_F32 && y_dtype == P4C_BF16) P4C_GRAM(float, bf16);
    else return fail(P4C_ERR_INVALID, "p4c_ts_gram: bad dtype");
#undef P4C_GRAM
    P4C_CHECK_LAUNCH("p4c_ts_gram");
    return P4C_OK;
}

// X^T Y together with the column sums of squares of X and Y: partial (B, splits, heads, d*e + d + e) (see gram_kernel<NORMS>)
extern "C" int p4c_ts_gram_norms(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const void* y, int y_dtype,
                                 int64_t y_bs, int64_t y_hs, int64_t y_rs, float* partial, int B, int heads, int64_t N, int d, int e,
                                 p4c_stream_t stream) {
    P4C_CHECK_ARG(x && y && partial, "p4c_ts_gram_norms: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && N > 0 && d > 0 && e > 0 && d <= ts::MAXD && e <= ts::MAXD && d % 4 == 0 && e % 4 == 0,
                  "p4c_ts_gram_norms: d, e must be multiples of 4 up to %d (got %d, %d)", ts::MAXD, d, e);
    P4C_CHECK_ARG(x_dtype == P4C_BF16 && y_dtype == P4C_BF16, "p4c_ts_gram_norms: bf16 token matrices");
    P4C_CHECK_ARG(x_bs % 4 == 0 && x_hs % 4 == 0 && x_rs % 4 == 0 && y_bs % 4 == 0 && y_hs % 4 == 0 && y_rs % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0,
                  "p4c_ts_gram_norms: strides must be multiples of 4 elements, bases 16-byte aligned");
    const int ns = p4c_ts_gram_splits(N);
    int hc = 256 / ((d / 4) * (e / 4));
    if (hc > ts::MAXCOL / d) hc = ts::MAXCOL / d;
    if (hc > ts::MAXCOL / e) hc = ts::MAXCOL / e;
    if (hc > heads) hc = heads;
    if (hc < 1) hc = 1;
    int tpg = pow2_ge_i(hc * (d / 4) * (e / 4));
    if (tpg > 256) tpg = 256;
    if (tpg < 8) tpg = 8;
    const ts::Mat X{x, x_bs, x_hs, x_rs}, Y{y, y_bs, y_hs, y_rs};
    const dim3 grid(B, ns, (heads + hc - 1) / hc);
    hipLaunchKernelGGL((ts::gram_kernel<bf16, bf16, true>), grid, dim3(256), 0, as_stream(stream), X, Y, partial, heads, N, d, e, ns, hc, tpg);
    P4C_CHECK_LAUNCH("p4c_ts_gram_norms");
    return P4C_OK;
}

extern "C" int p4c_ts_apply(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m, int64_t m_gs,
                            void* out, int out_dtype, int64_t o_bs, int64_t o_hs, int64_t o_rs, int B, int heads, int64_t N, int d, int e,
                            int accumulate, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && m && out, "p4c_ts_apply: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && N > 0 && d > 0 && e > 0 && d <= ts::MAXD && e <= ts::MAXD && d % 4 == 0 && e % 4 == 0,
                  "p4c_ts_apply: d, e must be multiples of 4 up to %d (got %d, %d)", ts::MAXD, d, e);
    P4C_CHECK_ARG(x_bs % 4 == 0 && x_hs % 4 == 0 && x_rs % 4 == 0 && o_bs % 4 == 0 && o_hs % 4 == 0 && o_rs % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0,
                  "p4c_ts_apply: strides must be multiples of 4 elements, bases aligned");
    int hc = 8192 / (d * e);
    if (hc > ts::MAXCOL / d) hc = ts::MAXCOL / d;
    if (hc > heads) hc = heads;
    if (hc < 1) hc = 1;
    const int zc = (heads + hc - 1) / hc;
    const ts::Mat X{x, x_bs, x_hs, x_rs};
    const ts::MatOut O{out, o_bs, o_hs, o_rs};
    int64_t chunks = (N + ts::TOKA - 1) / ts::TOKA;
    const int64_t cap = (int64_t)num_cus() * 8 / ((int64_t)B * zc) + 1;
    if (chunks > cap) chunks = cap;
    const dim3 grid(B, (unsigned)chunks, zc);
    hipStream_t st = as_stream(stream);
    const int64_t m_hs = m_gs, m_bs = m_gs * heads;   // M is (B, heads, d, e) dense, or one matrix for all groups (m_gs == 0)
#define P4C_APPLY(TX, TO) hipLaunchKernelGGL((ts::apply_kernel<TX, TO>), grid, dim3(256), 0, st, X, m, m_bs, m_hs, O, heads, N, d, e, accumulate, hc)
    if (x_dtype == P4C_F32 && out_dtype == P4C_F32) P4C_APPLY(float, float);
    else if (x_dtype == P4C_BF16 && out_dtype == P4C_BF16) P4C_APPLY(bf16, bf16);
    else if (x_dtype == P4C_BF16 && out_dtype == P4C_F32) P4C_APPLY(bf16, float);
    else if (x_dtype == P4C_F32 && out_dtype == P4C_BF16) P4C_APPLY(float, bf16);
    else return fail(P4C_ERR_INVALID, "p4c_ts_apply: bad dtype");
#undef P4C_APPLY
    P4C_CHECK_LAUNCH("p4c_ts_apply");
    return P4C_OK;
}

// ---------------------------------------------------------------------------------------------
// The small matrices of one EPA block (UNETR++ efficient paired attention; py4cast_amd/unetrpp.py::EPA) in ONE launch each way.
// Per (sample, head), with G = q^T k, Gq = q^T q, Gk = k^T k (d x d, from p4c_ts_gram) and KP (d x p):
//   nq_i = max(sqrt(max(Gq_ii, 0)), eps), nk_j likewise;   A = softmax_j(t1 G_ij / (nq_i nk_j));   Mq_ic = t2 KP_ic / nq_i
// forward writes At = A^T (what the apply kernel multiplies v with), Mq, nq, nk.  As torch ops this was ~14 launches forward and ~35
// backward per block (clamps, square roots, outer products, broadcast reductions ...), ~6 000 per UNetRPP training step.
// One workgroup per (b, h); thread -> row i; everything in fp32, sums in index order (deterministic).
namespace p4c {
namespace ts {
constexpr float EPA_EPS = 1e-12f;   // F.normalize's clamp

__global__ void __launch_bounds__(64) epa_small_fwd_kernel(const float* __restrict__ G, const float* __restrict__ Gq, const float* __restrict__ Gk,
                                                           const float* __restrict__ KP, const float* __restrict__ t1, const float* __restrict__ t2,
                                                           float* __restrict__ At, float* __restrict__ Mq, float* __restrict__ nq_out,
                                                           float* __restrict__ nk_out, int heads, int d, int p, int dstride) {
    // dstride = d: Gq / Gk are the full (d x d) q^T q / k^T k, only their diagonals are read;  dstride = 1: they ARE the diagonals (B, h, d)
    __shared__ float nk[64];
    const int g = blockIdx.x, h = g % heads, i = threadIdx.x;
    const float* Gg = G + (int64_t)g * d * d;
    float nqi = 1.f;
    if (i < d) {
        nqi = fmaxf(sqrtf(fmaxf(Gq[(int64_t)g * d * dstride + i * dstride + (dstride > 1 ? i : 0)], 0.f)), EPA_EPS);
        const float nki = fmaxf(sqrtf(fmaxf(Gk[(int64_t)g * d * dstride + i * dstride + (dstride > 1 ? i : 0)], 0.f)), EPA_EPS);
        nk[i] = nki;
        nq_out[(int64_t)g * d + i] = nqi;
        nk_out[(int64_t)g * d + i] = nki;
    }
    __syncthreads();
    if (i >= d) return;
    const float s1 = t1[h], s2 = t2[h];
    float mx = -INFINITY;
    for (int j = 0; j < d; ++j) mx = fmaxf(mx, Gg[i * d + j] / (nqi * nk[j]) * s1);
    float sum = 0.f;
    for (int j = 0; j < d; ++j) sum += expf(Gg[i * d + j] / (nqi * nk[j]) * s1 - mx);
    const float inv = 1.f / sum;
    for (int j = 0; j < d; ++j) At[(int64_t)g * d * d + j * d + i] = expf(Gg[i * d + j] / (nqi * nk[j]) * s1 - mx) * inv;
    for (int c = 0; c < p; ++c) Mq[(int64_t)g * d * p + i * p + c] = KP[(int64_t)g * d * p + i * p + c] / nqi * s2;
}

// backward: dG, dGq, dGk (zero off the diagonal), dKP, and per-(b, h) partials of dt1 / dt2 (the caller sums them over b)
__global__ void __launch_bounds__(64) epa_small_bwd_kernel(const float* __restrict__ G, const float* __restrict__ Gq, const float* __restrict__ Gk,
                                                           const float* __restrict__ KP, const float* __restrict__ t1, const float* __restrict__ t2,
                                                           const float* __restrict__ At, const float* __restrict__ nq_in, const float* __restrict__ nk_in,
                                                           const float* __restrict__ dAt, const float* __restrict__ dMq, float* __restrict__ dG,
                                                           float* __restrict__ dGq, float* __restrict__ dGk, float* __restrict__ dKP,
                                                           float* __restrict__ dt1_part, float* __restrict__ dt2_part, int heads, int d, int p,
                                                           int dstride) {
    __shared__ float nk[64], col[64][65], red1[64], red2[64];
    const int g = blockIdx.x, h = g % heads, i = threadIdx.x;
    const int64_t o2 = (int64_t)g * d * d, op = (int64_t)g * d * p;
    if (i < d) nk[i] = nk_in[(int64_t)g * d + i];
    __syncthreads();
    const float s1 = t1[h], s2 = t2[h];
    float a1 = 0.f, a2 = 0.f, dnq = 0.f;
    const float nqi = i < d ? nq_in[(int64_t)g * d + i] : 1.f;
    if (i < d) {
        float dot = 0.f;                                    // sum_j dA_ij A_ij
        for (int j = 0; j < d; ++j) dot += dAt[o2 + j * d + i] * At[o2 + j * d + i];
        for (int j = 0; j < d; ++j) {
            const float a = At[o2 + j * d + i];
            const float dz = a * (dAt[o2 + j * d + i] - dot);          // softmax backward
            const float r = G[o2 + i * d + j] / (nqi * nk[j]);
            a1 += dz * r;                                   // dt1
            const float dr = dz * s1;
            dG[o2 + i * d + j] = dr / (nqi * nk[j]);
            dnq -= dr * r / nqi;
            col[i][j] = -dr * r;                            // contribution to dnk_j (/ nk_j below), summed over i by thread j
        }
        for (int c = 0; c < p; ++c) {
            const float dm = dMq[op + i * p + c], kp = KP[op + i * p + c];
            dKP[op + i * p + c] = dm * s2 / nqi;
            a2 += dm * kp / nqi;                            // dt2
            dnq -= dm * kp * s2 / (nqi * nqi);
        }
    }
    red1[i] = a1;
    red2[i] = a2;
    __syncthreads();
    if (i < d) {
        float dnk = 0.f;
        for (int r2 = 0; r2 < d; ++r2) dnk += col[r2][i];
        dnk /= nk[i];
        // n = max(sqrt(max(x, 0)), eps): dn/dx = 1 / (2 sqrt(x)) where x > 0 and sqrt(x) > eps, else 0
        const int64_t od = (int64_t)g * d * dstride + i * dstride;
        const float xq = Gq[od + (dstride > 1 ? i : 0)], xk = Gk[od + (dstride > 1 ? i : 0)];
        const float gq = (xq > 0.f && sqrtf(xq) > EPA_EPS) ? dnq * 0.5f / sqrtf(xq) : 0.f;
        const float gk = (xk > 0.f && sqrtf(xk) > EPA_EPS) ? dnk * 0.5f / sqrtf(xk) : 0.f;
        if (dstride > 1) {
            for (int j = 0; j < d; ++j) {
                dGq[o2 + i * d + j] = j == i ? gq : 0.f;
                dGk[o2 + i * d + j] = j == i ? gk : 0.f;
            }
        } else {
            dGq[od] = gq;
            dGk[od] = gk;
        }
    }
    if (i == 0) {
        float u = 0.f, v = 0.f;
        for (int r2 = 0; r2 < d; ++r2) { u += red1[r2]; v += red2[r2]; }
        dt1_part[g] = u;
        dt2_part[g] = v;
    }
}
}  // namespace ts
}  // namespace p4c

extern "C" int p4c_epa_small_fwd(const float* G, const float* Gq, const float* Gk, const float* KP, const float* t1, const float* t2, float* At,
                                 float* Mq, float* nq, float* nk, int B, int heads, int d, int p, int diag_only, p4c_stream_t stream) {
    P4C_CHECK_ARG(G && Gq && Gk && KP && t1 && t2 && At && Mq && nq && nk, "p4c_epa_small_fwd: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && d > 0 && d <= 64 && p > 0, "p4c_epa_small_fwd: head width 1..64 (got %d)", d);
    hipLaunchKernelGGL(ts::epa_small_fwd_kernel, dim3(B * heads), dim3(64), 0, as_stream(stream), G, Gq, Gk, KP, t1, t2, At, Mq, nq, nk, heads, d, p,
                       diag_only ? 1 : d);
    P4C_CHECK_LAUNCH("p4c_epa_small_fwd");
    return P4C_OK;
}

extern "C" int p4c_epa_small_bwd(const float* G, const float* Gq, const float* Gk, const float* KP, const float* t1, const float* t2,
                                 const float* At, const float* nq, const float* nk, const float* dAt, const float* dMq, float* dG, float* dGq,
                                 float* dGk, float* dKP, float* dt1_part, float* dt2_part, int B, int heads, int d, int p, int diag_only,
                                 p4c_stream_t stream) {
    P4C_CHECK_ARG(G && Gq && Gk && KP && t1 && t2 && At && nq && nk && dAt && dMq && dG && dGq && dGk && dKP && dt1_part && dt2_part,
                  "p4c_epa_small_bwd: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && d > 0 && d <= 64 && p > 0, "p4c_epa_small_bwd: head width 1..64 (got %d)", d);
    hipLaunchKernelGGL(ts::epa_small_bwd_kernel, dim3(B * heads), dim3(64), 0, as_stream(stream), G, Gq, Gk, KP, t1, t2, At, nq, nk, dAt, dMq, dG, dGq,
                       dGk, dKP, dt1_part, dt2_part, heads, d, p, diag_only ? 1 : d);
    P4C_CHECK_LAUNCH("p4c_epa_small_bwd");
    return P4C_OK;
}
