// Tall-skinny products for the efficient paired attention (EPA) of UNETR++ (config/CLI/model/unetrpp.yaml; mfai v5.0.1's UNetRPP, absent
// from the reference checkout).  Per (sample, head) group the attention works on N x d token matrices with N = H*W/16 ... H*W/1024
// tokens and d, e <= 64 columns; every product is either
//   gram :  C[g] (d x e) = X[g]^T Y[g]                   -- a reduction over the N tokens (q^T k, k^T W^T, dA = dXca^T v, ...)
//   apply:  O[g] (N x e) = X[g] (N x d) M[g] (d x e)      -- a per-token small matrix product (v A^T, q KP, S VP^T, ...)
// and the two are each other's adjoints, so forward AND backward of the block are these two kernels (py4cast_amd/ops_ts.py).
// Both are HBM-streaming passes over the token matrices (arithmetic intensity <= 2*min(d,e)/esz flop per byte, far below the
// ridge for the stages that matter: d = 8..32 at N = 16 384..1 024), operands addressed in place inside the (B, N, 4, heads, d)
// output of the qkvv projection / the (B, N, C) token tensor through (group, row) strides: no permute / contiguous copies.
// Two forms of each: fp32-exact VALU kernels on fp32 LDS tiles (gram_kernel / apply_kernel: the fp32 parity flavour, widths that are
// not multiples of 8, unaligned views; d, e <= 64 per launch) and, for bf16 token matrices, the matrix-core kernels further down
// (gram_mfma_kernel / apply_mfma_kernel: any width in one launch, the row softmax of the spatial branch and its adjoint as apply
// epilogues) -- per-shape times of both in profiles/r03_ts_micro.txt (tools/diagnostics/ts_micro.py).
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


// ---- apply on the matrix cores (bf16 token matrices) ------------------------------------------------------------------------
// O^T = M^T X^T per (sample, head): the A operand is M^T (rows = output columns), the B operand X^T (columns = tokens), so a lane of
// v_mfma_f32_32x32x16_bf16 needs eight consecutive reduction indices of ONE token -- a single 16-byte load of its row of X straight
// from HBM, no staging of the token matrix at all -- and ends up with 4 consecutive output columns of its token per 8-row block of the
// accumulator; the two lanes of a token trade halves (v_permlane32_swap) and store 16 bytes each.  M^T of the workgroup's heads
// (64 output columns x d, bf16) sits in LDS.  A workgroup = 4 waves = `hw` heads x 4/hw token tiles of 32; blockIdx.z = (head group,
// 64-column block of the output).  KC = 16-wide reduction chunks (d <= 16 KC): every load of a tile is in flight before its first MFMA.
typedef float ts_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 ts_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ts_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ts_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ts_u32x2 __attribute__((ext_vector_type(2)));

// EPI (e <= 64: a token's whole output row sits in the two lanes of its token): 1 = row softmax of the products before they are stored
// (S = softmax(q Mq): the logits never reach memory); 2 = softmax backward: the products are dS, the stored values
// S (dS - sum_j dS_j S_j) with S read in the accumulator layout -- the spatial branch of EPA without the N x p softmax passes.
template <typename TO, int KC, int EPI = 0>
__global__ void __launch_bounds__(256) apply_mfma_kernel(Mat X, const float* __restrict__ M, int64_t m_bs, int64_t m_hs, MatOut O, int heads,
                                                         int64_t N, int d, int e, int accumulate, int hw, Mat S, int mt) {
    // mt: M is given TRANSPOSED in memory ((e x d) row-major per group: element (k, c) at c * d + k) -- the adjoint applies of a block
    // multiply with At^T, Mq^T, dG^T, VP^T; reading them as stored saves the small strided copy each one cost
    extern __shared__ __attribute__((aligned(16))) unsigned short lmt[];          // hw x 64 x ldm bf16: M^T of the heads served
    constexpr int DP = KC * 16, LDM = DP + 8;
    const int b = blockIdx.x, ncp = (e + 63) >> 6;
    const int cp = blockIdx.z % ncp, h0 = (blockIdx.z / ncp) * hw;
    const int nh = (heads - h0) < hw ? (heads - h0) : hw;
    const int c0 = cp * 64;
    const int ncols = (e - c0) < 64 ? (e - c0) : 64;                               // multiple of 8
    const bool two = ncols > 32;
    // (rows of the image beyond ncols are never multiplied with anything stored: only the live column quads are staged)
    const int cq = (ncols + 3) >> 2, cqa = two ? 16 : (ncols > 0 ? 8 : 0);        // live quads; quads read by the MFMAs (32 / 64 image rows)
    for (int i = threadIdx.x; i < hw * DP * cqa; i += 256) {
        const int c4 = (i % cqa) << 2, r = i / cqa, k = r % DP, hh = r / DP;
        p4c_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (hh < nh && k < d && (c4 >> 2) < cq) {
            const float* mg = M + b * m_bs + (int64_t)(h0 + hh) * m_hs;
            if (!mt) v = load4f(mg + (int64_t)k * e + c0 + c4);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (c0 + c4 + j < e) ? mg[(int64_t)(c0 + c4 + j) * d + k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 t = (__bf16)v[j];
            lmt[(hh * 64 + c4 + j) * LDM + k] = __builtin_bit_cast(unsigned short, t);
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hh = wave % hw, tg = wave / hw, ntg = 4 / hw;
    if (hh >= nh) return;
    const int r32 = lane & 31, kg = lane >> 5;
    const bf16* xb = reinterpret_cast<const bf16*>(X.base) + b * X.bs + (int64_t)(h0 + hh) * X.hs + kg * 8;
    TO* ob = reinterpret_cast<TO*>(O.base) + b * O.bs + (int64_t)(h0 + hh) * O.hs + c0;
    const unsigned short* la = lmt + (hh * 64 + r32) * LDM + kg * 8;
    const int64_t ntiles = (N + 31) >> 5;
    for (int64_t tile = (int64_t)blockIdx.y * ntg + tg; tile < ntiles; tile += (int64_t)gridDim.y * ntg) {
        const int64_t n = tile * 32 + r32;
        const bool live = n < N;
        const bf16* xr = xb + n * X.rs;
        ts_bf16x8 xv[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            ts_u32x4 u = {0u, 0u, 0u, 0u};
            if (live && kc * 16 + kg * 8 < d) u = *reinterpret_cast<const ts_u32x4*>(xr + kc * 16);
            xv[kc] = __builtin_bit_cast(ts_bf16x8, u);
        }
        ts_f32x16 acc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[q][j] = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const ts_bf16x8 a0 = __builtin_bit_cast(ts_bf16x8, *reinterpret_cast<const ts_u32x4*>(la + kc * 16));
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, xv[kc], acc[0], 0, 0, 0);
            if (two) {
                const ts_bf16x8 a1 = __builtin_bit_cast(ts_bf16x8, *reinterpret_cast<const ts_u32x4*>(la + 32 * LDM + kc * 16));
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, xv[kc], acc[1], 0, 0, 0);
            }
        }
        // accumulator element 4 j + i of a lane = output column 8 j + 4 kg + i of its token
        TO* op = ob + n * O.rs;
        if constexpr (EPI == 1) {
            float mx = -INFINITY;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    if (q * 32 + 8 * (v >> 2) + 4 * kg + (v & 3) < ncols) mx = fmaxf(mx, acc[q][v]);
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const bool in = q * 32 + 8 * (v >> 2) + 4 * kg + (v & 3) < ncols;
                    const float ex = in ? __expf(acc[q][v] - mx) : 0.f;
                    acc[q][v] = ex;
                    sum += ex;
                }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.f / sum;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[q][v] *= inv;
        }
        if constexpr (EPI == 2) {
            const bf16* sp = reinterpret_cast<const bf16*>(S.base) + b * S.bs + (int64_t)(h0 + hh) * S.hs + n * S.rs + c0;
            float sv[2][16];
            float dot = 0.f;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = q * 32 + 8 * j + 4 * kg;
                    p4c_f32x4 t = {0.f, 0.f, 0.f, 0.f};
                    if (live && col < ncols) t = load4f(sp + col);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        sv[q][4 * j + i] = t[i];
                        dot = __builtin_fmaf(acc[q][4 * j + i], t[i], dot);
                    }
                }
            dot += __shfl_xor(dot, 32);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[q][v] = sv[q][v] * (acc[q][v] - dot);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q == 1 && !two) continue;
            if constexpr (sizeof(TO) == 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = q * 32 + 8 * j + 4 * kg;
                    if (live && col < ncols) {
                        p4c_f32x4 v = {acc[q][4 * j], acc[q][4 * j + 1], acc[q][4 * j + 2], acc[q][4 * j + 3]};
                        float* o4 = reinterpret_cast<float*>(op) + col;
                        if (accumulate) {
                            const p4c_f32x4 old = load4f(o4);
                            v[0] += old[0]; v[1] += old[1]; v[2] += old[2]; v[3] += old[3];
                        }
                        store4f(o4, v);
                    }
                }
            } else {
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2) {
                    const int col = q * 32 + 16 * m2 + 8 * kg;          // after the trade: 8 consecutive columns per lane
                    const bool st = live && col < ncols;
                    float lo[4], hi[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        lo[i] = acc[q][8 * m2 + i];
                        hi[i] = acc[q][8 * m2 + 4 + i];
                    }
                    if (accumulate) {
                        // read-modify-write without the trade (rare: the second apply of a gram_norms backward): 8-byte pieces, the
                        // lane's own 4 columns of the two 8-row blocks (fp32 values swapped between lanes came back wrong from the
                        // compiler here -- both results of v_permlane32_swap read as the first one)
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            const int c4 = q * 32 + 16 * m2 + 8 * half + 4 * kg;
                            if (live && c4 < ncols) {
                                bf16* o4 = reinterpret_cast<bf16*>(op) + c4;
                                const p4c_f32x4 old = load4f(o4);
                                const float* sv = half ? hi : lo;
                                store4f(o4, p4c_f32x4{sv[0] + old[0], sv[1] + old[1], sv[2] + old[2], sv[3] + old[3]});
                            }
                        }
                    } else {
                        ts_bf16x4 pa, pb;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { pa[i] = (__bf16)lo[i]; pb[i] = (__bf16)hi[i]; }
                        const ts_u32x2 ua = __builtin_bit_cast(ts_u32x2, pa), ub = __builtin_bit_cast(ts_u32x2, pb);
                        const ts_u32x2 s0 = __builtin_amdgcn_permlane32_swap(ua.x, ub.x, false, false);
                        const ts_u32x2 s1 = __builtin_amdgcn_permlane32_swap(ua.y, ub.y, false, false);
                        if (st) *reinterpret_cast<ts_u32x4*>(reinterpret_cast<bf16*>(op) + col) = ts_u32x4{s0.x, s1.x, s0.y, s1.y};
                    }
                }
            }
        }
    }
}


// ---- gram on the matrix cores (bf16 token matrices) -------------------------------------------------------------------------
// G = X^T Y reduces over the tokens, so BOTH operands of v_mfma_f32_32x32x16_bf16 are transposed reads (ds_read_b64_tr_b16) of
// [token][column] images in LDS, exactly as in a weight gradient (csrc/rowgemm.hip).  A workgroup = 2 waves serves one (sample, head),
// one token split and one 64 x 64 block of the result; each wave stages its own 32-token tiles (16-byte loads of the rows, in place in
// the strided views), the two waves add in wave order through LDS: one partial per (sample, split, head), summed by the caller in a
// fixed order -- bit-identical reruns.  NORMS: the column sums of squares of X and Y come from the loader's registers (a lane always
// stages the same 8 columns), reduced in lane order through LDS.
typedef short ts_s16x4 __attribute__((ext_vector_type(4)));
constexpr int GROWS = 32;                    // tokens per tile
constexpr int GROWB = 64 * 2 + 16;           // bytes per image row: 64 columns + padding (conflict-free transposed reads)

__device__ __forceinline__ ts_bf16x8 read_tr(const char* p) {
    union { ts_s16x4 s[2]; ts_bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ts_s16x4 __attribute__((address_space(3)))*)(p));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ts_s16x4 __attribute__((address_space(3)))*)(p + 4 * GROWB));
    return u.v;
}

template <int MT, int NT, bool NORMS>
__global__ void __launch_bounds__(128) gram_mfma_kernel(Mat X, Mat Y, float* __restrict__ part, int heads, int64_t N, int d, int e, int nsplit,
                                                        int pstride) {
    constexpr int IMG = GROWS * GROWB;
    constexpr int MAIN = 4 * IMG > 64 * 64 * 4 ? 4 * IMG : 64 * 64 * 4;             // the four images (4 x 4.5 KB), later the 64 x 64 sum
    __shared__ __attribute__((aligned(16))) char smem[MAIN + (NORMS ? 2 * 2 * 64 * 8 * 4 : 0)];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads, sp = blockIdx.y;
    const int eblocks = (e + 63) >> 6;
    const int i0 = 64 * (blockIdx.z / eblocks), j0 = 64 * (blockIdx.z % eblocks);
    const int dcols = (d - i0) < 64 ? (d - i0) : 64, ecols = (e - j0) < 64 ? (e - j0) : 64;
    const int dv = dcols >> 3, ev = ecols >> 3;                              // 16-byte vectors per staged row
    char* imgX = smem + wv * 2 * IMG;
    char* imgY = imgX + IMG;
    const bf16* xb = reinterpret_cast<const bf16*>(X.base) + b * X.bs + (int64_t)h * X.hs + i0;
    const bf16* yb = reinterpret_cast<const bf16*>(Y.base) + b * Y.bs + (int64_t)h * Y.hs + j0;
    for (int i = lane; i < 2 * IMG / 16; i += 64) reinterpret_cast<ts_u32x4*>(imgX)[i] = ts_u32x4{0u, 0u, 0u, 0u};   // padding columns stay zero
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");

    ts_f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
    float nx[8], ny[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) nx[j] = ny[j] = 0.f;

    const int64_t ntiles = (N + GROWS - 1) / GROWS;
    const int64_t tps = (ntiles + nsplit - 1) / nsplit;
    const int64_t tend = ((int64_t)(sp + 1) * tps) < ntiles ? ((int64_t)(sp + 1) * tps) : ntiles;
    const int i16 = lane & 15, tg = (lane >> 4) & 1, hh = lane >> 5;
    // a lane stages the same (row, 16-byte column) slots of every tile: up to four of X and four of Y (32 rows x <= 8 vectors / 64 lanes).
    // Round 6: the loads of the wave's NEXT tile are issued before the matrix phase of the current one (they used to start after it:
    // with two waves per workgroup and a few workgroups per CU nothing else covered their latency).
    int64_t gx[4], gy[4];
    int lx[4], ly[4], rx[4], ry[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = lane + 64 * i;
        const bool vx = v < GROWS * dv, vy = v < GROWS * ev;
        const int rrx = vx ? v / dv : 0, cx = vx ? v - rrx * dv : 0;
        const int rry = vy ? v / ev : 0, cy = vy ? v - rry * ev : 0;
        rx[i] = vx ? rrx : -1;
        ry[i] = vy ? rry : -1;
        gx[i] = (int64_t)rrx * X.rs + 8 * cx;
        gy[i] = (int64_t)rry * Y.rs + 8 * cy;
        lx[i] = rrx * GROWB + cx * 16;
        ly[i] = rry * GROWB + cy * 16;
    }
    ts_u32x4 qx[4], qy[4];
    auto load_tile = [&](int64_t t) {
        const int64_t row0 = t * GROWS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qx[i] = ts_u32x4{0u, 0u, 0u, 0u};
            qy[i] = ts_u32x4{0u, 0u, 0u, 0u};
            if (rx[i] >= 0 && row0 + rx[i] < N) qx[i] = *reinterpret_cast<const ts_u32x4*>(xb + row0 * X.rs + gx[i]);
            if (ry[i] >= 0 && row0 + ry[i] < N) qy[i] = *reinterpret_cast<const ts_u32x4*>(yb + row0 * Y.rs + gy[i]);
        }
    };
    int64_t t = (int64_t)sp * tps + wv;
    if (t < tend) load_tile(t);
    for (; t < tend; t += 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (rx[i] >= 0) {
                *reinterpret_cast<ts_u32x4*>(imgX + lx[i]) = qx[i];
                if (NORMS) {
                    const ts_bf16x8 f = __builtin_bit_cast(ts_bf16x8, qx[i]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) nx[j] = __builtin_fmaf((float)f[j], (float)f[j], nx[j]);
                }
            }
            if (ry[i] >= 0) {
                *reinterpret_cast<ts_u32x4*>(imgY + ly[i]) = qy[i];
                if (NORMS) {
                    const ts_bf16x8 f = __builtin_bit_cast(ts_bf16x8, qy[i]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ny[j] = __builtin_fmaf((float)f[j], (float)f[j], ny[j]);
                }
            }
        }
        if (t + 2 < tend) load_tile(t + 2);
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < GROWS / 16; ++ks) {
            // lane (column 32 tile + (lane & 31), half hh) receives tokens 16 ks + 8 hh + j, j = 0..7, of its column
            const int rbase = 16 * ks + 8 * hh + (i16 >> 2);
            const int coff = (tg * 16 + (i16 & 3) * 4) * 2;
            ts_bf16x8 av[MT], bv[NT];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[m] = read_tr(imgX + rbase * GROWB + 64 * m + coff);
#pragma unroll
            for (int n = 0; n < NT; ++n) bv[n] = read_tr(imgY + rbase * GROWB + 64 * n + coff);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
    }

    // the two waves add in wave order through LDS: red[i][j], 64 x 64 floats over the images
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    float* nbuf = reinterpret_cast<float*>(smem + MAIN);                     // [X | Y][wave][lane][8]
    const int r = lane & 31;
    for (int turn = 0; turn < 2; ++turn) {
        if (wv == turn) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int o = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * hh, k = 32 * n + r;
                        if (turn == 0) red[o * 64 + k] = acc[m][n][i];
                        else red[o * 64 + k] += acc[m][n][i];
                    }
        }
        __syncthreads();
    }
    if (NORMS) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            nbuf[(wv * 64 + lane) * 8 + j] = nx[j];
            nbuf[((2 + wv) * 64 + lane) * 8 + j] = ny[j];
        }
        __syncthreads();
    }
    float* pb = part + (((int64_t)b * nsplit + sp) * heads + h) * pstride;
    for (int i = threadIdx.x; i < dcols * ecols; i += 128) {
        const int ii = i / ecols, jj = i - ii * ecols;
        pb[(int64_t)(i0 + ii) * e + j0 + jj] = red[ii * 64 + jj];
    }
    if (NORMS) {
        // column 8 c + j was staged by the lanes with lane % dv == c (64 % dv == 0: a lane keeps its columns from tile to tile)
        for (int i = threadIdx.x; i < dcols + ecols; i += 128) {
            const bool isx = i < dcols;
            const int col = isx ? i : i - dcols, nv = isx ? dv : ev;
            const int c = col >> 3, j = col & 7;
            const float* src = nbuf + (isx ? 0 : 2 * 64 * 8);
            float sacc = 0.f;
            for (int w = 0; w < 2; ++w)
                for (int l = c; l < 64; l += nv) sacc += src[(w * 64 + l) * 8 + j];
            pb[(int64_t)d * e + (isx ? i0 : d + j0) + col] = sacc;      // (every result block of a column range writes the same sums)
        }
    }
}

}  // namespace ts
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_ts_gram_splits(int64_t N) {
    // tokens per split (P4C_TS_SPLIT_TOKENS; from 4 096 tokens 256 = eight 32-token tiles: four per wave of the matrix-core kernel, whose
    // per-workgroup costs -- clearing the images, the LDS sum of the two waves, one partial per result block -- a single tile per wave
    // did not amortise; with (sample, head) groups in blockIdx.x the launches still have hundreds of workgroups).
    static const int forced = [] { const char* v = diag_env("P4C_TS_SPLIT_TOKENS"); const int n = v ? atoi(v) : 0; return n > 0 && n < 32 ? 32 : n; }();
    const int per = forced ? forced : (N >= 4096 ? 256 : 64);      // (measured per stage, profiles/r03_ts_micro.txt: the short stages keep 64)
    int64_t s = (N + per - 1) / per;
    if (s > 256) s = 256;
    if (s < 1) s = 1;
    return (int)s;
}

static int pow2_ge_i(int v) { int p = 1; while (p < v) p <<= 1; return p; }

// the matrix-core form serves bf16 x bf16 with d, e multiples of 8 (any width: 64 x 64 result blocks in blockIdx.z)
extern "C" int p4c_ts_gram_wide_ok(int x_dtype, int y_dtype, int d, int e) {
    static const bool off = [] { const char* v = diag_env("P4C_TS_NO_MFMA"); return v && v[0] == '1'; }();
    return !off && x_dtype == P4C_BF16 && y_dtype == P4C_BF16 && d > 0 && e > 0 && d % 8 == 0 && e % 8 == 0;
}

static bool gram_mfma_aligned(const void* x, int64_t x_bs, int64_t x_hs, int64_t x_rs, const void* y, int64_t y_bs, int64_t y_hs, int64_t y_rs) {
    return x_bs % 8 == 0 && x_hs % 8 == 0 && x_rs % 8 == 0 && y_bs % 8 == 0 && y_hs % 8 == 0 && y_rs % 8 == 0 &&
           (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0;
}

template <bool NORMS>
static void launch_gram_mfma(const ts::Mat& X, const ts::Mat& Y, float* partial, int B, int heads, int64_t N, int d, int e, int ns, hipStream_t st) {
    const dim3 grid(B * heads, ns, ((d + 63) / 64) * ((e + 63) / 64));
    const int pstride = d * e + (NORMS ? d + e : 0);
    const bool m2 = d > 32, n2 = e > 32;
#define P4C_GRAM_M(MT, NT) hipLaunchKernelGGL((ts::gram_mfma_kernel<MT, NT, NORMS>), grid, dim3(128), 0, st, X, Y, partial, heads, N, d, e, ns, pstride)
    if (m2 && n2) P4C_GRAM_M(2, 2);
    else if (m2) P4C_GRAM_M(2, 1);
    else if (n2) P4C_GRAM_M(1, 2);
    else P4C_GRAM_M(1, 1);
#undef P4C_GRAM_M
}

extern "C" int p4c_ts_gram(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const void* y, int y_dtype,
                           int64_t y_bs, int64_t y_hs, int64_t y_rs, float* partial, int B, int heads, int64_t N, int d, int e,
                           p4c_stream_t stream) {
    P4C_CHECK_ARG(x && y && partial, "p4c_ts_gram: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && N > 0, "p4c_ts_gram: empty problem");
    if (p4c_ts_gram_wide_ok(x_dtype, y_dtype, d, e) && gram_mfma_aligned(x, x_bs, x_hs, x_rs, y, y_bs, y_hs, y_rs)) {
        const ts::Mat Xm{x, x_bs, x_hs, x_rs}, Ym{y, y_bs, y_hs, y_rs};
        launch_gram_mfma<false>(Xm, Ym, partial, B, heads, N, d, e, p4c_ts_gram_splits(N), as_stream(stream));
        P4C_CHECK_LAUNCH("p4c_ts_gram");
        return P4C_OK;
    }
    P4C_CHECK_ARG(d > 0 && e > 0 && d <= ts::MAXD && e <= ts::MAXD && d % 4 == 0 && e % 4 == 0,
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
    P4C_CHECK_ARG(B > 0 && heads > 0 && N > 0 && d > 0 && e > 0, "p4c_ts_gram_norms: empty problem");
    P4C_CHECK_ARG(x_dtype == P4C_BF16 && y_dtype == P4C_BF16, "p4c_ts_gram_norms: bf16 token matrices");
    // matrix-core form: a lane keeps its 8 columns from tile to tile when the 16-byte vectors of a staged row divide 64 lanes -- widths
    // 8, 16, 32, 64, or (round 6) whole 64-column result blocks (128-wide decoder heads)
    const auto lanes_ok = [](int w) { return w % 8 == 0 && (w <= 64 ? 64 % (w / 8) == 0 : w % 64 == 0); };
    if (p4c_ts_gram_wide_ok(x_dtype, y_dtype, d, e) && lanes_ok(d) && lanes_ok(e) && gram_mfma_aligned(x, x_bs, x_hs, x_rs, y, y_bs, y_hs, y_rs)) {
        const ts::Mat Xm{x, x_bs, x_hs, x_rs}, Ym{y, y_bs, y_hs, y_rs};
        launch_gram_mfma<true>(Xm, Ym, partial, B, heads, N, d, e, p4c_ts_gram_splits(N), as_stream(stream));
        P4C_CHECK_LAUNCH("p4c_ts_gram_norms");
        return P4C_OK;
    }
    P4C_CHECK_ARG(d <= ts::MAXD && e <= ts::MAXD && d % 4 == 0 && e % 4 == 0,
                  "p4c_ts_gram_norms: off the matrix-core form d, e must be multiples of 4 up to %d (got %d, %d)", ts::MAXD, d, e);
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

// the matrix-core form serves bf16 token matrices with d, e multiples of 8 and d <= 256 (any e): no 64-column chunking by the caller
extern "C" int p4c_ts_apply_wide_ok(int x_dtype, int out_dtype, int d, int e) {
    static const bool off = [] { const char* v = diag_env("P4C_TS_NO_MFMA"); return v && v[0] == '1'; }();
    return !off && x_dtype == P4C_BF16 && (out_dtype == P4C_BF16 || out_dtype == P4C_F32) && d > 0 && e > 0 && d % 8 == 0 && e % 8 == 0 && d <= 256;
}

template <typename TO, int EPI = 0>
static void launch_apply_mfma(const ts::Mat& X, const float* m, int64_t m_bs, int64_t m_hs, const ts::MatOut& O, int B, int heads, int64_t N,
                              int d, int e, int accumulate, hipStream_t st, const ts::Mat& S = ts::Mat{nullptr, 0, 0, 0}, int mt = 0) {
    const int kc = d <= 16 ? 1 : d <= 32 ? 2 : d <= 64 ? 4 : d <= 128 ? 8 : 16;
    int hw = heads >= 4 ? 4 : heads >= 2 ? 2 : 1;
    while (hw > 1 && hw * 64 * (kc * 16 + 8) * 2 > 65536) hw >>= 1;      // M^T of the heads a workgroup serves: <= 64 KB of LDS
    const size_t lds = (size_t)hw * 64 * (kc * 16 + 8) * 2;
    const int ntg = 4 / hw, zc = ((heads + hw - 1) / hw) * ((e + 63) / 64);
    const int64_t ntiles = (N + 31) / 32;
    int64_t ny = (ntiles + ntg - 1) / ntg;
    static const int wgs_per_cu = [] { const char* v = diag_env("P4C_TS_APPLY_WGS"); return v ? atoi(v) : 4; }();
    const int64_t cap = (int64_t)num_cus() * wgs_per_cu / ((int64_t)B * zc) + 1;     // several tiles per wave: M^T is staged once per workgroup
    if (ny > cap) ny = cap;
    const dim3 grid(B, (unsigned)ny, zc);
#define P4C_APPLY_M(KC) hipLaunchKernelGGL((ts::apply_mfma_kernel<TO, KC, EPI>), grid, dim3(256), lds, st, X, m, m_bs, m_hs, O, heads, N, d, e, accumulate, hw, S, mt)
    switch (kc) {
        case 1: P4C_APPLY_M(1); break;
        case 2: P4C_APPLY_M(2); break;
        case 4: P4C_APPLY_M(4); break;
        case 8: P4C_APPLY_M(8); break;
        default: P4C_APPLY_M(16); break;
    }
#undef P4C_APPLY_M
}

static bool apply_mfma_route(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m, int64_t m_gs, const void* out,
                             int out_dtype, int64_t o_bs, int64_t o_hs, int64_t o_rs, int d, int e) {
    return p4c_ts_apply_wide_ok(x_dtype, out_dtype, d, e) && x_bs % 8 == 0 && x_hs % 8 == 0 && x_rs % 8 == 0 && o_bs % 8 == 0 && o_hs % 8 == 0 &&
           o_rs % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
           (reinterpret_cast<uintptr_t>(m) & 15) == 0 && m_gs % 4 == 0;
}

extern "C" int p4c_ts_apply_mt_ok(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m, int64_t m_gs,
                                  const void* out, int out_dtype, int64_t o_bs, int64_t o_hs, int64_t o_rs, int d, int e) {
    return apply_mfma_route(x, x_dtype, x_bs, x_hs, x_rs, m, m_gs, out, out_dtype, o_bs, o_hs, o_rs, d, e) ? 1 : 0;
}

extern "C" int p4c_ts_apply_mt(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m_t, int64_t m_gs,
                               void* out, int out_dtype, int64_t o_bs, int64_t o_hs, int64_t o_rs, int B, int heads, int64_t N, int d, int e,
                               int accumulate, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && m_t && out, "p4c_ts_apply_mt: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && N > 0, "p4c_ts_apply_mt: empty problem");
    if (!apply_mfma_route(x, x_dtype, x_bs, x_hs, x_rs, m_t, m_gs, out, out_dtype, o_bs, o_hs, o_rs, d, e))
        return fail(P4C_ERR_UNSUPPORTED, "p4c_ts_apply_mt: operands off the matrix-core kernel's conditions (p4c_ts_apply_mt_ok)");
    const ts::Mat Xm{x, x_bs, x_hs, x_rs};
    const ts::MatOut Om{out, o_bs, o_hs, o_rs};
    const ts::Mat none{nullptr, 0, 0, 0};
    if (out_dtype == P4C_BF16) launch_apply_mfma<bf16>(Xm, m_t, m_gs * heads, m_gs, Om, B, heads, N, d, e, accumulate, as_stream(stream), none, 1);
    else launch_apply_mfma<float>(Xm, m_t, m_gs * heads, m_gs, Om, B, heads, N, d, e, accumulate, as_stream(stream), none, 1);
    P4C_CHECK_LAUNCH("p4c_ts_apply_mt");
    return P4C_OK;
}

extern "C" int p4c_ts_apply(const void* x, int x_dtype, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m, int64_t m_gs,
                            void* out, int out_dtype, int64_t o_bs, int64_t o_hs, int64_t o_rs, int B, int heads, int64_t N, int d, int e,
                            int accumulate, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && m && out, "p4c_ts_apply: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && N > 0, "p4c_ts_apply: empty problem");
    if (apply_mfma_route(x, x_dtype, x_bs, x_hs, x_rs, m, m_gs, out, out_dtype, o_bs, o_hs, o_rs, d, e)) {
        const ts::Mat Xm{x, x_bs, x_hs, x_rs};
        const ts::MatOut Om{out, o_bs, o_hs, o_rs};
        if (out_dtype == P4C_BF16) launch_apply_mfma<bf16>(Xm, m, m_gs * heads, m_gs, Om, B, heads, N, d, e, accumulate, as_stream(stream));
        else launch_apply_mfma<float>(Xm, m, m_gs * heads, m_gs, Om, B, heads, N, d, e, accumulate, as_stream(stream));
        P4C_CHECK_LAUNCH("p4c_ts_apply");
        return P4C_OK;
    }
    P4C_CHECK_ARG(d > 0 && e > 0 && d <= ts::MAXD && e <= ts::MAXD && d % 4 == 0 && e % 4 == 0,
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

// apply with a fused epilogue over the token's output row (bf16, matrix-core form only; e <= 64):
//   epi = 1: out = softmax_row(X M)                                     (s unused)
//   epi = 2: out = S * (X M - rowsum(X M * S)), S (B, heads, N, e) bf16  (the softmax backward of the products X M = dS)
extern "C" int p4c_ts_apply_softmax(const void* x, int64_t x_bs, int64_t x_hs, int64_t x_rs, const float* m, int64_t m_gs, void* out, int64_t o_bs,
                                    int64_t o_hs, int64_t o_rs, int B, int heads, int64_t N, int d, int e, int epi, const void* s, int64_t s_bs,
                                    int64_t s_hs, int64_t s_rs, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && m && out && B > 0 && heads > 0 && N > 0, "p4c_ts_apply_softmax: null pointer / empty problem");
    P4C_CHECK_ARG(epi == 1 || (epi == 2 && s), "p4c_ts_apply_softmax: epi must be 1 (softmax) or 2 (softmax backward, with S)");
    P4C_CHECK_ARG(p4c_ts_apply_wide_ok(P4C_BF16, P4C_BF16, d, e) && e <= 64, "p4c_ts_apply_softmax: bf16, d, e multiples of 8, d <= 256, e <= 64 (got %d, %d)", d, e);
    P4C_CHECK_ARG(x_bs % 8 == 0 && x_hs % 8 == 0 && x_rs % 8 == 0 && o_bs % 8 == 0 && o_hs % 8 == 0 && o_rs % 8 == 0 && m_gs % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (reinterpret_cast<uintptr_t>(m) & 15) == 0,
                  "p4c_ts_apply_softmax: strides must be multiples of 8 elements, bases 16-byte aligned");
    const ts::Mat Xm{x, x_bs, x_hs, x_rs};
    const ts::MatOut Om{out, o_bs, o_hs, o_rs};
    if (epi == 1) {
        launch_apply_mfma<bf16, 1>(Xm, m, m_gs * heads, m_gs, Om, B, heads, N, d, e, 0, as_stream(stream));
    } else {
        P4C_CHECK_ARG(s_bs % 4 == 0 && s_hs % 4 == 0 && s_rs % 4 == 0 && (reinterpret_cast<uintptr_t>(s) & 7) == 0, "p4c_ts_apply_softmax: S strides must be multiples of 4 elements");
        const ts::Mat Sm{s, s_bs, s_hs, s_rs};
        launch_apply_mfma<bf16, 2>(Xm, m, m_gs * heads, m_gs, Om, B, heads, N, d, e, 0, as_stream(stream), Sm);
    }
    P4C_CHECK_LAUNCH("p4c_ts_apply_softmax");
    return P4C_OK;
}

// ---------------------------------------------------------------------------------------------
// The small matrices of one EPA block (UNETR++ efficient paired attention; py4cast_amd/unetrpp.py::EPA) in ONE launch each way.
// Per (sample, head), with G = q^T k, Gq = q^T q, Gk = k^T k (d x d, from p4c_ts_gram) and KP (d x p):
//   nq_i = max(sqrt(max(Gq_ii, 0)), eps), nk_j likewise;   A = softmax_j(t1 G_ij / (nq_i nk_j));   Mq_ic = t2 KP_ic / nq_i
// forward writes At = A^T (what the apply kernel multiplies v with), Mq, nq, nk.  As torch ops this was ~14 launches forward and ~35
// backward per block (clamps, square roots, outer products, broadcast reductions ...), ~6 000 per UNetRPP training step.
// One workgroup per (b, h); everything in fp32, sums in a fixed order (deterministic).
namespace p4c {
namespace ts {
constexpr float EPA_EPS = 1e-12f;   // F.normalize's clamp

// Workgroup = 256 threads = row i (threadIdx.x & 63) x slice s (threadIdx.x >> 6): a slice takes the columns j = s (mod 4) of its row's
// inner loops (and the projection columns c = s (mod 4)); the four slices' partial maxima / sums meet in LDS and are added in slice
// order (deterministic).  G (d x d, row-major) and KP are staged through LDS with coalesced loads: a thread per row walking its
// row in global memory touched 64 cache lines per instruction (the first version: one 64-thread wave per (b, h), 20 / 36 us per launch).
// Round 6: the head width is a template parameter (DM = 64: 64 rows x 4 slices; DM = 128: 128 rows x 2 slices -- the decoder stage with
// 128-channel heads ran on separate tensor-library nodes before); the tiles live in dynamic LDS (DM = 128: 101 / 136 KB).
__device__ __forceinline__ void epa_stage(float* dst, int ld, const float* __restrict__ src, int rows, int cols) {
    for (int e = threadIdx.x; e < rows * cols; e += 256) {
        const int r = e / cols, c = e - r * cols;
        dst[r * ld + c] = src[e];
    }
}

template <int DM>
__global__ void __launch_bounds__(256) epa_small_fwd_kernel(const float* __restrict__ G, const float* __restrict__ Gq, const float* __restrict__ Gk,
                                                            const float* __restrict__ KP, const float* __restrict__ t1, const float* __restrict__ t2,
                                                            float* __restrict__ At, float* __restrict__ Mq, float* __restrict__ nq_out,
                                                            float* __restrict__ nk_out, int heads, int d, int p, int dstride) {
    // dstride = d: Gq / Gk are the full (d x d) q^T q / k^T k, only their diagonals are read;  dstride = 1: they ARE the diagonals (B, h, d)
    constexpr int LD = DM + 1, NSL = 256 / DM, KLD = 65;
    extern __shared__ float epa_sm[];
    float* nq = epa_sm;                 // [DM]
    float* nk = nq + DM;                // [DM]
    float* gl = nk + DM;                // [DM][LD]
    float* kl = gl + DM * LD;           // [DM][KLD]
    float* part = kl + DM * KLD;        // [NSL][DM]
    const int g = blockIdx.x, h = g % heads, i = threadIdx.x % DM, sl = threadIdx.x / DM;
    epa_stage(gl, LD, G + (int64_t)g * d * d, d, d);
    epa_stage(kl, KLD, KP + (int64_t)g * d * p, d, p);
    if (threadIdx.x < d) {
        const int64_t od = (int64_t)g * d * dstride + i * dstride + (dstride > 1 ? i : 0);
        const float a = fmaxf(sqrtf(fmaxf(Gq[od], 0.f)), EPA_EPS), c = fmaxf(sqrtf(fmaxf(Gk[od], 0.f)), EPA_EPS);
        nq[i] = a;
        nk[i] = c;
        nq_out[(int64_t)g * d + i] = a;
        nk_out[(int64_t)g * d + i] = c;
    }
    __syncthreads();
    const bool row = i < d;
    const float s1 = t1[h], s2 = t2[h];
    const float nqi = row ? nq[i] : 1.f;
    float mx = -INFINITY;
    if (row)
        for (int j = sl; j < d; j += NSL) mx = fmaxf(mx, gl[i * LD + j] / (nqi * nk[j]) * s1);
    part[sl * DM + i] = mx;
    __syncthreads();
    mx = part[i];
#pragma unroll
    for (int t = 1; t < NSL; ++t) mx = fmaxf(mx, part[t * DM + i]);
    __syncthreads();
    float sum = 0.f;
    if (row)
        for (int j = sl; j < d; j += NSL) sum += expf(gl[i * LD + j] / (nqi * nk[j]) * s1 - mx);
    part[sl * DM + i] = sum;
    __syncthreads();
    if (row) {
        float tot = part[i];
#pragma unroll
        for (int t = 1; t < NSL; ++t) tot += part[t * DM + i];
        const float inv = 1.f / tot;
        for (int j = sl; j < d; j += NSL) At[(int64_t)g * d * d + j * d + i] = expf(gl[i * LD + j] / (nqi * nk[j]) * s1 - mx) * inv;
    }
    // Mq rows: coalesced over c (thread -> column)
    for (int e = threadIdx.x; e < d * p; e += 256) {
        const int r = e / p, c = e - r * p;
        Mq[(int64_t)g * d * p + e] = kl[r * KLD + c] / nq[r] * s2;
    }
}

// backward: dG, dGq, dGk (zero off the diagonal), dKP, and per-(b, h) partials of dt1 / dt2 (the caller sums them over b)
template <int DM>
__global__ void __launch_bounds__(256) epa_small_bwd_kernel(const float* __restrict__ G, const float* __restrict__ Gq, const float* __restrict__ Gk,
                                                            const float* __restrict__ KP, const float* __restrict__ t1, const float* __restrict__ t2,
                                                            const float* __restrict__ At, const float* __restrict__ nq_in, const float* __restrict__ nk_in,
                                                            const float* __restrict__ dAt, const float* __restrict__ dMq, float* __restrict__ dG,
                                                            float* __restrict__ dGq, float* __restrict__ dGk, float* __restrict__ dKP,
                                                            float* __restrict__ dt1_part, float* __restrict__ dt2_part, int heads, int d, int p,
                                                            int dstride) {
    constexpr int LD = DM + 1, NSL = 256 / DM;
    extern __shared__ float epa_sm[];
    float* nq = epa_sm;                 // [DM]
    float* nk = nq + DM;                // [DM]
    float* gl = nk + DM;                // [DM][LD]
    float* col = gl + DM * LD;          // [DM][LD]
    float* pa = col + DM * LD;          // [NSL][DM]
    float* pb = pa + NSL * DM;
    float* pc = pb + NSL * DM;
    const int g = blockIdx.x, h = g % heads, i = threadIdx.x % DM, sl = threadIdx.x / DM;
    const int64_t o2 = (int64_t)g * d * d, op = (int64_t)g * d * p;
    epa_stage(gl, LD, G + o2, d, d);
    if (threadIdx.x < d) {
        nq[i] = nq_in[(int64_t)g * d + i];
        nk[i] = nk_in[(int64_t)g * d + i];
    }
    __syncthreads();
    const bool row = i < d;
    const float s1 = t1[h], s2 = t2[h];
    const float nqi = row ? nq[i] : 1.f;
    float dot = 0.f;                                        // sum_j dA_ij A_ij
    if (row)
        for (int j = sl; j < d; j += NSL) dot += dAt[o2 + j * d + i] * At[o2 + j * d + i];
    pa[sl * DM + i] = dot;
    __syncthreads();
    dot = pa[i];
#pragma unroll
    for (int t = 1; t < NSL; ++t) dot += pa[t * DM + i];
    __syncthreads();
    float a1 = 0.f, a2 = 0.f, dnq = 0.f;
    if (row) {
        for (int j = sl; j < d; j += NSL) {
            const float a = At[o2 + j * d + i];
            const float dz = a * (dAt[o2 + j * d + i] - dot);          // softmax backward
            const float r = gl[i * LD + j] / (nqi * nk[j]);
            a1 += dz * r;                                   // dt1
            const float dr = dz * s1;
            gl[i * LD + j] = dr / (nqi * nk[j]);            // dG_ij (the element is this thread's own: read above, stored below)
            dnq -= dr * r / nqi;
            col[i * LD + j] = -dr * r;                      // contribution to dnk_j (/ nk_j below), summed over i
        }
        for (int c = sl; c < p; c += NSL) {                 // (d x p: a row per thread, its cache lines shared by the row's slices)
            const float dm = dMq[op + i * p + c], kp = KP[op + i * p + c];
            dKP[op + i * p + c] = dm * s2 / nqi;
            a2 += dm * kp / nqi;                            // dt2
            dnq -= dm * kp * s2 / (nqi * nqi);
        }
    }
    pa[sl * DM + i] = a1;
    pb[sl * DM + i] = a2;
    pc[sl * DM + i] = dnq;
    __syncthreads();
    for (int e = threadIdx.x; e < d * d; e += 256) {        // coalesced stores of the staged results
        const int r = e / d, c = e - r * d;
        dG[o2 + e] = gl[r * LD + c];
    }
    if (row && sl == 0) {
        dnq = pc[i];
#pragma unroll
        for (int t = 1; t < NSL; ++t) dnq += pc[t * DM + i];
        float dnk = 0.f;
        for (int r2 = 0; r2 < d; ++r2) dnk += col[r2 * LD + i];
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
    if (threadIdx.x == 0) {
        float u = 0.f, v = 0.f;
        for (int r2 = 0; r2 < d; ++r2) {
            float ur = pa[r2], vr = pb[r2];
#pragma unroll
            for (int t = 1; t < NSL; ++t) { ur += pa[t * DM + r2]; vr += pb[t * DM + r2]; }
            u += ur;
            v += vr;
        }
        dt1_part[g] = u;
        dt2_part[g] = v;
    }
}
}  // namespace ts
}  // namespace p4c

extern "C" int p4c_epa_small_fwd(const float* G, const float* Gq, const float* Gk, const float* KP, const float* t1, const float* t2, float* At,
                                 float* Mq, float* nq, float* nk, int B, int heads, int d, int p, int diag_only, p4c_stream_t stream) {
    P4C_CHECK_ARG(G && Gq && Gk && KP && t1 && t2 && At && Mq && nq && nk, "p4c_epa_small_fwd: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && d > 0 && d <= 128 && p > 0 && p <= 64, "p4c_epa_small_fwd: head width 1..128, projection size 1..64 (got %d, %d)", d, p);
    const int ds = diag_only ? 1 : d;
    if (d <= 64) {
        constexpr int DM = 64;
        const size_t smem = (2 * DM + DM * (DM + 1) + DM * 65 + 256) * sizeof(float);
        hipLaunchKernelGGL(ts::epa_small_fwd_kernel<DM>, dim3(B * heads), dim3(256), smem, as_stream(stream), G, Gq, Gk, KP, t1, t2, At, Mq, nq, nk, heads,
                           d, p, ds);
    } else {
        constexpr int DM = 128;
        const size_t smem = (2 * DM + DM * (DM + 1) + DM * 65 + 256) * sizeof(float);
        P4C_TRY(ensure_dyn_smem((const void*)ts::epa_small_fwd_kernel<DM>, (int)smem));
        hipLaunchKernelGGL(ts::epa_small_fwd_kernel<DM>, dim3(B * heads), dim3(256), smem, as_stream(stream), G, Gq, Gk, KP, t1, t2, At, Mq, nq, nk, heads,
                           d, p, ds);
    }
    P4C_CHECK_LAUNCH("p4c_epa_small_fwd");
    return P4C_OK;
}

extern "C" int p4c_epa_small_bwd(const float* G, const float* Gq, const float* Gk, const float* KP, const float* t1, const float* t2,
                                 const float* At, const float* nq, const float* nk, const float* dAt, const float* dMq, float* dG, float* dGq,
                                 float* dGk, float* dKP, float* dt1_part, float* dt2_part, int B, int heads, int d, int p, int diag_only,
                                 p4c_stream_t stream) {
    P4C_CHECK_ARG(G && Gq && Gk && KP && t1 && t2 && At && nq && nk && dAt && dMq && dG && dGq && dGk && dKP && dt1_part && dt2_part,
                  "p4c_epa_small_bwd: null pointer");
    P4C_CHECK_ARG(B > 0 && heads > 0 && d > 0 && d <= 128 && p > 0, "p4c_epa_small_bwd: head width 1..128 (got %d)", d);
    const int ds = diag_only ? 1 : d;
    if (d <= 64) {
        constexpr int DM = 64;
        const size_t smem = (2 * DM + 2 * DM * (DM + 1) + 3 * 256) * sizeof(float);
        hipLaunchKernelGGL(ts::epa_small_bwd_kernel<DM>, dim3(B * heads), dim3(256), smem, as_stream(stream), G, Gq, Gk, KP, t1, t2, At, nq, nk, dAt, dMq,
                           dG, dGq, dGk, dKP, dt1_part, dt2_part, heads, d, p, ds);
    } else {
        constexpr int DM = 128;
        const size_t smem = (2 * DM + 2 * DM * (DM + 1) + 3 * 256) * sizeof(float);
        P4C_TRY(ensure_dyn_smem((const void*)ts::epa_small_bwd_kernel<DM>, (int)smem));
        hipLaunchKernelGGL(ts::epa_small_bwd_kernel<DM>, dim3(B * heads), dim3(256), smem, as_stream(stream), G, Gq, Gk, KP, t1, t2, At, nq, nk, dAt, dMq,
                           dG, dGq, dGk, dKP, dt1_part, dt2_part, heads, d, p, ds);
    }
    P4C_CHECK_LAUNCH("p4c_epa_small_bwd");
    return P4C_OK;
}

// ---- sums of partials (round 6) ---------------------------------------------------------------------------------------------
// The gram kernels leave one partial per token split; the caller used to sum them with the tensor library (a reduction launch, then
// one strided copy per piece of a gram_norms row, a cast and a bias addition for the token-axis projection).  reduce_splits_kernel
// does all of that in ONE pass: part is (A, S, R, E) fp32; the S partials of 4 consecutive columns of one (a, r) row are added in a
// fixed order (bit-identical reruns), bias[column % bias_len] is added and the quad stored into the segment the columns belong to --
// up to three dense outputs (G | nq2 | nk2 of a gram_norms row).
namespace p4c {
namespace ts {
struct RedSegs {
    float* out[3];
    int end[3];      // exclusive column ends of the segments (multiples of 4); segment i = [end[i-1], end[i])
    int nseg;
};

// A workgroup = 32 column quads x 8 split lanes: lane sl adds the partials sl, sl + 8, ... (loads of different splits in flight
// together), the eight sums meet in LDS and are added in lane order -- the order is fixed, a rerun reproduces every bit.
__global__ void __launch_bounds__(256) reduce_splits_kernel(const float* __restrict__ part, int S, int R, int E, RedSegs segs,
                                                            const float* __restrict__ bias, int bias_len, int accumulate, int64_t total4) {
    __shared__ p4c_f32x4 red[8][32];
    const int q = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int64_t idx = (int64_t)blockIdx.x * 32 + q;
    const bool live = idx < total4;
    const int e4 = E >> 2;
    const int64_t row = live ? idx / e4 : 0;            // (a, r)
    const int j = live ? (int)(idx - row * e4) << 2 : 0;
    const int64_t a = row / R, r = row - a * R;
    const float* p = part + ((a * S) * R + r) * (int64_t)E + j;
    const int64_t sstride = (int64_t)R * E;
    p4c_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        int s = sl;
        for (; s + 24 < S; s += 32) {                    // four loads in flight per lane
            const p4c_f32x4 v0 = *reinterpret_cast<const p4c_f32x4*>(p + (int64_t)s * sstride);
            const p4c_f32x4 v1 = *reinterpret_cast<const p4c_f32x4*>(p + (int64_t)(s + 8) * sstride);
            const p4c_f32x4 v2 = *reinterpret_cast<const p4c_f32x4*>(p + (int64_t)(s + 16) * sstride);
            const p4c_f32x4 v3 = *reinterpret_cast<const p4c_f32x4*>(p + (int64_t)(s + 24) * sstride);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (((acc[i] + v0[i]) + v1[i]) + v2[i]) + v3[i];
        }
        for (; s < S; s += 8) {
            const p4c_f32x4 v = *reinterpret_cast<const p4c_f32x4*>(p + (int64_t)s * sstride);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += v[i];
        }
    }
    red[sl][q] = acc;
    __syncthreads();
    if (sl != 0 || !live) return;
#pragma unroll
    for (int t = 1; t < 8; ++t) {
        const p4c_f32x4 v = red[t][q];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] += v[i];
    }
    if (bias) {
        const int bj = j % bias_len;                     // bias_len % 4 == 0: the four columns stay inside one period
        acc[0] += bias[bj]; acc[1] += bias[bj + 1]; acc[2] += bias[bj + 2]; acc[3] += bias[bj + 3];
    }
    int seg = 0, begin = 0;
    while (seg + 1 < segs.nseg && j >= segs.end[seg]) begin = segs.end[seg++];
    const int len = segs.end[seg] - begin;
    float* o = segs.out[seg] + row * len + (j - begin);
    if (accumulate) {
        const p4c_f32x4 old = *reinterpret_cast<const p4c_f32x4*>(o);
        acc[0] += old[0]; acc[1] += old[1]; acc[2] += old[2]; acc[3] += old[3];
    }
    *reinterpret_cast<p4c_f32x4*>(o) = acc;
}

// out (E x R) (+)= sum over the S groups of part[s] (R x E), E <= 64 (a multiple of 4): the weight gradient of the token-axis
// projection, whose per-group products come token-major ((N x p) from an apply) while the parameter is (p x N).  A workgroup
// transposes 32 rows through LDS: 16-byte reads of whole rows (all groups' loads of a quad in flight together), 128-byte runs of one
// output row on the way out.
__global__ void __launch_bounds__(256) reduce_transpose_kernel(const float* __restrict__ part, int S, int64_t R, int E, float* __restrict__ out,
                                                               int accumulate) {
    __shared__ float tile[32][65];
    const int64_t r0 = (int64_t)blockIdx.x * 32;
    const int e4 = E >> 2;
    for (int i = threadIdx.x; i < 32 * e4; i += 256) {
        const int rr = i / e4, j = (i - rr * e4) << 2;
        p4c_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + rr < R) {
            const float* p = part + (r0 + rr) * E + j;
            for (int s = 0; s < S; ++s) {
                const p4c_f32x4 u = *reinterpret_cast<const p4c_f32x4*>(p + (int64_t)s * R * E);
                v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
            }
        }
        tile[rr][j] = v[0]; tile[rr][j + 1] = v[1]; tile[rr][j + 2] = v[2]; tile[rr][j + 3] = v[3];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * E; i += 256) {
        const int j = i >> 5, rr = i & 31;
        if (r0 + rr < R) {
            float* o = out + (int64_t)j * R + r0 + rr;
            *o = (accumulate ? *o : 0.f) + tile[rr][j];
        }
    }
}
}  // namespace ts
}  // namespace p4c

namespace p4c {
namespace ts {
// Column sums of up to four small dense fp32 matrices in one launch (job = blockIdx.x): out[j] = sum over the rows of in[row][j],
// cols <= 64.  The tails of an EPA backward: the bias gradient of the token-axis Linear (2 B C rows of p columns) and the two
// temperature gradients (B rows of `heads` columns each) -- three tensor-library reductions before.  64 column lanes x 16 row lanes,
// the row lanes' sums added in lane order.
struct ColSumJobs {
    const float* in[4];
    float* out[4];
    int64_t rows[4];
    int cols[4];
};

__global__ void __launch_bounds__(1024) colsums_kernel(ColSumJobs jobs) {
    __shared__ float red[16][64];
    const int job = blockIdx.x, c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int cols = jobs.cols[job];
    const int64_t rows = jobs.rows[job];
    const float* in = jobs.in[job];
    float acc = 0.f;
    if (c < cols) {
        int64_t r = rl;
        for (; r + 48 < rows; r += 64) {
            const float v0 = in[r * cols + c], v1 = in[(r + 16) * cols + c], v2 = in[(r + 32) * cols + c], v3 = in[(r + 48) * cols + c];
            acc = (((acc + v0) + v1) + v2) + v3;
        }
        for (; r < rows; r += 16) acc += in[r * cols + c];
    }
    red[rl][c] = acc;
    __syncthreads();
    if (rl == 0 && c < cols) {
#pragma unroll
        for (int t = 1; t < 16; ++t) acc += red[t][c];
        jobs.out[job][c] = acc;
    }
}
}  // namespace ts
}  // namespace p4c

namespace p4c {
namespace ts {
// The published EPA code merges the spatial branch as `x_SA.permute(0, 3, 1, 2).reshape(B, N, C)` on a (B, heads, N, d) tensor: the
// memory order becomes (d, heads, N) -- per sample the TRANSPOSE of the token-major (N x C) matrix the kernels produce, with the
// channels reordered c = hh d + j -> row r = j heads + hh.  The tensor library does that as a strided gather (1 TB/s); here a
// workgroup moves a 64-token x 64-channel tile through LDS: 16-byte reads of token rows, 16-byte writes of 8 consecutive tokens of
// one output row.  inverse = 1: the adjoint (the gradient back to token-major).
constexpr int MP_LD = 72;       // LDS row stride in bf16 elements (144 bytes: 16-byte aligned rows)

__global__ void __launch_bounds__(256) merge_published_kernel(const unsigned short* __restrict__ in, unsigned short* __restrict__ out, int64_t N,
                                                              int heads, int d, int inverse) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * MP_LD];
    const int C = heads * d;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64, b = blockIdx.z;
    const int64_t base = (int64_t)b * N * C;
    if (!inverse) {
        // token-major in: tile[t][c]
        for (int v = threadIdx.x; v < 64 * 8; v += 256) {
            const int t = v >> 3, c8 = (v & 7) << 3;
            ts_u32x4 q = {0u, 0u, 0u, 0u};
            if (n0 + t < N && c0 + c8 < C) q = *reinterpret_cast<const ts_u32x4*>(in + base + (n0 + t) * C + c0 + c8);
            *reinterpret_cast<ts_u32x4*>(tile + t * MP_LD + c8) = q;
        }
        __syncthreads();
        for (int v = threadIdx.x; v < 64 * 8; v += 256) {
            const int cl = v >> 3, t8 = (v & 7) << 3, c = c0 + cl;
            if (c >= C || n0 + t8 >= N) continue;
            const int hh = c / d, j = c - hh * d;
            union { unsigned short s[8]; ts_u32x4 q; } u;
#pragma unroll
            for (int i = 0; i < 8; ++i) u.s[i] = tile[(t8 + i) * MP_LD + cl];
            *reinterpret_cast<ts_u32x4*>(out + base + ((int64_t)j * heads + hh) * N + n0 + t8) = u.q;
        }
    } else {
        // transposed in: tile[c][t]
        for (int v = threadIdx.x; v < 64 * 8; v += 256) {
            const int cl = v >> 3, t8 = (v & 7) << 3, c = c0 + cl;
            ts_u32x4 q = {0u, 0u, 0u, 0u};
            if (c < C && n0 + t8 < N) {
                const int hh = c / d, j = c - hh * d;
                q = *reinterpret_cast<const ts_u32x4*>(in + base + ((int64_t)j * heads + hh) * N + n0 + t8);
            }
            *reinterpret_cast<ts_u32x4*>(tile + cl * MP_LD + t8) = q;
        }
        __syncthreads();
        for (int v = threadIdx.x; v < 64 * 8; v += 256) {
            const int t = v >> 3, c8 = (v & 7) << 3;
            if (n0 + t >= N || c0 + c8 >= C) continue;
            union { unsigned short s[8]; ts_u32x4 q; } u;
#pragma unroll
            for (int i = 0; i < 8; ++i) u.s[i] = tile[(c8 + i) * MP_LD + t];
            *reinterpret_cast<ts_u32x4*>(out + base + (n0 + t) * C + c0 + c8) = u.q;
        }
    }
}
}  // namespace ts
}  // namespace p4c

extern "C" int p4c_ts_merge_published(const void* x, void* out, int B, int64_t N, int heads, int d, int inverse, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && out && x != out, "p4c_ts_merge_published: null pointer or in-place call");
    P4C_CHECK_ARG(B > 0 && N > 0 && heads > 0 && d > 0, "p4c_ts_merge_published: empty problem");
    const int C = heads * d;
    P4C_CHECK_ARG(N % 8 == 0 && C % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0,
                  "p4c_ts_merge_published: tokens and channels must be multiples of 8 (got %lld, %d), buffers 16-byte aligned", (long long)N, C);
    hipLaunchKernelGGL(ts::merge_published_kernel, dim3((unsigned)((N + 63) / 64), (C + 63) / 64, B), dim3(256), 0, as_stream(stream),
                       static_cast<const unsigned short*>(x), static_cast<unsigned short*>(out), N, heads, d, inverse);
    P4C_CHECK_LAUNCH("p4c_ts_merge_published");
    return P4C_OK;
}

extern "C" int p4c_ts_colsums(int njobs, const float* const* in, const int64_t* rows, const int* cols, float* const* out, p4c_stream_t stream) {
    P4C_CHECK_ARG(njobs >= 1 && njobs <= 4 && in && rows && cols && out, "p4c_ts_colsums: 1..4 jobs");
    ts::ColSumJobs jobs{};
    for (int i = 0; i < njobs; ++i) {
        P4C_CHECK_ARG(in[i] && out[i] && rows[i] > 0 && cols[i] > 0 && cols[i] <= 64, "p4c_ts_colsums: job %d: null pointer, no rows or more than 64 columns", i);
        jobs.in[i] = in[i];
        jobs.out[i] = out[i];
        jobs.rows[i] = rows[i];
        jobs.cols[i] = cols[i];
    }
    hipLaunchKernelGGL(ts::colsums_kernel, dim3(njobs), dim3(1024), 0, as_stream(stream), jobs);
    P4C_CHECK_LAUNCH("p4c_ts_colsums");
    return P4C_OK;
}

extern "C" int p4c_ts_reduce_splits(const float* part, int A, int S, int R, int E, int nseg, const int* seg_len, float* const* outs,
                                    const float* bias, int bias_len, int accumulate, p4c_stream_t stream) {
    P4C_CHECK_ARG(part && seg_len && outs, "p4c_ts_reduce_splits: null pointer");
    P4C_CHECK_ARG(A > 0 && S > 0 && R > 0 && E > 0 && E % 4 == 0, "p4c_ts_reduce_splits: empty problem or a row length that is not a multiple of 4 (%d)", E);
    P4C_CHECK_ARG(nseg >= 1 && nseg <= 3, "p4c_ts_reduce_splits: 1..3 output segments (got %d)", nseg);
    P4C_CHECK_ARG(!bias || (bias_len > 0 && bias_len % 4 == 0 && E % bias_len == 0), "p4c_ts_reduce_splits: bias period %d must divide the row length %d (multiples of 4)",
                  bias_len, E);
    ts::RedSegs segs{};
    segs.nseg = nseg;
    int end = 0;
    for (int i = 0; i < nseg; ++i) {
        P4C_CHECK_ARG(seg_len[i] > 0 && seg_len[i] % 4 == 0 && outs[i] && (reinterpret_cast<uintptr_t>(outs[i]) & 15) == 0,
                      "p4c_ts_reduce_splits: segment %d: length must be a positive multiple of 4, output 16-byte aligned", i);
        end += seg_len[i];
        segs.end[i] = end;
        segs.out[i] = outs[i];
    }
    P4C_CHECK_ARG(end == E, "p4c_ts_reduce_splits: the segments cover %d of %d columns", end, E);
    P4C_CHECK_ARG((reinterpret_cast<uintptr_t>(part) & 15) == 0, "p4c_ts_reduce_splits: partials must be 16-byte aligned");
    const int64_t total4 = (int64_t)A * R * (E / 4);
    hipLaunchKernelGGL(ts::reduce_splits_kernel, dim3((unsigned)((total4 + 31) / 32)), dim3(256), 0, as_stream(stream), part, S, R, E, segs, bias,
                       bias_len, accumulate, total4);
    P4C_CHECK_LAUNCH("p4c_ts_reduce_splits");
    return P4C_OK;
}

extern "C" int p4c_ts_reduce_transpose(const float* part, int S, int64_t R, int E, float* out, int accumulate, p4c_stream_t stream) {
    P4C_CHECK_ARG(part && out, "p4c_ts_reduce_transpose: null pointer");
    P4C_CHECK_ARG(S > 0 && R > 0 && E > 0 && E <= 64 && E % 4 == 0, "p4c_ts_reduce_transpose: 4..64 columns per row, a multiple of 4 (got %d)", E);
    P4C_CHECK_ARG((reinterpret_cast<uintptr_t>(part) & 15) == 0, "p4c_ts_reduce_transpose: partials must be 16-byte aligned");
    hipLaunchKernelGGL(ts::reduce_transpose_kernel, dim3((unsigned)((R + 31) / 32)), dim3(256), 0, as_stream(stream), part, S, R, E, out, accumulate);
    P4C_CHECK_LAUNCH("p4c_ts_reduce_transpose");
    return P4C_OK;
}
