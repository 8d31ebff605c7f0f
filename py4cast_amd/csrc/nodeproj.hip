// Mesh-GNN launch grouping (GraphLAM / HiLAM / HiLAMParallel, config/CLI/model/graphlam.yaml:19-26, hilam.yaml, hilamparallel.yaml; the
// classes come from mfai through py4cast/models.py:66-89).  A training step of HiLAM is ~190 InteractionNets per optimizer step, most of
// them on mesh levels of 18 ... 13 122 rows: a launch there costs its fixed latency whatever it computes, so what counts is how many
// dependent launches an InteractionNet needs.  Two things live here:
//
// 1. The node projections of an InteractionNet as ONE launch per direction.  The first Linear of the edge MLP is distributed over
//    cat[e, x_s[src], x_r[dst]] and the node-update MLP's over cat[x_r, agg] (py4cast_amd/graphlam.py), so a node tensor x (R, 64) is
//    multiplied with up to three 64 x 64 blocks of wider fp32 weights:
//      node_proj_fwd   : y_i = x W_i^T                 (x read once; i < n <= 3)
//      node_proj_dgrad : dx  = sum_i dy_i W_i (+ acc)  (one K = 64 n product instead of n accumulating launches)
//      node_proj_wgrad : dW_i = dy_i^T x               (x read once; per-wave fp32 partials, reduced by the queue below)
//    Before: n row-GEMM launches forward, n backward, and per block a weight-gradient launch + reduction + `+=` (library GEMM + add on
//    the small levels): ~14 launches per InteractionNet, now 3 + a share of the batched reduction.
//
// 2. The deferred reduction of parameter-gradient partials.  p4c_row_mlp_bwd_accumulate and node_proj_wgrad leave per-workgroup /
//    per-wave partials; instead of one dependent ~5 us reduction launch behind each of them (~550 per HiLAM step) the jobs are queued
//    and reduced GRAD_BATCH at a time by grad_reduce_batch_kernel (block -> (job, block of the job) through a prefix table in the kernel
//    arguments; the pattern of wgrad_reduce_batch, conv_f32.hip).  Every job is summed exactly as its own launch would sum it and jobs
//    are applied in submission order (a job whose destination already appears in the batch being assembled starts a new launch), so
//    the accumulated gradients are bit-identical to the undeferred ones.
#include <mutex>
#include <vector>

#include "kernels.hpp"

namespace p4c {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int C = 64;                 // features of a node / edge representation
constexpr int PROW = C * 2 + 16;      // LDS row stride of a [row][64 features] bf16 image (padded: transposed reads spread over banks)

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
__device__ __forceinline__ int rowmap(int i) { return (i & 3) + 8 * (i >> 2); }

struct NodeProjArgs {
    const bf16* x;          // (R, 64) node rows (fwd, wgrad)
    int64_t R;
    int n;                  // projections, 1..3
    const float* w[3];      // 64 x 64 blocks of fp32 master weights, W_i[o][k] = w[i][o * ldw[i] + k]
    int ldw[3];
    bf16* y[3];             // (R, 64) each: outputs (fwd) / upstream gradients (dgrad, wgrad)
    bf16* dx;               // dgrad output (R, 64)
    const bf16* acc;        // dgrad: added to the product (may be dx itself), or NULL
    float* partial;         // wgrad: [active waves][n][64][64]
};

// LDS operand image: img[(tile * S + s) * 64 + lane] = 8 bf16 = M[32 tile + (lane & 31)][16 s + 8 (lane >> 5) + j]
// forward:       M = [W_0; W_1; W_2]          (64 n output features x 64 inputs):   tiles = 2 n, S = 4
// data gradient: M = [W_0^T | W_1^T | W_2^T]  (64 input features x 64 n outputs):   tiles = 2,   S = 4 n
template <int NP, bool DGRAD>
__device__ __forceinline__ void build_images(__bf16* img, const NodeProjArgs& a) {
    constexpr int TILES = DGRAD ? 2 : 2 * NP, S = DGRAD ? 4 * NP : 4;
    constexpr int pieces = TILES * S * 64;
    static_assert(pieces % 256 == 0, "whole trips of the 256-thread workgroup");
    // (a compile-time trip count: unrolled, every trip's weight loads are in flight together -- one memory round trip per launch)
#pragma unroll
    for (int it = 0; it < pieces / 256; ++it) {
        const int idx = threadIdx.x + 256 * it;
        const int ln = idx & 63, ts = idx >> 6;
        const int s = ts % S, tile = ts / S;
        const int m = 32 * tile + (ln & 31), k0 = 16 * s + 8 * (ln >> 5);
        bf16x8 o;
        if (!DGRAD) {
            const int i = m >> 6, oo = m & 63;
            const float* p = a.w[i] + (int64_t)oo * a.ldw[i] + k0;
            if (((a.ldw[i] & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.w[i]) & 15) == 0)) {
                const float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
                o[0] = (__bf16)lo.x; o[1] = (__bf16)lo.y; o[2] = (__bf16)lo.z; o[3] = (__bf16)lo.w;
                o[4] = (__bf16)hi.x; o[5] = (__bf16)hi.y; o[6] = (__bf16)hi.z; o[7] = (__bf16)hi.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (__bf16)p[j];
            }
        } else {
            const int i = k0 >> 6, n0 = k0 & 63;     // the piece's 8 reduction indices are outputs n0 .. n0 + 7 of W_i; m = input feature
            const float* p = a.w[i] + (int64_t)n0 * a.ldw[i] + m;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)p[(int64_t)j * a.ldw[i]];
        }
        *reinterpret_cast<bf16x8*>(img + (int64_t)idx * 8) = o;
    }
}

template <int NP>
__global__ void __launch_bounds__(256, 2) node_proj_fwd_kernel(NodeProjArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 img[2 * NP * 4 * 64 * 8];
    build_images<NP, false>(img, a);
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (a.R + 31) / 32;
    auto load = [&](bf16x8 (&xs)[4], int64_t t) __attribute__((always_inline)) {
        const int64_t row = t * 32 + r;
        const bool live = t < ntiles && row < a.R;
        const bf16* p = a.x + (live ? row : 0) * C;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + 16 * s + 8 * h);
            xs[s] = live ? v : zero8();
        }
    };
    bf16x8 cur[4], nxt[4];
    load(cur, wave);
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        load(nxt, t + nwaves);
        const int64_t row = t * 32 + r;
        const bool live = row < a.R;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(img + (((2 * i + m) * 4 + s) * 64 + lane) * 8),
                                                                     cur[s], acc[m], 0, 0, 0);
            bf16* yrow = a.y[i] + (live ? row : 0) * C;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[m][4 * q + e];
                    if (live) *reinterpret_cast<bf16x4*>(yrow + 32 * m + 8 * q + 4 * h) = o;
                }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) cur[s] = nxt[s];
    }
}

template <int NP>
__global__ void __launch_bounds__(256, 2) node_proj_dgrad_kernel(NodeProjArgs a) {
    __shared__ __attribute__((aligned(16))) __bf16 img[2 * NP * 4 * 64 * 8];
    build_images<NP, true>(img, a);
    __syncthreads();
    constexpr int S = 4 * NP;
    const int lane = threadIdx.x & 63, h = lane >> 5, r = lane & 31;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (a.R + 31) / 32;
    auto load = [&](bf16x8 (&ds)[S], int64_t t) __attribute__((always_inline)) {
        const int64_t row = t * 32 + r;
        const bool live = t < ntiles && row < a.R;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const bf16* p = a.y[i] + (live ? row : 0) * C;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(p + 16 * s + 8 * h);
                ds[4 * i + s] = live ? v : zero8();
            }
        }
    };
    bf16x8 cur[S], nxt[S];
    load(cur, wave);
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        load(nxt, t + nwaves);
        const int64_t row = t * 32 + r;
        const bool live = row < a.R;
        f32x16 acc[2] = {zero16(), zero16()};
#pragma unroll
        for (int s = 0; s < S; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(img + ((m * S + s) * 64 + lane) * 8), cur[s], acc[m],
                                                                 0, 0, 0);
        const int64_t rc = live ? row : 0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 32 * m + 8 * q + 4 * h;
                float v[4] = {acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]};
                if (a.acc) {
                    const bf16x4 rq = *reinterpret_cast<const bf16x4*>(a.acc + rc * C + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += (float)rq[e];
                }
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
                if (live) *reinterpret_cast<bf16x4*>(a.dx + rc * C + k) = o;
            }
#pragma unroll
        for (int s = 0; s < S; ++s) cur[s] = nxt[s];
    }
}

// natural-order transposed operand of k-step ks (rows 16 ks .. +15) from a [row][feature] image: lane (feature 32 tile + (lane & 31), h)
// receives rows 16 ks + 8 h + j, j = 0..7   (mlp.hip's read_tr_nat)
__device__ __forceinline__ bf16x8 read_tr(const char* img, int ks, int tile, int lane) {
    const int i = lane & 15, tg = (lane >> 4) & 1, h = lane >> 5;
    const char* p = img + (16 * ks + 8 * h + (i >> 2)) * PROW + (32 * tile + tg * 16 + (i & 3) * 4) * 2;
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * PROW));
    return u.v;
}

// dW_i[o][k] = sum_r dy_i[r][o] x[r][k]: the reduction index is the row, both operands are transposed LDS reads of [row][feature]
// images of a 32-row tile (per wave); the next tile's rows are in flight while the current one is multiplied; every wave that had a
// tile leaves its 64 x 64 n accumulators in its own partial slot (waves 0 .. min(tiles, 4 G) - 1: contiguous slots).
template <int NP>
__global__ void __launch_bounds__(256, 1) node_proj_wgrad_kernel(NodeProjArgs a) {
    constexpr int WAVE_LDS = (NP + 1) * 32 * PROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* imgX = smem + wv * WAVE_LDS;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wv, nwaves = (int64_t)gridDim.x * 4;
    const int64_t ntiles = (a.R + 31) / 32;
    if (wave >= ntiles) return;
    const int rsub = lane >> 3, chunk = lane & 7;
    struct Tile { u32x4 v[NP + 1][4]; };
    auto load = [&](Tile& tl, int64_t t) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int64_t row = t * 32 + 8 * it + rsub;
            const bool live = t < ntiles && row < a.R;
            const int64_t rc = live ? row : 0;
            const u32x4 vx = *reinterpret_cast<const u32x4*>(a.x + rc * C + chunk * 8);
            tl.v[0][it] = live ? vx : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const u32x4 vd = *reinterpret_cast<const u32x4*>(a.y[i] + rc * C + chunk * 8);
                tl.v[i + 1][it] = live ? vd : u32x4{0u, 0u, 0u, 0u};
            }
        }
    };
    f32x16 acc[NP][2][2];
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m) acc[i][m][0] = acc[i][m][1] = zero16();
    Tile cur, nxt;
    load(cur, wave);
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        load(nxt, t + nwaves);
#pragma unroll
        for (int g = 0; g < NP + 1; ++g)
#pragma unroll
            for (int it = 0; it < 4; ++it) *reinterpret_cast<u32x4*>(imgX + g * 32 * PROW + (8 * it + rsub) * PROW + chunk * 16) = cur.v[g][it];
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 bx[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) bx[n] = read_tr(imgX, ks, n, lane);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                bf16x8 ad[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) ad[m] = read_tr(imgX + (i + 1) * 32 * PROW, ks, m, lane);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[i][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad[m], bx[n], acc[i][m][n], 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
        cur = nxt;
    }
    const int h = lane >> 5, r = lane & 31;
    float* dst = a.partial + wave * (int64_t)(NP * C * C);
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[i * C * C + (32 * m + rowmap(e) + 4 * h) * C + 32 * n + r] = acc[i][m][n][e];
}

int proj_grid(int64_t R, int per_cu) {
    const int64_t tiles = (R + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * per_cu;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}
// weight gradient: one tile per wave on the small mesh levels (latency), four from 64 tiles on (4 x fewer 48 KB partial slots)
int wgrad_grid(int64_t R) {
    const int64_t tiles = (R + 31) / 32;
    const int64_t per_wave = tiles <= 64 ? 1 : 4;
    int64_t blocks = (tiles + 4 * per_wave - 1) / (4 * per_wave);
    if (blocks > num_cus()) blocks = num_cus();
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}
int wgrad_slots(int64_t R) {
    const int64_t tiles = (R + 31) / 32, waves = (int64_t)wgrad_grid(R) * 4;
    return (int)(tiles < waves ? tiles : waves);
}

// ------------------------------------------------------------------------------------------------ deferred gradient reductions
constexpr int GRAD_BATCH = 32;
struct GradBatchArgs {
    GradReduceJob job[GRAD_BATCH];
    int first[GRAD_BATCH + 1];
    int n;
};
static_assert(sizeof(GradBatchArgs) <= 4000, "the job table travels in the kernel arguments");

// a block owns 32 outputs of one job: its 8 thread rows take the slots s = sg (mod 8) with two independent partial sums each, then
// the 8 rows are added in order through LDS (the order of mlp.hip's mlp_param_reduce_acc_kernel, which this replaces)
__global__ void __launch_bounds__(256) grad_reduce_batch_kernel(GradBatchArgs a) {
    __shared__ float red[8][33];
    int jb = 0;
    while (jb + 1 < a.n && (int)blockIdx.x >= a.first[jb + 1]) ++jb;
    const GradReduceJob& J = a.job[jb];
    const float* __restrict__ partial = J.partial;
    const int n = J.n, slots = J.slots;
    const int jj = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const int j = ((int)blockIdx.x - a.first[jb]) * 32 + jj;
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
        if (J.kind == GRAD_JOB_MLP) {
            // [dW1 64 x K | dW2 64 x 64 | db1 | db2 | dgamma | dbeta]; p = {dw1, dw2, db1, db2, dgamma, dbeta}, ld[0] = row stride of dw1
            const int K = J.K, w2_0 = C * K, b1_0 = w2_0 + C * C;
            if (j < w2_0) {
                const int o = j / K, k = j - o * K;
                if (J.p[0] && k < J.k_real) J.p[0][(int64_t)o * J.ld[0] + k] += t;
            } else if (j < b1_0) {
                const int o = (j - w2_0) >> 6;
                if (J.p[1] && o < J.o_real) J.p[1][j - w2_0] += t;
            } else {
                const int which = (j - b1_0) >> 6, c = (j - b1_0) & 63;
                if (which == 0 && J.p[2]) J.p[2][c] += t;
                if (which == 1 && J.p[3] && c < J.o_real) J.p[3][c] += t;
                if (which == 2 && J.p[4]) J.p[4][c] += t;
                if (which == 3 && J.p[5]) J.p[5][c] += t;
            }
        } else {
            // [i][o][k] 64 x 64 blocks; p[i] with row stride ld[i]
            const int i = j >> 12, o = (j >> 6) & 63, k = j & 63;
            if (J.p[i]) J.p[i][(int64_t)o * J.ld[i] + k] += t;
        }
    }
}

struct Pending {
    GradReduceJob job;
    hipStream_t stream;
};
std::mutex g_mu;
std::vector<Pending> g_pending;
bool g_defer = false;

int launch_batch(const GradBatchArgs& b, hipStream_t st) {
    hipLaunchKernelGGL(grad_reduce_batch_kernel, dim3(b.first[b.n]), dim3(256), 0, st, b);
    P4C_CHECK_LAUNCH("grad_reduce_batch");
    return P4C_OK;
}

// jobs[lo, hi) in order, GRAD_BATCH per launch; a job that adds into a buffer some job of the launch being assembled already adds into
// starts a new launch (launches of one stream run in order: the additions into one element happen in submission order)
int flush_range(const std::vector<Pending>& v, hipStream_t st) {
    GradBatchArgs b;
    b.n = 0;
    b.first[0] = 0;
    for (const Pending& pj : v) {
        bool clash = false;
        for (int q = 0; q < b.n && !clash; ++q)
            for (int u = 0; u < 6 && !clash; ++u)
                for (int w = 0; w < 6; ++w)
                    if (pj.job.p[u] && pj.job.p[u] == b.job[q].p[w]) { clash = true; break; }
        if (b.n == GRAD_BATCH || clash) {
            P4C_TRY(launch_batch(b, st));
            b.n = 0;
        }
        b.job[b.n] = pj.job;
        b.first[b.n + 1] = b.first[b.n] + (pj.job.n + 31) / 32;
        ++b.n;
    }
    if (b.n) P4C_TRY(launch_batch(b, st));
    return P4C_OK;
}

}  // namespace

bool grad_reduce_deferring() {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_defer;
}

int grad_reduce_submit(const GradReduceJob& job, hipStream_t st) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_defer) {
            g_pending.push_back(Pending{job, st});
            return P4C_OK;
        }
    }
    std::vector<Pending> one{Pending{job, st}};
    return flush_range(one, st);
}

}  // namespace p4c

using namespace p4c;

extern "C" int p4c_grad_reduce_defer(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    const int was = g_defer ? 1 : 0;
    if (on < 0) {                       // -1: drop what is queued (a backward pass that died before its flush) and reduce at once again
        g_pending.clear();
        tn_reduce_drop();
    }
    g_defer = on > 0;
    return was;
}

extern "C" int p4c_grad_reduce_pending(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_pending.size() + tn_reduce_pending();
}

extern "C" int p4c_grad_reduce_flush(p4c_stream_t stream) {
    std::vector<Pending> jobs;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        jobs.swap(g_pending);
    }
    hipStream_t st = as_stream(stream);
    P4C_TRY(tn_reduce_flush(st));
    if (jobs.empty()) return P4C_OK;
    for (const Pending& pj : jobs)
        P4C_CHECK_ARG(pj.stream == st, "p4c_grad_reduce_flush: a queued job was produced on another stream (flush on the stream of the backward)");
    return flush_range(jobs, st);
}

static int check_proj(const char* name, int64_t R, int n, const float* const* w, const int32_t* ldw, void* const* y) {
    P4C_CHECK_ARG(R > 0 && R < ((int64_t)1 << 31), "%s: bad row count %lld", name, (long long)R);
    P4C_CHECK_ARG(n >= 1 && n <= 3, "%s: 1..3 projections, got %d", name, n);
    for (int i = 0; i < n; ++i) {
        P4C_CHECK_ARG(w && w[i] && y && y[i], "%s: NULL pointer (projection %d)", name, i);
        P4C_CHECK_ARG(ldw[i] >= C, "%s: weight row stride %d < 64", name, ldw[i]);
        P4C_CHECK_ARG((reinterpret_cast<uintptr_t>(y[i]) & 15) == 0, "%s: rows must be 16-byte aligned", name);
    }
    return P4C_OK;
}

static NodeProjArgs proj_args(const void* x, int64_t R, int n, const float* const* w, const int32_t* ldw, void* const* y) {
    NodeProjArgs a{};
    a.x = (const bf16*)x;
    a.R = R;
    a.n = n;
    for (int i = 0; i < n; ++i) {
        a.w[i] = w[i];
        a.ldw[i] = ldw[i];
        a.y[i] = (bf16*)y[i];
    }
    return a;
}

extern "C" int p4c_node_proj_fwd(const void* x, int64_t R, int n, const float* const* w, const int32_t* ldw, void* const* y, p4c_stream_t stream) {
    P4C_TRY(check_proj("p4c_node_proj_fwd", R, n, w, ldw, y));
    P4C_CHECK_ARG(x && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "p4c_node_proj_fwd: x must be a 16-byte aligned pointer");
    NodeProjArgs a = proj_args(x, R, n, w, ldw, y);
    hipStream_t st = as_stream(stream);
    const int G = proj_grid(R, 2);
    switch (n) {
        case 1: hipLaunchKernelGGL(node_proj_fwd_kernel<1>, dim3(G), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(node_proj_fwd_kernel<2>, dim3(G), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(node_proj_fwd_kernel<3>, dim3(G), dim3(256), 0, st, a); break;
    }
    P4C_CHECK_LAUNCH("node_proj_fwd");
    return P4C_OK;
}

extern "C" int p4c_node_proj_dgrad(const void* const* dy, int64_t R, int n, const float* const* w, const int32_t* ldw, void* dx, const void* acc,
                                   p4c_stream_t stream) {
    P4C_TRY(check_proj("p4c_node_proj_dgrad", R, n, w, ldw, const_cast<void* const*>(dy)));
    P4C_CHECK_ARG(dx && (reinterpret_cast<uintptr_t>(dx) & 15) == 0 && (reinterpret_cast<uintptr_t>(acc) & 7) == 0,
                  "p4c_node_proj_dgrad: dx / acc must be aligned row pointers");
    NodeProjArgs a = proj_args(nullptr, R, n, w, ldw, const_cast<void* const*>(dy));
    a.dx = (bf16*)dx;
    a.acc = (const bf16*)acc;
    hipStream_t st = as_stream(stream);
    const int G = proj_grid(R, 2);
    switch (n) {
        case 1: hipLaunchKernelGGL(node_proj_dgrad_kernel<1>, dim3(G), dim3(256), 0, st, a); break;
        case 2: hipLaunchKernelGGL(node_proj_dgrad_kernel<2>, dim3(G), dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL(node_proj_dgrad_kernel<3>, dim3(G), dim3(256), 0, st, a); break;
    }
    P4C_CHECK_LAUNCH("node_proj_dgrad");
    return P4C_OK;
}

extern "C" size_t p4c_node_proj_wgrad_workspace_bytes(int64_t R, int n) {
    if (R <= 0 || n < 1 || n > 3) return 0;
    return (size_t)wgrad_slots(R) * n * C * C * sizeof(float);
}

extern "C" int p4c_node_proj_wgrad(const void* const* dy, const void* x, int64_t R, int n, float* const* dw, const int32_t* ld_dw, void* workspace,
                                   p4c_stream_t stream) {
    static const int32_t ld_any[3] = {C, C, C};
    static const float dummy = 0.f;
    const float* wfake[3] = {&dummy, &dummy, &dummy};
    P4C_TRY(check_proj("p4c_node_proj_wgrad", R, n, wfake, ld_any, const_cast<void* const*>(dy)));
    P4C_CHECK_ARG(x && workspace && dw && ld_dw && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "p4c_node_proj_wgrad: NULL / unaligned pointer");
    for (int i = 0; i < n; ++i) P4C_CHECK_ARG(dw[i] == nullptr || ld_dw[i] >= C, "p4c_node_proj_wgrad: gradient row stride %d < 64", ld_dw[i]);
    NodeProjArgs a = proj_args(x, R, n, wfake, ld_any, const_cast<void* const*>(dy));
    a.partial = (float*)workspace;
    hipStream_t st = as_stream(stream);
    const int G = wgrad_grid(R);
    const int smem = 4 * (n + 1) * 32 * PROW;
    switch (n) {
        case 1:
            P4C_TRY(ensure_dyn_smem((const void*)node_proj_wgrad_kernel<1>, smem));
            hipLaunchKernelGGL(node_proj_wgrad_kernel<1>, dim3(G), dim3(256), smem, st, a);
            break;
        case 2:
            P4C_TRY(ensure_dyn_smem((const void*)node_proj_wgrad_kernel<2>, smem));
            hipLaunchKernelGGL(node_proj_wgrad_kernel<2>, dim3(G), dim3(256), smem, st, a);
            break;
        default:
            P4C_TRY(ensure_dyn_smem((const void*)node_proj_wgrad_kernel<3>, smem));
            hipLaunchKernelGGL(node_proj_wgrad_kernel<3>, dim3(G), dim3(256), smem, st, a);
            break;
    }
    P4C_CHECK_LAUNCH("node_proj_wgrad");
    GradReduceJob job{};
    job.partial = a.partial;
    job.slots = wgrad_slots(R);
    job.n = n * C * C;
    job.kind = GRAD_JOB_PROJ;
    for (int i = 0; i < n; ++i) {
        job.p[i] = dw[i];
        job.ld[i] = ld_dw[i];
    }
    return grad_reduce_submit(job, st);
}
