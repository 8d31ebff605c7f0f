// Edge gather / scatter kernels of the mesh-GNN models (GraphLAM / HiLAM / HiLAMParallel: config/CLI/model/graphlam.yaml,
// hilam.yaml, hilamparallel.yaml; the networks themselves come from mfai, py4cast/models.py:10-20).  One InteractionNet layer
// is   m_e = MLP_e([e, x_s[src(e)], x_r[dst(e)]]);  agg_n = sum_{e: dst(e)=n} m_e;  x_r += MLP_n([x_r, agg]).
// What is HBM-bound in it -- and what torch runs as index_select + cat + index_add_ (atomics) -- is here:
//   * edge_gather_add:  h[e] = act(base[e] + a[ia[e]] + b[ib[e]])   (the first Linear of MLP_e distributes over the concat,
//     so nodes are projected ONCE per node and the E x 3C concat never exists);
//   * segment_sum:      out[n] = sum_{j in [off[n], off[n+1])} msg[perm[j]]   (receiver-sorted CSR built once per graph:
//     no atomics, fixed summation order => bitwise reproducible), also the adjoint of every gather;
//   * edge_gather_add_bwd: dpre[e] = dh[e] * act'(base[e] + a[ia[e]] + b[ib[e]])  (pre-activation recomputed, not stored).
// Rows are C contiguous features (C * sizeof(T) a multiple of 16 B); a row is covered by `lpr` lanes holding 16 B each,
// a wave moves 64/lpr rows per instruction with fully coalesced 16-byte accesses.  All arithmetic in fp32.
#include "common.hpp"
#include "kernels.hpp"

namespace p4c {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Row16;
template <> struct Row16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
    }
    __device__ static __forceinline__ u32x4 pack(const float* f) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
        return v;
    }
};
template <> struct Row16<bf16> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void unpack(const u32x4& v, float* f) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(v[i] << 16);
            f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ unsigned int rne(float x) {   // round-to-nearest-even bf16 bits (NaN kept quiet)
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

template <int ACT> __device__ __forceinline__ float act_fwd(float v) {
    if (ACT == 1) return v > 0.f ? v : 0.f;
    if (ACT == 2) return v / (1.f + __expf(-v));
    return v;
}
template <int ACT> __device__ __forceinline__ float act_grad(float v) {
    if (ACT == 1) return v > 0.f ? 1.f : 0.f;
    if (ACT == 2) {
        const float s = 1.f / (1.f + __expf(-v));
        return s * (1.f + v * (1.f - s));
    }
    return 1.f;
}

// ---------------------------------------------------------------- gather-add (forward, and its pre-activation backward)
// BWD = 0: out[e] = act(pre[e]);  BWD = 1: out[e] = dh[e] * act'(pre[e]);  pre[e] = base[e] + a[ia[e]] + b[ib[e]]
// (base, a, b each optional).  Rows per wave-instruction: 64 >> lpr_log2; two row batches in flight per wave.
template <typename T, int ACT, int BWD>
__global__ void __launch_bounds__(256)
    edge_gather_add_kernel(const T* __restrict__ base, const T* __restrict__ a, const int32_t* __restrict__ ia,
                           const T* __restrict__ b, const int32_t* __restrict__ ib, const T* __restrict__ dh, T* __restrict__ out,
                           int64_t E, int chunks, int lpr_log2) {
    constexpr int NV = Row16<T>::N;
    const int lane = threadIdx.x & 63;
    const int lpr = 1 << lpr_log2, rpw = 64 >> lpr_log2;
    const int chunk = lane & (lpr - 1), sub = lane >> lpr_log2;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t e0 = wave * rpw * 2; e0 < E; e0 += nwaves * rpw * 2) {
        int64_t e[2];
        bool ok[2];
        int32_t ja[2], jb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            e[u] = e0 + u * rpw + sub;
            ok[u] = e[u] < E;
            ja[u] = (ok[u] && a) ? ia[e[u]] : 0;
            jb[u] = (ok[u] && b) ? ib[e[u]] : 0;
        }
        for (int c = chunk; c < chunks; c += lpr) {
            u32x4 vb[2], va[2], vc[2], vd[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                vb[u] = va[u] = vc[u] = vd[u] = u32x4{0, 0, 0, 0};
                if (ok[u]) {
                    if (base) vb[u] = reinterpret_cast<const u32x4*>(base)[e[u] * chunks + c];
                    if (a) va[u] = reinterpret_cast<const u32x4*>(a)[(int64_t)ja[u] * chunks + c];
                    if (b) vc[u] = reinterpret_cast<const u32x4*>(b)[(int64_t)jb[u] * chunks + c];
                    if (BWD) vd[u] = reinterpret_cast<const u32x4*>(dh)[e[u] * chunks + c];
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!ok[u]) continue;
                float fb[NV], fa[NV], fc[NV], fd[NV], fo[NV];
                Row16<T>::unpack(vb[u], fb);
                Row16<T>::unpack(va[u], fa);
                Row16<T>::unpack(vc[u], fc);
                if (BWD) Row16<T>::unpack(vd[u], fd);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const float pre = (fb[i] + fa[i]) + fc[i];
                    fo[i] = BWD ? fd[i] * act_grad<ACT>(pre) : act_fwd<ACT>(pre);
                }
                reinterpret_cast<u32x4*>(out)[e[u] * chunks + c] = Row16<T>::pack(fo);
            }
        }
    }
}

// ---------------------------------------------------------------- CSR segment sum
// out[n] = (init ? init[n] : 0) + sum_{j in [off[n], off[n+1])} msg[perm ? perm[j] : j].
// `split` = 2^split_log2 lane groups (of lpr lanes) share one segment: group g takes list positions g, g+split, ... four rows in
// flight each, partial sums combined by a butterfly over the groups -- the order depends only on (split, lpr): reproducible.
template <typename T, typename TO>
__device__ __forceinline__ void segment_sum_body(const T* __restrict__ msg, const int32_t* __restrict__ off,
                                                 const int32_t* __restrict__ perm, const TO* __restrict__ init, TO* __restrict__ out,
                                                 int64_t N, int chunks, int lpr_log2, int split_log2, int block, int nblocks) {
    constexpr int NV = Row16<T>::N;
    const int lane = threadIdx.x & 63;
    const int lpr = 1 << lpr_log2, split = 1 << split_log2;
    const int chunk = lane & (lpr - 1);
    const int grp = lane >> lpr_log2;               // lane group in the wave
    const int g = grp & (split - 1);                // position among the groups of its segment
    const int seg_in_wave = grp >> split_log2;
    const int spw = (64 >> lpr_log2) >> split_log2; // segments per wave pass
    const int64_t wave = (int64_t)block * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)nblocks * (blockDim.x >> 6);
    const int64_t row_v = chunks;                   // row length in 16-byte vectors
    for (int64_t n0 = wave * spw; n0 < N; n0 += nwaves * spw) {
        const int64_t n = n0 + seg_in_wave;
        const bool live = n < N;
        const int j0 = live ? off[n] : 0, j1 = live ? off[n + 1] : 0;
        for (int c0 = 0; c0 < chunks; c0 += lpr) {
            const int c = c0 + chunk;
            const bool cok = c < chunks;
            float acc[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) acc[i] = 0.f;
            for (int j = j0 + g; j < j1; j += 4 * split) {
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u * split;
                    v[u] = u32x4{0, 0, 0, 0};
                    if (jj < j1 && cok) {
                        const int64_t r = perm ? perm[jj] : jj;
                        v[u] = reinterpret_cast<const u32x4*>(msg)[r * row_v + c];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[NV];
                    Row16<T>::unpack(v[u], f);
#pragma unroll
                    for (int i = 0; i < NV; ++i) acc[i] += f[i];
                }
            }
            // combine the groups of one segment (wave-uniform trip count: every lane takes part)
            for (int o = lpr; o < lpr * split; o <<= 1)
#pragma unroll
                for (int i = 0; i < NV; ++i) acc[i] += __shfl_xor(acc[i], o, 64);
            if (live && cok && g == 0) {
                // T row of NV values -> TO storage: NV*sizeof(TO) bytes at element offset c*NV
                TO* dst = out + n * (row_v * NV) + (int64_t)c * NV;
                if (init) {
                    const TO* src = init + n * (row_v * NV) + (int64_t)c * NV;
#pragma unroll
                    for (int i = 0; i < NV; ++i) acc[i] += to_f32<TO>(src[i]);
                }
                if (sizeof(TO) == sizeof(T)) {
                    *reinterpret_cast<u32x4*>(dst) = Row16<TO>::pack(acc);
                } else {   // bf16 messages summed into fp32 rows: two 16-byte stores
                    float* d = reinterpret_cast<float*>(dst);
#pragma unroll
                    for (int q = 0; q < NV / 4; ++q) *reinterpret_cast<u32x4*>(d + 4 * q) = Row16<float>::pack(acc + 4 * q);
                }
            }
        }
    }
}

template <typename T, typename TO>
__global__ void __launch_bounds__(256)
    segment_sum_kernel(const T* __restrict__ msg, const int32_t* __restrict__ off, const int32_t* __restrict__ perm,
                       const TO* __restrict__ init, TO* __restrict__ out, int64_t N, int chunks, int lpr_log2, int split_log2) {
    segment_sum_body<T, TO>(msg, off, perm, init, out, N, chunks, lpr_log2, split_log2, blockIdx.x, gridDim.x);
}

// Two segment sums over the SAME rows in one launch (the two adjoints of an edge MLP's gathers: the gradient rows summed by sender
// and by receiver): blocks [0, blocks_a) take the first CSR structure, the rest the second.  Each half computes exactly what its
// own launch would (same block count, same order).
struct SegSide {
    const int32_t* off;
    const int32_t* perm;
    void* out;
    int64_t N;
    int split_log2, blocks;
};
template <typename T>
__global__ void __launch_bounds__(256) segment_sum_pair_kernel(const T* __restrict__ msg, SegSide a, SegSide b, int chunks, int lpr_log2) {
    if ((int)blockIdx.x < a.blocks)
        segment_sum_body<T, T>(msg, a.off, a.perm, nullptr, (T*)a.out, a.N, chunks, lpr_log2, a.split_log2, blockIdx.x, a.blocks);
    else
        segment_sum_body<T, T>(msg, b.off, b.perm, nullptr, (T*)b.out, b.N, chunks, lpr_log2, b.split_log2, blockIdx.x - a.blocks, b.blocks);
}

static inline int ceil_log2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

template <typename T, int ACT>
int launch_gather_add(const void* base, const void* a, const int32_t* ia, const void* b, const int32_t* ib, const void* dh,
                      void* out, int64_t E, int chunks, hipStream_t stream) {
    int lpr_log2 = ceil_log2(chunks);
    if (lpr_log2 > 6) lpr_log2 = 6;
    const int rpw = 64 >> lpr_log2;
    int64_t waves = (E + 2 * rpw - 1) / (2 * rpw);
    int64_t blocks = (waves + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (dh)
        hipLaunchKernelGGL((edge_gather_add_kernel<T, ACT, 1>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T*)base,
                           (const T*)a, ia, (const T*)b, ib, (const T*)dh, (T*)out, E, chunks, lpr_log2);
    else
        hipLaunchKernelGGL((edge_gather_add_kernel<T, ACT, 0>), dim3((unsigned)blocks), dim3(256), 0, stream, (const T*)base,
                           (const T*)a, ia, (const T*)b, ib, (const T*)nullptr, (T*)out, E, chunks, lpr_log2);
    P4C_CHECK_LAUNCH("edge_gather_add");
    return P4C_OK;
}

template <typename T>
int dispatch_gather_add(int act, const void* base, const void* a, const int32_t* ia, const void* b, const int32_t* ib,
                        const void* dh, void* out, int64_t E, int chunks, hipStream_t stream) {
    switch (act) {
        case P4C_ACT_NONE: return launch_gather_add<T, 0>(base, a, ia, b, ib, dh, out, E, chunks, stream);
        case P4C_ACT_RELU: return launch_gather_add<T, 1>(base, a, ia, b, ib, dh, out, E, chunks, stream);
        case P4C_ACT_SILU: return launch_gather_add<T, 2>(base, a, ia, b, ib, dh, out, E, chunks, stream);
    }
    return fail(P4C_ERR_INVALID, "edge_gather_add: unknown activation %d", act);
}

int gather_add_common(const char* name, const void* base, const void* a, const int32_t* ia, const void* b, const int32_t* ib,
                      const void* dh, void* out, int64_t E, int C, int dtype, int act, hipStream_t stream) {
    P4C_CHECK_ARG(out != nullptr, "%s: out is NULL", name);
    P4C_CHECK_ARG(E >= 0 && C > 0, "%s: bad sizes E=%lld C=%d", name, (long long)E, C);
    P4C_CHECK_ARG(dtype == P4C_F32 || dtype == P4C_BF16, "%s: dtype must be P4C_F32 or P4C_BF16", name);
    const int esz = dtype == P4C_F32 ? 4 : 2;
    P4C_CHECK_ARG((C * esz) % 16 == 0, "%s: a row (C=%d x %d B) must be a multiple of 16 bytes", name, C, esz);
    P4C_CHECK_ARG((a == nullptr) == (ia == nullptr) && (b == nullptr) == (ib == nullptr), "%s: a/ia, b/ib go together", name);
    if (E == 0) return P4C_OK;
    const int chunks = C * esz / 16;
    return dtype == P4C_F32 ? dispatch_gather_add<float>(act, base, a, ia, b, ib, dh, out, E, chunks, stream)
                            : dispatch_gather_add<bf16>(act, base, a, ia, b, ib, dh, out, E, chunks, stream);
}

// launch shape of a segment sum: lane groups per segment from the mean list length (one group moves 4 rows at a time)
int64_t segment_sum_shape(int64_t N, int64_t E, int chunks, int* lpr_log2_out, int* split_log2_out) {
    int lpr_log2 = ceil_log2(chunks);
    if (lpr_log2 > 6) lpr_log2 = 6;
    const int groups_log2 = 6 - lpr_log2;
    const int64_t mean_len = (E + N - 1) / N;
    int split_log2 = 0;
    while (split_log2 < groups_log2 && (int64_t)(4 << split_log2) < mean_len) ++split_log2;
    const int spw = (64 >> lpr_log2) >> split_log2;
    int64_t waves = (N + spw - 1) / spw;
    int64_t blocks = (waves + 3) / 4;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    *lpr_log2_out = lpr_log2;
    *split_log2_out = split_log2;
    return blocks;
}

}  // namespace
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_edge_gather_add_fwd(const void* base, const void* a, const int32_t* ia, const void* b, const int32_t* ib,
                                       void* out, int64_t E, int C, int dtype, int act, p4c_stream_t stream) {
    return gather_add_common("p4c_edge_gather_add_fwd", base, a, ia, b, ib, nullptr, out, E, C, dtype, act, as_stream(stream));
}

extern "C" int p4c_edge_gather_add_bwd(const void* dh, const void* base, const void* a, const int32_t* ia, const void* b,
                                       const int32_t* ib, void* dpre, int64_t E, int C, int dtype, int act, p4c_stream_t stream) {
    P4C_CHECK_ARG(dh != nullptr, "p4c_edge_gather_add_bwd: dh is NULL");
    return gather_add_common("p4c_edge_gather_add_bwd", base, a, ia, b, ib, dh, dpre, E, C, dtype, act, as_stream(stream));
}

extern "C" int p4c_segment_sum(const void* msg, const int32_t* offsets, const int32_t* perm, const void* init, void* out,
                               int64_t N, int64_t E, int C, int dtype, int out_dtype, p4c_stream_t stream) {
    P4C_CHECK_ARG(out != nullptr && offsets != nullptr, "p4c_segment_sum: NULL pointer");
    P4C_CHECK_ARG(msg != nullptr || E == 0, "p4c_segment_sum: msg is NULL");
    P4C_CHECK_ARG(N >= 0 && E >= 0 && C > 0, "p4c_segment_sum: bad sizes");
    P4C_CHECK_ARG(dtype == P4C_F32 || dtype == P4C_BF16, "p4c_segment_sum: dtype must be P4C_F32 or P4C_BF16");
    P4C_CHECK_ARG(out_dtype == dtype || (dtype == P4C_BF16 && out_dtype == P4C_F32),
                  "p4c_segment_sum: out_dtype must equal dtype, or be P4C_F32 for bf16 messages");
    const int esz = dtype == P4C_F32 ? 4 : 2;
    P4C_CHECK_ARG((C * esz) % 16 == 0, "p4c_segment_sum: a row (C=%d x %d B) must be a multiple of 16 bytes", C, esz);
    if (N == 0) return P4C_OK;
    const int chunks = C * esz / 16;
    int lpr_log2, split_log2;
    const int64_t blocks = segment_sum_shape(N, E, chunks, &lpr_log2, &split_log2);
    hipStream_t s = as_stream(stream);
    if (dtype == P4C_F32)
        hipLaunchKernelGGL((segment_sum_kernel<float, float>), dim3((unsigned)blocks), dim3(256), 0, s, (const float*)msg, offsets,
                           perm, (const float*)init, (float*)out, N, chunks, lpr_log2, split_log2);
    else if (out_dtype == P4C_BF16)
        hipLaunchKernelGGL((segment_sum_kernel<bf16, bf16>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16*)msg, offsets, perm,
                           (const bf16*)init, (bf16*)out, N, chunks, lpr_log2, split_log2);
    else
        hipLaunchKernelGGL((segment_sum_kernel<bf16, float>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16*)msg, offsets, perm,
                           (const float*)init, (float*)out, N, chunks, lpr_log2, split_log2);
    P4C_CHECK_LAUNCH("segment_sum");
    return P4C_OK;
}

extern "C" int p4c_segment_sum_pair(const void* msg, const int32_t* offsets_a, const int32_t* perm_a, void* out_a, int64_t Na,
                                    const int32_t* offsets_b, const int32_t* perm_b, void* out_b, int64_t Nb, int64_t E, int C, int dtype,
                                    p4c_stream_t stream) {
    P4C_CHECK_ARG(msg && offsets_a && offsets_b && out_a && out_b, "p4c_segment_sum_pair: NULL pointer");
    P4C_CHECK_ARG(Na > 0 && Nb > 0 && E > 0 && C > 0, "p4c_segment_sum_pair: bad sizes (use p4c_segment_sum for empty sides)");
    P4C_CHECK_ARG(dtype == P4C_F32 || dtype == P4C_BF16, "p4c_segment_sum_pair: dtype must be P4C_F32 or P4C_BF16");
    const int esz = dtype == P4C_F32 ? 4 : 2;
    P4C_CHECK_ARG((C * esz) % 16 == 0, "p4c_segment_sum_pair: a row (C=%d x %d B) must be a multiple of 16 bytes", C, esz);
    const int chunks = C * esz / 16;
    int lpr_log2, lpr_b;
    SegSide a{offsets_a, perm_a, out_a, Na, 0, 0}, b{offsets_b, perm_b, out_b, Nb, 0, 0};
    a.blocks = (int)segment_sum_shape(Na, E, chunks, &lpr_log2, &a.split_log2);
    b.blocks = (int)segment_sum_shape(Nb, E, chunks, &lpr_b, &b.split_log2);
    hipStream_t s = as_stream(stream);
    if (dtype == P4C_F32)
        hipLaunchKernelGGL((segment_sum_pair_kernel<float>), dim3((unsigned)(a.blocks + b.blocks)), dim3(256), 0, s, (const float*)msg, a, b,
                           chunks, lpr_log2);
    else
        hipLaunchKernelGGL((segment_sum_pair_kernel<bf16>), dim3((unsigned)(a.blocks + b.blocks)), dim3(256), 0, s, (const bf16*)msg, a, b,
                           chunks, lpr_log2);
    P4C_CHECK_LAUNCH("segment_sum_pair");
    return P4C_OK;
}
