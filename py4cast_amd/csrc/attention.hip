// Windowed multi-head self-attention of the Swin path (model_name SwinUNetR: config/CLI/model/swinunetr.yaml:19-30 -- depths
// [2,2,2,2], heads [3,6,12,24], feature_size 24 => head_dim 8; the network itself comes from mfai, py4cast/models.py:10-20).
// A Swin block computes, per (window, head):  softmax(q k^T * scale + rel_pos_bias [+ shift mask]) v  on the ws x ws tokens of a
// window of the cyclically shifted token grid.  torch runs it as roll -> window_partition (copy) -> reshape/permute (copy) ->
// bmm -> add -> add -> softmax -> bmm -> permute (copy) -> window_reverse (copy) -> roll back.  Here it is ONE kernel over the
// qkv tensor in its natural (B, Hp, Wp, 3, heads, head_dim) layout (the output of the qkv Linear on the features-last grid): the
// shift, the window partition and their inverses are index arithmetic on the loads / stores.
//
// One wave per (window, head) task, N = ws*ws <= 64 tokens, head_dim D in {8, 16, 32}; v_mfma_f32_32x32x16_bf16 throughout:
//   S^T = K Q^T (key on the register rows, query on the lane) -- operands are the 16-byte token rows straight from HBM;
//   softmax over the keys is then a per-lane reduction (+ one exchange between the two lane halves);
//   the S^T accumulators, rounded to bf16, ARE the B operand of O^T = V^T P^T (no LDS round trip, cdna_hip_programming.md
//   "An accumulator tile as the next MFMA's operand"); V^T comes from the LDS-staged [token][channel] image with
//   ds_read_b64_tr_b16.
// Backward recomputes the softmax in both orientations (key-on-lane for dV, dK; query-on-lane for dQ and the bias gradient), so
// no attention matrix is ever stored or transposed through memory; the relative-position-bias gradient accumulates in registers
// over the windows of a persistent wave (one head per workgroup) and is reduced in a fixed order (no atomics).
#include <stdlib.h>

#include "kernels.hpp"

namespace p4c {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr int VROW = 64;               // bytes per token row of an LDS image: 32 bf16 channel slots
constexpr int IMG = 64 * VROW;         // one [token][channel] image: 4 KB
constexpr int BIAS_LD = 65;            // bias image [key][query] fp32, rows padded: conflict-free along either index
constexpr int BIAS_BYTES = 64 * BIAS_LD * 4;
constexpr int STAT_BYTES = 3 * 64 * 4; // per wave: row max, 1/row sum, delta
constexpr float MASK_VALUE = -100.0f;  // Swin's attn_mask fill value

struct Geo {
    int B, Hp, Wp, heads, ws, shift, nwy, nwx, N, C;
};

__device__ __forceinline__ int rowmap(int i) { return (i & 3) + 8 * (i >> 2); }   // + 4 * lane half

template <typename T> __device__ __forceinline__ bf16x8 load8(const T* p);
template <> __device__ __forceinline__ bf16x8 load8<float>(const float* p) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    bf16x8 r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
}
template <> __device__ __forceinline__ bf16x8 load8<bf16>(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)0.f;
    return r;
}

__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}
__device__ __forceinline__ void store4(bf16* p, float a, float b, float c, float d) {
    bf16x4 o;
    o[0] = (__bf16)a; o[1] = (__bf16)b; o[2] = (__bf16)c; o[3] = (__bf16)d;
    *reinterpret_cast<bf16x4*>(p) = o;
}

// registers 8s .. 8s+7 of an accumulator tile as the operand of k-step s of the next product (k order permuted: element j of
// lane half h is row 16s + 8(j>>2) + 4h + (j&3); read_tr() below delivers the other operand in the same order)
__device__ __forceinline__ bf16x8 acc_op(const float* x, int s) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)x[8 * s + j];
    return r;
}

// transposed operand of k-step sp (tokens 16sp .. 16sp+15) from a [token][channel] image: lane (m = lane&31, h) receives
// image[token 16sp + 8(j>>2) + 4h + (j&3)][channel m], j = 0..7
__device__ __forceinline__ bf16x8 read_tr(const char* img, int sp, int lane) {
    const int i = lane & 15, tg = (lane >> 4) & 1, h = lane >> 5;
    const char* p = img + (16 * sp + 4 * h + (i >> 2)) * VROW + (tg * 16 + (i & 3) * 4) * 2;
    union { s16x4 s[2]; bf16x8 v; } u;
    u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 8 * VROW));
    return u.v;
}

__device__ __forceinline__ void lds_order() {   // LDS accesses of one wave execute in order; keep the compiler from moving them
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// per-lane constants of the 32 register slots (tile t = slot >> 4, register i = slot & 15): token = 32 t + rowmap(i) + 4h
struct SlotTables {
    unsigned int valid, yge, xge;   // token < N;  token row / column inside the window >= ws - shift
};
__device__ __forceinline__ SlotTables slot_tables(const Geo& g, int h) {
    SlotTables t{0u, 0u, 0u};
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) {
        const int tok = 32 * (sl >> 4) + rowmap(sl & 15) + 4 * h;
        const int ty = tok / g.ws, tx = tok - ty * g.ws;
        if (tok < g.N) t.valid |= 1u << sl;
        if (ty >= g.ws - g.shift) t.yge |= 1u << sl;
        if (tx >= g.ws - g.shift) t.xge |= 1u << sl;
    }
    return t;
}

// the lane's own two tokens (tile 0 / 1): position inside the window
struct OwnTok {
    int wy[2], wx[2];
    bool ok[2];
};
__device__ __forceinline__ OwnTok own_tokens(const Geo& g, int r) {
    OwnTok o;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int tok = 32 * t + r;
        o.ok[t] = tok < g.N;
        o.wy[t] = tok / g.ws;
        o.wx[t] = tok - o.wy[t] * g.ws;
    }
    return o;
}

struct Win {
    int b, wi, wj;
    bool last_y, last_x;
};
__device__ __forceinline__ Win decode_window(const Geo& g, int w) {
    Win q;
    q.wj = w % g.nwx;
    const int rest = w / g.nwx;
    q.wi = rest % g.nwy;
    q.b = rest / g.nwy;
    q.last_y = g.shift > 0 && q.wi == g.nwy - 1;
    q.last_x = g.shift > 0 && q.wj == g.nwx - 1;
    return q;
}
// pixel index (b, y, x flattened) of the lane's token in the UNSHIFTED grid
__device__ __forceinline__ int64_t token_pixel(const Geo& g, const Win& w, int wy, int wx) {
    int y = w.wi * g.ws + wy + g.shift, x = w.wj * g.ws + wx + g.shift;
    y -= (y >= g.Hp) ? g.Hp : 0;
    x -= (x >= g.Wp) ? g.Wp : 0;
    return ((int64_t)w.b * g.Hp + y) * g.Wp + x;
}
// slots whose token lies in another shift region than the lane's own token (mask of lightning-independent Swin semantics:
// region = (row >= ws - shift, col >= ws - shift) inside the last window row / column of the shifted grid)
__device__ __forceinline__ unsigned int masked_slots(const SlotTables& st, const Win& w, bool own_yge, bool own_xge) {
    unsigned int m = 0u;
    if (w.last_y) m |= own_yge ? ~st.yge : st.yge;
    if (w.last_x) m |= own_xge ? ~st.xge : st.xge;
    return m;
}

// operand registers of one token row: k-step s holds channels 16s + 8h .. +7 (zero beyond D or for padding tokens)
template <typename T, int D>
__device__ __forceinline__ void load_rows(const T* base, int64_t pix, int64_t row_stride, bool ok, int h, bf16x8* op) {
    constexpr int KS = (D + 15) / 16;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 16 * s + 8 * h;
        op[s] = (ok && c < D) ? load8<T>(base + pix * row_stride + c) : zero8();
    }
}
template <int D>
__device__ __forceinline__ void write_image(char* img, int tok, int h, const bf16x8* op) {
    constexpr int KS = (D + 15) / 16;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int c = 16 * s + 8 * h;
        if (c < D) *reinterpret_cast<bf16x8*>(img + tok * VROW + c * 2) = op[s];
    }
}

__device__ __forceinline__ void stage_bias(float* lbias, const float* bias_t, int head, int N) {
    for (int i = threadIdx.x; i < 64 * 64; i += blockDim.x) {
        const int k = i >> 6, q = i & 63;
        lbias[k * BIAS_LD + q] = (bias_t && k < N && q < N) ? bias_t[((int64_t)head * N + k) * N + q] * LOG2E : 0.f;
    }
}

// scores of one lane-column tile (lane token = "own", register slots = "other"): sv[sl] = logit * log2(e); returns the row max
// over the lane's slots (both halves combined).  own_is_query selects which index of the [key][query] bias image the lane holds.
template <bool OWN_IS_QUERY>
__device__ __forceinline__ void logits(const f32x16& a0, const f32x16& a1, const float* lbias, int own_tok, int h, float scale_l2,
                                       unsigned int masked, float* sv) {
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) {
        const int other = 32 * (sl >> 4) + rowmap(sl & 15) + 4 * h;
        const float acc = (sl < 16) ? a0[sl & 15] : a1[sl & 15];
        const float bias = OWN_IS_QUERY ? lbias[other * BIAS_LD + own_tok] : lbias[own_tok * BIAS_LD + other];
        float v = acc * scale_l2 + bias;
        if ((masked >> sl) & 1u) v += MASK_VALUE * LOG2E;
        sv[sl] = v;
    }
}

template <typename T, int D>
__global__ void __launch_bounds__(256, 2)
    window_attn_fwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_t, T* __restrict__ out, Geo g, float scale, int GW) {
    constexpr int KS = (D + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    char* img = smem + BIAS_BYTES + wv * IMG;
    // workgroup -> (window group gw, head): the `heads` workgroups that read the same token rows (the same windows, another channel
    // slice) get block ids equal modulo 8, i.e. land on ONE XCD, next to each other in launch order: its L2 serves the re-reads
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = slot % g.heads, gw = (slot / g.heads) * 8 + xcd;
    if (gw >= GW) return;
    stage_bias(lbias, bias_t, head, g.N);
    __syncthreads();
    const SlotTables st = slot_tables(g, h);
    const OwnTok own = own_tokens(g, r);
    const int nwin = g.B * g.nwy * g.nwx;
    const int64_t rs = 3 * (int64_t)g.C;
    const float scale_l2 = scale * LOG2E;
    const int ksteps = (g.N + 15) >> 4;

    for (int w = gw * 4 + wv; w < nwin; w += GW * 4) {
        const Win win = decode_window(g, w);
        int64_t pix[2];
        bf16x8 qop[2][KS], kop[2][KS];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            pix[t] = token_pixel(g, win, own.wy[t], own.wx[t]);
            bf16x8 vop[KS];
            load_rows<T, D>(qkv + head * D, pix[t], rs, own.ok[t], h, qop[t]);
            load_rows<T, D>(qkv + g.C + head * D, pix[t], rs, own.ok[t], h, kop[t]);
            load_rows<T, D>(qkv + 2 * g.C + head * D, pix[t], rs, own.ok[t], h, vop);
            write_image<D>(img, 32 * t + r, h, vop);
        }
        lds_order();
        bf16x8 vt[4];
#pragma unroll
        for (int sp = 0; sp < 4; ++sp) vt[sp] = (sp < ksteps) ? read_tr(img, sp, lane) : zero8();
        lds_order();

#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 a[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                a[kt] = zero16();
#pragma unroll
                for (int s = 0; s < KS; ++s) a[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kop[kt][s], qop[qt][s], a[kt], 0, 0, 0);
            }
            const unsigned int masked = masked_slots(st, win, own.wy[qt] >= g.ws - g.shift, own.wx[qt] >= g.ws - g.shift);
            float sv[32];
            logits<true>(a[0], a[1], lbias, 32 * qt + r, h, scale_l2, masked, sv);
            float mx = -INFINITY;
#pragma unroll
            for (int sl = 0; sl < 32; ++sl) {
                if (!((st.valid >> sl) & 1u)) sv[sl] = -INFINITY;
                mx = fmaxf(mx, sv[sl]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int sl = 0; sl < 32; ++sl) {
                sv[sl] = __builtin_amdgcn_exp2f(sv[sl] - mx);
                sum += sv[sl];
            }
            sum += __shfl_xor(sum, 32, 64);
            f32x16 o = zero16();
#pragma unroll
            for (int sp = 0; sp < 4; ++sp)
                if (sp < ksteps) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vt[sp], acc_op(sv + 16 * (sp >> 1), sp & 1), o, 0, 0, 0);
            const float inv = 1.f / sum;
            if (own.ok[qt]) {
                T* dst = out + pix[qt] * g.C + head * D + 4 * h;
#pragma unroll
                for (int q4 = 0; q4 < D / 8; ++q4)
                    store4(dst + 8 * q4, o[4 * q4] * inv, o[4 * q4 + 1] * inv, o[4 * q4 + 2] * inv, o[4 * q4 + 3] * inv);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward: dqkv (B,Hp,Wp,3,heads,D) and, per workgroup, a partial of the bias gradient [key][query] (64 x 64 floats)
template <typename T, int D>
__global__ void __launch_bounds__(256, 1)
    window_attn_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ bias_t, const T* __restrict__ dout,
                           T* __restrict__ dqkv, float* __restrict__ dbias_partial, Geo g, float scale, int GW) {
    constexpr int KS = (D + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    char* wbase = smem + BIAS_BYTES + wv * (3 * IMG + STAT_BYTES);
    char* imgK = wbase;
    char* imgQ = wbase + IMG;
    char* imgO = wbase + 2 * IMG;
    float* st_m = reinterpret_cast<float*>(wbase + 3 * IMG);
    float* st_il = st_m + 64;
    float* st_d = st_m + 128;
    // workgroup -> (window group gw, head): the `heads` workgroups that read the same token rows (the same windows, another channel
    // slice) get block ids equal modulo 8, i.e. land on ONE XCD, next to each other in launch order: its L2 serves the re-reads
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = slot % g.heads, gw = (slot / g.heads) * 8 + xcd;
    if (gw >= GW) return;
    stage_bias(lbias, bias_t, head, g.N);
    __syncthreads();
    const SlotTables st = slot_tables(g, h);
    const OwnTok own = own_tokens(g, r);
    const int nwin = g.B * g.nwy * g.nwx;
    const int64_t rs = 3 * (int64_t)g.C;
    const float scale_l2 = scale * LOG2E;
    const int ksteps = (g.N + 15) >> 4;

    float dbias[64];   // [qt][slot]: query 32 qt + r on the lane, key of the slot
#pragma unroll
    for (int i = 0; i < 64; ++i) dbias[i] = 0.f;

    for (int w = gw * 4 + wv; w < nwin; w += GW * 4) {
        const Win win = decode_window(g, w);
        int64_t pix[2];
        bool oyge[2], oxge[2];
        bf16x8 qop[2][KS], kop[2][KS], vop[2][KS], dop[2][KS];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            pix[t] = token_pixel(g, win, own.wy[t], own.wx[t]);
            oyge[t] = own.wy[t] >= g.ws - g.shift;
            oxge[t] = own.wx[t] >= g.ws - g.shift;
            load_rows<T, D>(qkv + head * D, pix[t], rs, own.ok[t], h, qop[t]);
            load_rows<T, D>(qkv + g.C + head * D, pix[t], rs, own.ok[t], h, kop[t]);
            load_rows<T, D>(qkv + 2 * g.C + head * D, pix[t], rs, own.ok[t], h, vop[t]);
            load_rows<T, D>(dout + head * D, pix[t], g.C, own.ok[t], h, dop[t]);
            write_image<D>(imgK, 32 * t + r, h, kop[t]);
            write_image<D>(imgQ, 32 * t + r, h, qop[t]);
            write_image<D>(imgO, 32 * t + r, h, dop[t]);
        }
        lds_order();

        // ---- phase 1: query on the lane.  dS^T -> bias gradient and dQ
        {
            bf16x8 kt_tr[4];
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) kt_tr[sp] = (sp < ksteps) ? read_tr(imgK, sp, lane) : zero8();
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                f32x16 a[2], dp[2];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    a[kt] = zero16();
                    dp[kt] = zero16();
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        a[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kop[kt][s], qop[qt][s], a[kt], 0, 0, 0);
                        dp[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vop[kt][s], dop[qt][s], dp[kt], 0, 0, 0);
                    }
                }
                const unsigned int masked = masked_slots(st, win, oyge[qt], oxge[qt]);
                float sv[32];
                logits<true>(a[0], a[1], lbias, 32 * qt + r, h, scale_l2, masked, sv);
                float mx = -INFINITY;
#pragma unroll
                for (int sl = 0; sl < 32; ++sl) {
                    if (!((st.valid >> sl) & 1u)) sv[sl] = -INFINITY;
                    mx = fmaxf(mx, sv[sl]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                float sum = 0.f;
#pragma unroll
                for (int sl = 0; sl < 32; ++sl) {
                    sv[sl] = __builtin_amdgcn_exp2f(sv[sl] - mx);
                    sum += sv[sl];
                }
                sum += __shfl_xor(sum, 32, 64);
                const float inv = 1.f / sum;
                float delta = 0.f;
#pragma unroll
                for (int sl = 0; sl < 32; ++sl) {
                    sv[sl] *= inv;
                    delta += sv[sl] * ((sl < 16) ? dp[0][sl & 15] : dp[1][sl & 15]);
                }
                delta += __shfl_xor(delta, 32, 64);
#pragma unroll
                for (int sl = 0; sl < 32; ++sl) {
                    sv[sl] *= ((sl < 16) ? dp[0][sl & 15] : dp[1][sl & 15]) - delta;   // dS^T
                    dbias[32 * qt + sl] += sv[sl];
                }
                if (h == 0) {
                    st_m[32 * qt + r] = mx;
                    st_il[32 * qt + r] = inv;
                    st_d[32 * qt + r] = delta;
                }
                f32x16 dq = zero16();
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
                    if (sp < ksteps)
                        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kt_tr[sp], acc_op(sv + 16 * (sp >> 1), sp & 1), dq, 0, 0, 0);
                if (own.ok[qt]) {
                    T* dst = dqkv + pix[qt] * rs + head * D + 4 * h;
#pragma unroll
                    for (int q4 = 0; q4 < D / 8; ++q4)
                        store4(dst + 8 * q4, dq[4 * q4] * scale, dq[4 * q4 + 1] * scale, dq[4 * q4 + 2] * scale, dq[4 * q4 + 3] * scale);
                }
            }
        }
        lds_order();

        // ---- phase 2: key on the lane.  P and dS (rows = queries) -> dV and dK
        {
            bf16x8 q_tr[4], o_tr[4];
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                q_tr[sp] = (sp < ksteps) ? read_tr(imgQ, sp, lane) : zero8();
                o_tr[sp] = (sp < ksteps) ? read_tr(imgO, sp, lane) : zero8();
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                f32x16 a[2], dp[2];
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    a[qt] = zero16();
                    dp[qt] = zero16();
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        a[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qop[qt][s], kop[kt][s], a[qt], 0, 0, 0);
                        dp[qt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dop[qt][s], vop[kt][s], dp[qt], 0, 0, 0);
                    }
                }
                const unsigned int masked = masked_slots(st, win, oyge[kt], oxge[kt]);
                float pv[32], ds[32];
                logits<false>(a[0], a[1], lbias, 32 * kt + r, h, scale_l2, masked, pv);
#pragma unroll
                for (int g4 = 0; g4 < 8; ++g4) {   // slots 4 g4 .. 4 g4 + 3 = queries 32 (g4>>2) + 8 (g4&3) + 4h + (0..3)
                    const int q0 = 32 * (g4 >> 2) + 8 * (g4 & 3) + 4 * h;
                    const f32x4 m4 = *reinterpret_cast<const f32x4*>(st_m + q0);
                    const f32x4 il4 = *reinterpret_cast<const f32x4*>(st_il + q0);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(st_d + q0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int sl = 4 * g4 + e;
                        float p = __builtin_amdgcn_exp2f(pv[sl] - m4[e]) * il4[e];
                        if (!own.ok[kt]) p = 0.f;
                        pv[sl] = p;
                        ds[sl] = p * (((sl < 16) ? dp[0][sl & 15] : dp[1][sl & 15]) - d4[e]);
                    }
                }
                f32x16 dv = zero16(), dk = zero16();
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
                    if (sp < ksteps) {
                        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o_tr[sp], acc_op(pv + 16 * (sp >> 1), sp & 1), dv, 0, 0, 0);
                        dk = __builtin_amdgcn_mfma_f32_32x32x16_bf16(q_tr[sp], acc_op(ds + 16 * (sp >> 1), sp & 1), dk, 0, 0, 0);
                    }
                if (own.ok[kt]) {
                    T* dstk = dqkv + pix[kt] * rs + g.C + head * D + 4 * h;
                    T* dstv = dqkv + pix[kt] * rs + 2 * g.C + head * D + 4 * h;
#pragma unroll
                    for (int q4 = 0; q4 < D / 8; ++q4) {
                        store4(dstk + 8 * q4, dk[4 * q4] * scale, dk[4 * q4 + 1] * scale, dk[4 * q4 + 2] * scale, dk[4 * q4 + 3] * scale);
                        store4(dstv + 8 * q4, dv[4 * q4], dv[4 * q4 + 1], dv[4 * q4 + 2], dv[4 * q4 + 3]);
                    }
                }
            }
        }
        lds_order();
    }

    // ---- bias gradient of this workgroup: the four waves add their registers in wave order (fixed order, no atomics)
    if (dbias_partial) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + BIAS_BYTES);   // [key][query] 64 x 64 floats over the waves' images
        for (int turn = 0; turn < 4; ++turn) {
            if (wv == turn) {
#pragma unroll
                for (int i = 0; i < 64; ++i) {
                    const int qt = i >> 5, sl = i & 31;
                    const int key = 32 * (sl >> 4) + rowmap(sl & 15) + 4 * h, q = 32 * qt + r;
                    if (turn == 0) red[key * 64 + q] = dbias[i];
                    else red[key * 64 + q] += dbias[i];
                }
            }
            __syncthreads();
        }
        float* dst = dbias_partial + (int64_t)(gw * g.heads + head) * 4096;
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) dst[i] = red[i];
    }
}

// dbias_t[head][k][q] = sum over the workgroups of that head: 32 outputs x 8 slices of the workgroup index per block, the slices
// combined in slice order (fixed order; one thread walking all GW partials of an output took 35 us at GW = 256)
__global__ void __launch_bounds__(256) window_attn_dbias_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dbias_t,
                                                                       int heads, int N, int GW) {
    __shared__ float red[8][33];
    const int total = heads * N * N;
    const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    float s = 0.f;
    if (i < total) {
        const int q = i % N, k = (i / N) % N, hd = i / (N * N);
        const float* p = partial + (int64_t)hd * 4096 + k * 64 + q;
        for (int gw = sl; gw < GW; gw += 8) s += p[(int64_t)gw * heads * 4096];
    }
    red[sl][e] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
        float t = red[0][e];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += red[j][e];
        dbias_t[i] = t;
    }
}

// ---------------------------------------------------------------------------------------------
// fp32-EXACT flavour (fp32 activations: the parity flavour of SwinUNetR, held to 1e-4 against the float64 oracle like every other
// model).  Same task mapping and bias-gradient reduction as above, no matrix cores: one wave per (window, head), LANE = TOKEN, the
// window's K / V (backward: Q / dO too) rows as fp32 images in LDS which every lane reads row by row (broadcast reads); logits,
// softmax and both products are fp32 FMA chains over the channels / the keys in index order.  Forward: lane = query.  Backward phase 1:
// lane = query (P, delta = dO . O, dS -> bias gradient, dQ); phase 2: lane = key (P and dS recomputed from the stored row statistics ->
// dV, dK).  Slower than the bf16 kernels by the matrix cores' factor; its job is to be right, not fast.
template <int D> constexpr int exact_fwd_smem() { return BIAS_BYTES + 4 * 2 * 64 * D * 4; }
template <int D> constexpr int exact_bwd_smem() { return BIAS_BYTES + 4 * (4 * 64 * D * 4 + STAT_BYTES); }

struct KeyRegions {   // bit j: token j of a window lies at row / column >= ws - shift (the wrap-around part of the last windows)
    unsigned long long yge, xge;
};
__device__ __forceinline__ KeyRegions key_regions(const Geo& g) {
    KeyRegions k{0ull, 0ull};
    for (int j = 0; j < g.N; ++j) {
        const int jy = j / g.ws, jx = j - jy * g.ws;
        if (jy >= g.ws - g.shift) k.yge |= 1ull << j;
        if (jx >= g.ws - g.shift) k.xge |= 1ull << j;
    }
    return k;
}
// bit j: token j lies in another wrap-around region than the lane's own token (the shift mask; the relation is symmetric)
__device__ __forceinline__ unsigned long long masked_tokens(const KeyRegions& kr, const Win& w, bool own_yge, bool own_xge) {
    unsigned long long m = 0ull;
    if (w.last_y) m |= own_yge ? ~kr.yge : kr.yge;
    if (w.last_x) m |= own_xge ? ~kr.xge : kr.xge;
    return m;
}
template <int D>
__device__ __forceinline__ void load_row_f32(const float* p, bool ok, float* r) {
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 a = ok ? *reinterpret_cast<const f32x4*>(p + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        r[c] = a.x; r[c + 1] = a.y; r[c + 2] = a.z; r[c + 3] = a.w;
    }
}
template <int D>
__device__ __forceinline__ void put_row_f32(float* img, int tok, const float* r) {
#pragma unroll
    for (int c = 0; c < D; c += 4) *reinterpret_cast<f32x4*>(img + tok * D + c) = f32x4{r[c], r[c + 1], r[c + 2], r[c + 3]};
}
template <int D>
__device__ __forceinline__ float dot_row(const float* own, const float* img, int tok) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(img + tok * D + c);
        acc = __builtin_fmaf(own[c], o.x, acc);
        acc = __builtin_fmaf(own[c + 1], o.y, acc);
        acc = __builtin_fmaf(own[c + 2], o.z, acc);
        acc = __builtin_fmaf(own[c + 3], o.w, acc);
    }
    return acc;
}
template <int D>
__device__ __forceinline__ void axpy_row(float a, const float* img, int tok, float* y) {
#pragma unroll
    for (int c = 0; c < D; c += 4) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(img + tok * D + c);
        y[c] = __builtin_fmaf(a, o.x, y[c]);
        y[c + 1] = __builtin_fmaf(a, o.y, y[c + 1]);
        y[c + 2] = __builtin_fmaf(a, o.z, y[c + 2]);
        y[c + 3] = __builtin_fmaf(a, o.w, y[c + 3]);
    }
}

template <int D>
__global__ void __launch_bounds__(256, 1)
    window_attn_fwd_exact_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_t, float* __restrict__ out, Geo g, float scale,
                                 int GW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* imgK = reinterpret_cast<float*>(smem + BIAS_BYTES) + wv * 2 * 64 * D;
    float* imgV = imgK + 64 * D;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = slot % g.heads, gw = (slot / g.heads) * 8 + xcd;
    if (gw >= GW) return;
    stage_bias(lbias, bias_t, head, g.N);
    __syncthreads();
    const KeyRegions kr = key_regions(g);
    const bool ok = lane < g.N;
    const int wy = lane / g.ws, wx = lane - wy * g.ws;
    const bool oyge = wy >= g.ws - g.shift, oxge = wx >= g.ws - g.shift;
    const int nwin = g.B * g.nwy * g.nwx;
    const int64_t rs = 3 * (int64_t)g.C;
    const float scale_l2 = scale * LOG2E;
    for (int w = gw * 4 + wv; w < nwin; w += GW * 4) {
        const Win win = decode_window(g, w);
        const int64_t pix = ok ? token_pixel(g, win, wy, wx) : 0;
        float q[D], t[D];
        load_row_f32<D>(qkv + pix * rs + head * D, ok, q);
        load_row_f32<D>(qkv + pix * rs + g.C + head * D, ok, t);
        put_row_f32<D>(imgK, lane, t);
        load_row_f32<D>(qkv + pix * rs + 2 * g.C + head * D, ok, t);
        put_row_f32<D>(imgV, lane, t);
        lds_order();
        const unsigned long long masked = masked_tokens(kr, win, oyge, oxge);
        float sv[64], mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 64; ++j) sv[j] = -INFINITY;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (j < g.N) {   // (uniform; its own basic block keeps the compiler from hoisting the reads of all 64 rows together)
                float v = dot_row<D>(q, imgK, j) * scale_l2 + lbias[j * BIAS_LD + lane];
                if ((masked >> j) & 1ull) v += MASK_VALUE * LOG2E;
                sv[j] = v;
                mx = fmaxf(mx, v);
            }
        }
        float sum = 0.f, o[D];
#pragma unroll
        for (int c = 0; c < D; ++c) o[c] = 0.f;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            if (j < g.N) {   // (uniform; its own basic block keeps the compiler from hoisting the reads of all 64 rows together)
                const float pj = __builtin_amdgcn_exp2f(sv[j] - mx);
                sum += pj;
                axpy_row<D>(pj, imgV, j, o);
            }
        }
        if (ok) {
            const float inv = 1.f / sum;
            float* dst = out + pix * g.C + head * D;
#pragma unroll
            for (int c = 0; c < D; c += 4) store4(dst + c, o[c] * inv, o[c + 1] * inv, o[c + 2] * inv, o[c + 3] * inv);
        }
        lds_order();
    }
}

template <int D>
__global__ void __launch_bounds__(256, 1)
    window_attn_bwd_exact_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_t, const float* __restrict__ dout,
                                 float* __restrict__ dqkv, float* __restrict__ dbias_partial, Geo g, float scale, int GW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lbias = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* wbase = reinterpret_cast<float*>(smem + BIAS_BYTES + wv * (4 * 64 * D * 4 + STAT_BYTES));
    float *imgK = wbase, *imgV = wbase + 64 * D, *imgQ = wbase + 2 * 64 * D, *imgO = wbase + 3 * 64 * D;
    float *st_m = wbase + 4 * 64 * D, *st_il = st_m + 64, *st_d = st_m + 128;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = slot % g.heads, gw = (slot / g.heads) * 8 + xcd;
    if (gw >= GW) return;
    stage_bias(lbias, bias_t, head, g.N);
    __syncthreads();
    const KeyRegions kr = key_regions(g);
    const bool ok = lane < g.N;
    const int wy = lane / g.ws, wx = lane - wy * g.ws;
    const bool oyge = wy >= g.ws - g.shift, oxge = wx >= g.ws - g.shift;
    const int nwin = g.B * g.nwy * g.nwx;
    const int64_t rs = 3 * (int64_t)g.C;
    const float scale_l2 = scale * LOG2E;
    float dbias[64];   // query = lane, key = index
#pragma unroll
    for (int j = 0; j < 64; ++j) dbias[j] = 0.f;

    for (int w = gw * 4 + wv; w < nwin; w += GW * 4) {
        const Win win = decode_window(g, w);
        const int64_t pix = ok ? token_pixel(g, win, wy, wx) : 0;
        const unsigned long long masked = masked_tokens(kr, win, oyge, oxge);
        float kown[D], vown[D];
        {
            // ---- phase 1: lane = query
            float q[D], dO[D];
            load_row_f32<D>(qkv + pix * rs + head * D, ok, q);
            load_row_f32<D>(qkv + pix * rs + g.C + head * D, ok, kown);
            load_row_f32<D>(qkv + pix * rs + 2 * g.C + head * D, ok, vown);
            load_row_f32<D>(dout + pix * g.C + head * D, ok, dO);
            put_row_f32<D>(imgQ, lane, q);
            put_row_f32<D>(imgK, lane, kown);
            put_row_f32<D>(imgV, lane, vown);
            put_row_f32<D>(imgO, lane, dO);
            lds_order();
            float pv[64], mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 64; ++j) pv[j] = -INFINITY;
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if (j < g.N) {   // (uniform; its own basic block keeps the compiler from hoisting the reads of all 64 rows together)
                    float v = dot_row<D>(q, imgK, j) * scale_l2 + lbias[j * BIAS_LD + lane];
                    if ((masked >> j) & 1ull) v += MASK_VALUE * LOG2E;
                    pv[j] = v;
                    mx = fmaxf(mx, v);
                }
            }
            float sum = 0.f, o[D];
#pragma unroll
            for (int c = 0; c < D; ++c) o[c] = 0.f;
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if (j < g.N) {   // (uniform; its own basic block keeps the compiler from hoisting the reads of all 64 rows together)
                    pv[j] = __builtin_amdgcn_exp2f(pv[j] - mx);
                    sum += pv[j];
                    axpy_row<D>(pv[j], imgV, j, o);
                }
            }
            const float inv = 1.f / sum;
            float delta = 0.f;   // sum_j P_ij dP_ij = dO_i . O_i
#pragma unroll
            for (int c = 0; c < D; ++c) delta = __builtin_fmaf(dO[c], o[c] * inv, delta);
            st_m[lane] = mx;
            st_il[lane] = inv;
            st_d[lane] = delta;
            float dq[D];
#pragma unroll
            for (int c = 0; c < D; ++c) dq[c] = 0.f;
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if (j < g.N) {   // (uniform; its own basic block keeps the compiler from hoisting the reads of all 64 rows together)
                    const float ds = pv[j] * inv * (dot_row<D>(dO, imgV, j) - delta);
                    dbias[j] += ok ? ds : 0.f;
                    axpy_row<D>(ds, imgK, j, dq);
                }
            }
            if (ok) {
                float* dst = dqkv + pix * rs + head * D;
#pragma unroll
                for (int c = 0; c < D; c += 4) store4(dst + c, dq[c] * scale, dq[c + 1] * scale, dq[c + 2] * scale, dq[c + 3] * scale);
            }
        }
        lds_order();
        {
            // ---- phase 2: lane = key
            float dk[D], dv[D];
#pragma unroll
            for (int c = 0; c < D; ++c) dk[c] = dv[c] = 0.f;
#pragma unroll
            for (int i = 0; i < 64; ++i) {
                if (i < g.N) {   // (uniform; its own basic block keeps the compiler from hoisting the reads of all 64 rows together)
                    float v = dot_row<D>(kown, imgQ, i) * scale_l2 + lbias[lane * BIAS_LD + i];
                    if ((masked >> i) & 1ull) v += MASK_VALUE * LOG2E;   // (the query's region against this lane's: symmetric)
                    const float pij = ok ? __builtin_amdgcn_exp2f(v - st_m[i]) * st_il[i] : 0.f;
                    const float ds = pij * (dot_row<D>(vown, imgO, i) - st_d[i]);
                    axpy_row<D>(pij, imgO, i, dv);
                    axpy_row<D>(ds, imgQ, i, dk);
                }
            }
            if (ok) {
                float* dstk = dqkv + pix * rs + g.C + head * D;
                float* dstv = dqkv + pix * rs + 2 * g.C + head * D;
#pragma unroll
                for (int c = 0; c < D; c += 4) {
                    store4(dstk + c, dk[c] * scale, dk[c + 1] * scale, dk[c + 2] * scale, dk[c + 3] * scale);
                    store4(dstv + c, dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
                }
            }
        }
        lds_order();
    }

    if (dbias_partial) {   // the four waves add their registers in wave order (fixed order, no atomics): [key][query]
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem + BIAS_BYTES);
        for (int turn = 0; turn < 4; ++turn) {
            if (wv == turn) {
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    if (j < g.N) {   // (uniform; its own basic block keeps the compiler from hoisting the reads of all 64 rows together)
                        if (turn == 0) red[j * 64 + lane] = dbias[j];
                        else red[j * 64 + lane] += dbias[j];
                    }
                }
            }
            __syncthreads();
        }
        float* dst = dbias_partial + (int64_t)(gw * g.heads + head) * 4096;
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) dst[i] = red[i];
    }
}

int attn_grid(int heads, int nwin) {
    int GW = (nwin + 3) / 4;
    int cap = (num_cus() * 3 + heads - 1) / heads;   // measured: 4 workgroups per CU is slower (fwd 39 vs 34 us, bwd 240 vs 189 us)
    if (cap < 1) cap = 1;
    if (GW > cap) GW = cap;
    return GW;
}

int check_geo(const char* name, Geo& g, int B, int Hp, int Wp, int heads, int head_dim, int ws, int shift, int dtype) {
    P4C_CHECK_ARG(B > 0 && Hp > 0 && Wp > 0 && heads > 0, "%s: bad sizes", name);
    P4C_CHECK_ARG(ws >= 3 && ws <= 8, "%s: window size %d not in 3..8 (N = ws*ws tokens must fit one 64-lane tile pair)", name, ws);
    P4C_CHECK_ARG(Hp % ws == 0 && Wp % ws == 0, "%s: token grid %dx%d is not a multiple of the window size %d (pad first, as Swin does)",
                  name, Hp, Wp, ws);
    P4C_CHECK_ARG(shift >= 0 && shift < ws, "%s: shift %d not in [0, ws)", name, shift);
    P4C_CHECK_ARG(head_dim == 8 || head_dim == 16 || head_dim == 32, "%s: head_dim %d not in {8,16,32}", name, head_dim);
    P4C_CHECK_ARG(dtype == P4C_F32 || dtype == P4C_BF16, "%s: dtype must be P4C_F32 or P4C_BF16", name);
    g = Geo{B, Hp, Wp, heads, ws, shift, Hp / ws, Wp / ws, ws * ws, heads * head_dim};
    return P4C_OK;
}

template <int D>
int launch_fwd_exact(const void* qkv, const float* bias_t, void* out, const Geo& g, float scale, hipStream_t stream) {
    constexpr int smem = exact_fwd_smem<D>();
    P4C_TRY(ensure_dyn_smem((const void*)window_attn_fwd_exact_kernel<D>, smem));
    const int GW = attn_grid(g.heads, g.B * g.nwy * g.nwx);
    hipLaunchKernelGGL((window_attn_fwd_exact_kernel<D>), dim3(((GW + 7) / 8) * 8 * g.heads), dim3(256), smem, stream, (const float*)qkv,
                       bias_t, (float*)out, g, scale, GW);
    P4C_CHECK_LAUNCH("window_attn_fwd_exact");
    return P4C_OK;
}

template <int D>
int launch_bwd_exact(const void* qkv, const float* bias_t, const void* dout, void* dqkv, float* partial, float* dbias_t, const Geo& g,
                     float scale, hipStream_t stream) {
    constexpr int smem = exact_bwd_smem<D>();
    P4C_TRY(ensure_dyn_smem((const void*)window_attn_bwd_exact_kernel<D>, smem));
    const int GW = attn_grid(g.heads, g.B * g.nwy * g.nwx);
    hipLaunchKernelGGL((window_attn_bwd_exact_kernel<D>), dim3(((GW + 7) / 8) * 8 * g.heads), dim3(256), smem, stream, (const float*)qkv,
                       bias_t, (const float*)dout, (float*)dqkv, partial, g, scale, GW);
    P4C_CHECK_LAUNCH("window_attn_bwd_exact");
    if (partial) {
        const int total = g.heads * g.N * g.N;
        hipLaunchKernelGGL(window_attn_dbias_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, stream, partial, dbias_t, g.heads,
                           g.N, GW);
        P4C_CHECK_LAUNCH("window_attn_dbias_reduce");
    }
    return P4C_OK;
}

template <typename T, int D>
int launch_fwd(const void* qkv, const float* bias_t, void* out, const Geo& g, float scale, hipStream_t stream) {
    const int smem = BIAS_BYTES + 4 * IMG;
    const int GW = attn_grid(g.heads, g.B * g.nwy * g.nwx);
    hipLaunchKernelGGL((window_attn_fwd_kernel<T, D>), dim3(((GW + 7) / 8) * 8 * g.heads), dim3(256), smem, stream, (const T*)qkv, bias_t, (T*)out, g,
                       scale, GW);
    P4C_CHECK_LAUNCH("window_attn_fwd");
    return P4C_OK;
}

template <typename T, int D>
int launch_bwd(const void* qkv, const float* bias_t, const void* dout, void* dqkv, float* partial, float* dbias_t, const Geo& g,
               float scale, hipStream_t stream) {
    const int smem = BIAS_BYTES + 4 * (3 * IMG + STAT_BYTES);
    P4C_TRY(ensure_dyn_smem((const void*)window_attn_bwd_kernel<T, D>, smem));
    const int GW = attn_grid(g.heads, g.B * g.nwy * g.nwx);
    hipLaunchKernelGGL((window_attn_bwd_kernel<T, D>), dim3(((GW + 7) / 8) * 8 * g.heads), dim3(256), smem, stream, (const T*)qkv, bias_t,
                       (const T*)dout, (T*)dqkv, partial, g, scale, GW);
    P4C_CHECK_LAUNCH("window_attn_bwd");
    if (partial) {
        const int total = g.heads * g.N * g.N;
        hipLaunchKernelGGL(window_attn_dbias_reduce_kernel, dim3((total + 31) / 32), dim3(256), 0, stream, partial, dbias_t, g.heads,
                           g.N, GW);
        P4C_CHECK_LAUNCH("window_attn_dbias_reduce");
    }
    return P4C_OK;
}

}  // namespace
}  // namespace p4c

using namespace p4c;

// fp32 activations run the fp32-exact kernels; P4C_ATTN_F32_MFMA=1 sends them through the bf16 matrix-core kernels instead (operands
// and P rounded to bf16: the round-1/2 behaviour, kept as an A/B switch)
static bool f32_on_matrix_cores() {
    const char* e = diag_env("P4C_ATTN_F32_MFMA");
    return e && e[0] == '1';
}

extern "C" size_t p4c_window_attn_bwd_workspace_bytes(int B, int Hp, int Wp, int heads, int ws) {
    if (B <= 0 || Hp <= 0 || Wp <= 0 || heads <= 0 || ws <= 0) return 0;
    const int nwin = B * (Hp / ws) * (Wp / ws);
    return (size_t)attn_grid(heads, nwin) * heads * 4096 * sizeof(float);
}

extern "C" int p4c_window_attn_fwd(const void* qkv, const float* bias_t, void* out, int B, int Hp, int Wp, int heads, int head_dim,
                                   int ws, int shift, float scale, int dtype, p4c_stream_t stream) {
    Geo g;
    int rc = check_geo("p4c_window_attn_fwd", g, B, Hp, Wp, heads, head_dim, ws, shift, dtype);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_ARG(qkv && out, "p4c_window_attn_fwd: NULL pointer");
    hipStream_t s = as_stream(stream);
#define P4C_ATTN_FWD(TT, DD) return launch_fwd<TT, DD>(qkv, bias_t, out, g, scale, s)
    if (dtype == P4C_F32) {
        if (!f32_on_matrix_cores()) {
            if (head_dim == 8) return launch_fwd_exact<8>(qkv, bias_t, out, g, scale, s);
            if (head_dim == 16) return launch_fwd_exact<16>(qkv, bias_t, out, g, scale, s);
            return launch_fwd_exact<32>(qkv, bias_t, out, g, scale, s);
        }
        if (head_dim == 8) P4C_ATTN_FWD(float, 8);
        if (head_dim == 16) P4C_ATTN_FWD(float, 16);
        P4C_ATTN_FWD(float, 32);
    }
    if (head_dim == 8) P4C_ATTN_FWD(bf16, 8);
    if (head_dim == 16) P4C_ATTN_FWD(bf16, 16);
    P4C_ATTN_FWD(bf16, 32);
#undef P4C_ATTN_FWD
}

extern "C" int p4c_window_attn_bwd(const void* qkv, const float* bias_t, const void* dout, void* dqkv, float* dbias_t, void* workspace,
                                   int B, int Hp, int Wp, int heads, int head_dim, int ws, int shift, float scale, int dtype,
                                   p4c_stream_t stream) {
    Geo g;
    int rc = check_geo("p4c_window_attn_bwd", g, B, Hp, Wp, heads, head_dim, ws, shift, dtype);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_ARG(qkv && dout && dqkv, "p4c_window_attn_bwd: NULL pointer");
    P4C_CHECK_ARG((dbias_t == nullptr) || (bias_t != nullptr && workspace != nullptr),
                  "p4c_window_attn_bwd: dbias_t needs bias_t and a workspace of p4c_window_attn_bwd_workspace_bytes()");
    float* partial = dbias_t ? reinterpret_cast<float*>(workspace) : nullptr;
    hipStream_t s = as_stream(stream);
#define P4C_ATTN_BWD(TT, DD) return launch_bwd<TT, DD>(qkv, bias_t, dout, dqkv, partial, dbias_t, g, scale, s)
    if (dtype == P4C_F32) {
        if (!f32_on_matrix_cores()) {
            if (head_dim == 8) return launch_bwd_exact<8>(qkv, bias_t, dout, dqkv, partial, dbias_t, g, scale, s);
            if (head_dim == 16) return launch_bwd_exact<16>(qkv, bias_t, dout, dqkv, partial, dbias_t, g, scale, s);
            return launch_bwd_exact<32>(qkv, bias_t, dout, dqkv, partial, dbias_t, g, scale, s);
        }
        if (head_dim == 8) P4C_ATTN_BWD(float, 8);
        if (head_dim == 16) P4C_ATTN_BWD(float, 16);
        P4C_ATTN_BWD(float, 32);
    }
    if (head_dim == 8) P4C_ATTN_BWD(bf16, 8);
    if (head_dim == 16) P4C_ATTN_BWD(bf16, 16);
    P4C_ATTN_BWD(bf16, 32);
#undef P4C_ATTN_BWD
}
