// bf16-matrix-core convolution kernels of the HalfUNet path (gfx950, v_mfma_f32_32x32x16_bf16):
// activations stay fp32 in HBM; they are rounded to bf16 (RNE) while the tile is staged into LDS,
// weights are rounded once per call by prep_weights_bf16, products accumulate in fp32.  This is the
// arithmetic torch.autocast(bfloat16) gives a conv (py4cast `trainer.precision: bf16`,
// lightning.py:479-493), with fp32 instead of bf16 activation storage.
//
//   conv_fwd_bf16   : persistent workgroups; the 9*CI*64 bf16 weights live in LDS for the whole launch
//                     (74 KB at CI=64), pixel tiles (8x32, or 4x32 at CI=96) are double buffered through
//                     registers: the next tile's global loads are in flight during the current tile's MFMAs.
//                     Same fused input normalisation+ReLU and statistics epilogue as conv_fwd_f32; also
//                     evaluates the data gradient (flipped/transposed weights).
//   conv_wgrad_bf16 : persistent, K = pixels.  The MFMA operands need 8 consecutive PIXELS of one channel per
//                     lane, i.e. the transpose of the NHWC tile: read straight from the [pixel][channel] LDS
//                     image with ds_read_b64_tr_b16 (gfx950 transposing LDS read), so tap shifts are plain row
//                     offsets and no transposed copy is staged.
//
// At 16x the fp32 MFMA rate these kernels are HBM-bound: 2 * 4 B * 64 ch per pixel (read + write).
#include <stdlib.h>

#include <type_traits>

#include "kernels.hpp"

namespace p4c {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BTW = 32;  // tile width in pixels

#ifndef P4C_PRIO
#define P4C_PRIO 0  // diagnostic builds only: static wave priority of the memory-side (1, 3) or matrix (2) waves
#endif
#ifndef P4C_EXP
#define P4C_EXP 0  // diagnostic builds only (tools/diagnostics/exp_build.sh): bit mask of pipeline stages to leave out
#endif

#ifdef P4C_STAMPS  // diagnostic build only: per-iteration s_memtime stamps of one compute and one loader wave
__device__ unsigned long long* g_stamps = nullptr;
__device__ __forceinline__ void stamp(int slot) {
    if (g_stamps && blockIdx.x == 7 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        g_stamps[slot] = t;
    }
}
#define P4C_STAMP(slot) stamp(slot)
__device__ __forceinline__ void stamp_rt(int slot) {   // constant 100 MHz counter: in-kernel clock = d(memtime)/d(realtime)*100 MHz
    if (g_stamps && blockIdx.x == 7 && blockIdx.y == 0 && (threadIdx.x & 63) == 0) g_stamps[slot] = __builtin_amdgcn_s_memrealtime();
}
#define P4C_STAMP_RT(slot) stamp_rt(slot)
#else
#define P4C_STAMP_RT(slot)
#define P4C_STAMP(slot)
#endif

// ---------------------------------------------------------------------------------------------
// prep_weights_bf16: canonical w[CO][CI][ks][ks] fp32 -> bf16 MFMA A-operand stream
//   out[mb][tap][kstep][h][m_l(64)][j(8)],  k = 16 kstep + 8 h + j,  m = 64 mb + m_l   (zero padded)
//   transpose_flip as in prep_weights (conv_f32.hip).
__global__ void prep_weights_bf16_kernel(const float* __restrict__ w, int CO, int CI, int ntaps, int transpose_flip,
                                         int M_pad, int K_pad, __bf16* __restrict__ out) {
    const int total = M_pad * K_pad * ntaps;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int t = i;
        const int j = t & 7; t >>= 3;
        const int m_l = t & 63; t >>= 6;
        const int h = t & 1; t >>= 1;
        const int ks = t % (K_pad / 16); t /= (K_pad / 16);
        const int tap = t % ntaps;
        const int mb = t / ntaps;
        const int k = 16 * ks + 8 * h + j, m = 64 * mb + m_l;
        float v = 0.0f;
        if (!transpose_flip) {
            if (m < CO && k < CI) v = w[((int64_t)m * CI + k) * ntaps + tap];
        } else {
            if (k < CO && m < CI) v = w[((int64_t)k * CI + m) * ntaps + (ntaps - 1 - tap)];
        }
        out[i] = (__bf16)v;
    }
}

__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    bf16x8 r;
    r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
    r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
    return r;
}

// ---------------------------------------------------------------------------------------------
// 8 channels of one pixel in registers, as loaded from an fp32 (32 B) or bf16 (16 B) NHWC tensor.
template <typename T>
struct Slot8;
template <>
struct Slot8<float> {
    f32x4 lo, hi;
    __device__ __forceinline__ void zero() { lo = f32x4{0.f, 0.f, 0.f, 0.f}; hi = lo; }
    __device__ __forceinline__ void load(const float* p) {
        lo = *reinterpret_cast<const f32x4*>(p);
        hi = *reinterpret_cast<const f32x4*>(p + 4);
    }
    __device__ __forceinline__ void get(f32x4& a, f32x4& b) const { a = lo; b = hi; }
    __device__ __forceinline__ bf16x8 raw() const { return pack8(lo, hi); }
};
template <>
struct Slot8<__bf16> {
    bf16x8 v;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f;
    }
    __device__ __forceinline__ void load(const __bf16* p) { v = *reinterpret_cast<const bf16x8*>(p); }
    __device__ __forceinline__ void get(f32x4& a, f32x4& b) const {
        a = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        b = f32x4{(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
    }
    __device__ __forceinline__ bf16x8 raw() const { return v; }
};

// Register image of a (LH x LW) halo tile of an NHWC tensor, 8 channels per slot.
template <typename T, int CI, int LH, int LW>
struct BTile {
    static constexpr int C8 = CI / 8;
    static constexpr int TOTAL = LH * LW * C8;
    static constexpr int ITERS = (TOTAL + 255) / 256;
    Slot8<T> s[ITERS];
};

template <typename T, int CI, int LH, int LW, int HALO>
__device__ __forceinline__ void btile_load(BTile<T, CI, LH, LW>& t, const T* __restrict__ in, int b, int y0, int x0, int H,
                                           int W, int in_cs, int tid = threadIdx.x) {
    constexpr int C8 = CI / 8;
#pragma unroll
    for (int it = 0; it < BTile<T, CI, LH, LW>::ITERS; ++it) {
        const int idx = tid + it * 256;
        const int pix = idx / C8, c8 = idx - pix * C8;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        t.s[it].zero();
        if (idx < BTile<T, CI, LH, LW>::TOTAL && gy >= 0 && gy < H && gx >= 0 && gx < W)
            t.s[it].load(in + (((int64_t)b * H + gy) * W + gx) * in_cs + 8 * c8);
    }
}

// transform (norm + relu of the producer layer), round to bf16, write [pixel][channel] rows of ROWB bytes.
// MODE 0: plain copy, 1: ReLU, 2: scale/shift (+ReLU if `relu`).  Branch-free per slot (out-of-image slots are
// selected to zero AFTER the transform: zero padding applies to the normalised activation), so the unrolled
// iterations stay one straight-line stream.
template <typename T, int CI, int LH, int LW, int HALO, int ROWB, int MODE>
__device__ __forceinline__ void btile_store_mode(const BTile<T, CI, LH, LW>& t, const float* __restrict__ scale,
                                                 const float* __restrict__ shift, int relu, char* lds, int b, int y0,
                                                 int x0, int H, int W, int sc_cs, int tid) {
    constexpr int C8 = CI / 8;
    constexpr bool FIXED = (256 % C8) == 0;  // the thread's channel octet is the same in every iteration
    f32x4 sc_lo = {1, 1, 1, 1}, sc_hi = {1, 1, 1, 1}, sh_lo = {0, 0, 0, 0}, sh_hi = {0, 0, 0, 0};
    if (MODE == 2 && FIXED) {
        const int c8 = tid % C8;
        const float* sc = scale + (int64_t)b * sc_cs + 8 * c8;
        const float* sh = shift + (int64_t)b * sc_cs + 8 * c8;
        sc_lo = *reinterpret_cast<const f32x4*>(sc); sc_hi = *reinterpret_cast<const f32x4*>(sc + 4);
        sh_lo = *reinterpret_cast<const f32x4*>(sh); sh_hi = *reinterpret_cast<const f32x4*>(sh + 4);
    }
    const float lo_clamp = (MODE == 1 || relu) ? 0.f : -3.402823466e38f;
#pragma unroll
    for (int it = 0; it < BTile<T, CI, LH, LW>::ITERS; ++it) {
        const int idx = tid + it * 256;
        if (idx >= BTile<T, CI, LH, LW>::TOTAL) break;
        const int pix = idx / C8, c8 = idx - pix * C8;
        bf16x8 o;
        if (MODE == 0) {
            o = t.s[it].raw();
        } else {
            const int ly = pix / LW, lx = pix - ly * LW;
            const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
            const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
            f32x4 lo, hi;
            t.s[it].get(lo, hi);
            if (MODE == 2) {
                if (!FIXED) {
                    const float* sc = scale + (int64_t)b * sc_cs + 8 * c8;
                    const float* sh = shift + (int64_t)b * sc_cs + 8 * c8;
                    sc_lo = *reinterpret_cast<const f32x4*>(sc); sc_hi = *reinterpret_cast<const f32x4*>(sc + 4);
                    sh_lo = *reinterpret_cast<const f32x4*>(sh); sh_hi = *reinterpret_cast<const f32x4*>(sh + 4);
                }
                lo = lo * sc_lo + sh_lo;
                hi = hi * sc_hi + sh_hi;
            }
            lo.x = fmaxf(lo.x, lo_clamp); lo.y = fmaxf(lo.y, lo_clamp); lo.z = fmaxf(lo.z, lo_clamp); lo.w = fmaxf(lo.w, lo_clamp);
            hi.x = fmaxf(hi.x, lo_clamp); hi.y = fmaxf(hi.y, lo_clamp); hi.z = fmaxf(hi.z, lo_clamp); hi.w = fmaxf(hi.w, lo_clamp);
            const float keep = inb ? 1.f : 0.f;
            o = pack8(lo * keep, hi * keep);
        }
        *reinterpret_cast<bf16x8*>(lds + pix * ROWB + 16 * c8) = o;
    }
}

template <typename T, int CI, int LH, int LW, int HALO, int ROWB>
__device__ __forceinline__ void btile_store(const BTile<T, CI, LH, LW>& t, const float* __restrict__ scale,
                                            const float* __restrict__ shift, int relu, char* lds, int b, int y0, int x0,
                                            int H, int W, int sc_cs, int tid = threadIdx.x) {
    if (scale)
        btile_store_mode<T, CI, LH, LW, HALO, ROWB, 2>(t, scale, shift, relu, lds, b, y0, x0, H, W, sc_cs, tid);
    else if (relu)
        btile_store_mode<T, CI, LH, LW, HALO, ROWB, 1>(t, scale, shift, relu, lds, b, y0, x0, H, W, sc_cs, tid);
    else
        btile_store_mode<T, CI, LH, LW, HALO, ROWB, 0>(t, scale, shift, relu, lds, b, y0, x0, H, W, sc_cs, tid);
}

// 4 consecutive output channels of one pixel
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(__bf16* p, f32x4 v) {
    bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(p) = o;
}

// ---------------------------------------------------------------------------------------------
// conv_fwd_bf16: persistent, grid = G workgroups x M_pad/64 (blockIdx.y), 256 threads.
// Tile 4 x 32 pixels x 64 output channels; wave w owns tile row w (one 32-pixel MFMA N-tile, two channel tiles).
// LDS: weights [tap][kstep][h][64][8] bf16 for the whole launch, NBUF halo-tile buffers with rows padded by one
// 16-byte slot (row stride = odd number of slots -> the 16 lanes of a ds_read_b128 group hit 16 distinct
// slots), and a small statistics scratch.
// NBUF = 2 (CI <= 64): software pipeline over tiles, ONE barrier per tile --
//     MFMA(tile i from buf[i&1]) ; write registers(tile i+1) -> buf[~i&1] ; issue global loads(tile i+2) ;
//     epilogue(tile i: stores + statistics) ; barrier
//   so global-load latency spans a whole iteration and LDS writes overlap other waves' MFMAs.
// NBUF = 1 (CI = 96, LDS-limited): load -> barrier -> MFMA with the next tile's loads in flight.
template <typename T, int CI, int KS, int NBUF>
__global__ void __launch_bounds__(256, 1)
    conv_fwd_bf16_kernel(const T* __restrict__ in, const __bf16* __restrict__ wp, const float* __restrict__ in_scale,
                         const float* __restrict__ in_shift, int in_relu, T* __restrict__ out, int out_cs,
                         float* __restrict__ stat_partial, int B, int H, int W) {
    constexpr int TH = 4;
    constexpr int HALO = KS / 2;
    constexpr int LH = TH + 2 * HALO, LW = BTW + 2 * HALO;
    constexpr int ROWB = CI * 2 + 16;
    constexpr int NTAPS = KS * KS;
    constexpr int NKS = CI / 16;
    constexpr int WBYTES = NTAPS * CI * 64 * 2;
    constexpr int TILEB = (LH * LW * ROWB + 15) / 16 * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lw = smem;
    char* lt = smem + WBYTES;
    // statistics scratch [4 waves][32][33] + 2 x [4][2][64]; with one tile buffer (LDS-limited) it aliases the tile
    float* sscr = reinterpret_cast<float*>(NBUF == 2 ? smem + WBYTES + NBUF * TILEB : lt);

    const int mb = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * B;

    {   // weights -> LDS once per workgroup (linear copy, 16 B per thread and step)
        const char* src = reinterpret_cast<const char*>(wp) + (int64_t)mb * WBYTES;
        for (int i = threadIdx.x * 16; i < WBYTES; i += 256 * 16)
            *reinterpret_cast<f32x4*>(lw + i) = *reinterpret_cast<const f32x4*>(src + i);
    }

    // each workgroup walks a contiguous run of tiles ordered down a 32-pixel-wide strip (ty fastest): the halo rows
    // shared with the previous tile were fetched by this same CU a moment ago (L2 hits, not HBM re-reads)
    const int t_begin = (int)((int64_t)ntiles * blockIdx.x / gridDim.x);
    const int t_end = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
    auto coords = [&](int t, int& b, int& y0, int& x0, int& canon) {
        const int ty = t % tiles_y;
        const int rest = t / tiles_y;
        const int tx = rest % tiles_x;
        b = rest / tiles_x;
        y0 = ty * TH;
        x0 = tx * BTW;
        canon = (b * tiles_y + ty) * tiles_x + tx;
    };
    BTile<T, CI, LH, LW> tr;
    auto load = [&](int t) {
        int b, y0, x0, cn;
        coords(t, b, y0, x0, cn);
        btile_load<T, CI, LH, LW, HALO>(tr, in, b, y0, x0, H, W, CI);
    };
    auto store = [&](int t, char* buf) {
        int b, y0, x0, cn;
        coords(t, b, y0, x0, cn);
        btile_store<T, CI, LH, LW, HALO, ROWB>(tr, in_scale, in_shift, in_relu, buf, b, y0, x0, H, W, CI);
    };

    if (t_begin >= t_end) return;
    if (NBUF == 2) {
        load(t_begin);
        store(t_begin, lt);
        if (t_begin + 1 < t_end) load(t_begin + 1);
    } else {
        load(t_begin);
    }
    __syncthreads();
    // staged epilogue (bf16 storage, one tile buffer): running statistics of this lane's 8 channels
    constexpr bool STAGED = NBUF == 1 && std::is_same<T, __bf16>::value;
    float st1[8], st2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) st1[q] = st2[q] = 0.f;
    int st_b = -1;
    auto st_flush = [&](int bb) __attribute__((always_inline)) {
        float* dst = stat_partial + ((int64_t)bb * gridDim.x * 4 + blockIdx.x * 4 + wv) * 128;
        const int c8 = threadIdx.x & 7;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float u = st1[q], v = st2[q];
            u += __shfl_xor(u, 8); v += __shfl_xor(v, 8);
            u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
            u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
            if (lane < 8) { dst[8 * c8 + q] = u; dst[64 + 8 * c8 + q] = v; }
            st1[q] = st2[q] = 0.f;
        }
    };

    for (int tile = t_begin; tile < t_end; ++tile) {
        int b, y0, x0, canon;
        coords(tile, b, y0, x0, canon);
        char* cur = lt;
        if (NBUF == 2) {
            cur = lt + ((tile - t_begin) & 1) * TILEB;
        } else {
            if (!(P4C_EXP & 1) || tile == t_begin) store(tile, lt);
            __syncthreads();
            if (tile + 1 < t_end && !(P4C_EXP & 2)) load(tile + 1);
        }

        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
        const char* wl = lw + (h * 64 + r) * 16;
        {
            // tap stream with operand double buffering at tap granularity: the 3*NKS LDS reads of tap t+1 are issued
            // before the 2*NKS MFMAs of tap t (see conv_fwd_bf16_ws_kernel)
            const char* pl0 = cur + (wv * LW + r) * ROWB + 16 * h;
            bf16x8 fa0[2][NKS], fa1[2][NKS], fb[2][NKS];
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                fa0[0][ks] = *reinterpret_cast<const bf16x8*>(wl + ks * 2048);
                fa1[0][ks] = *reinterpret_cast<const bf16x8*>(wl + ks * 2048 + 512);
                fb[0][ks] = *reinterpret_cast<const bf16x8*>(pl0 + 32 * ks);
            }
#pragma unroll
            for (int tap = 0; tap < ((P4C_EXP & 16) ? 1 : NTAPS); ++tap) {
                const int cb = tap & 1, nb = cb ^ 1;
                if (tap + 1 < NTAPS) {
                    const int ky = (tap + 1) / KS, kx = (tap + 1) % KS;
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) {
                        fa0[nb][ks] = *reinterpret_cast<const bf16x8*>(wl + ((tap + 1) * NKS + ks) * 2048);
                        fa1[nb][ks] = *reinterpret_cast<const bf16x8*>(wl + ((tap + 1) * NKS + ks) * 2048 + 512);
                        fb[nb][ks] = *reinterpret_cast<const bf16x8*>(pl0 + (ky * LW + kx) * ROWB + 32 * ks);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[cb][ks], fb[cb][ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[cb][ks], fb[cb][ks], acc1, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        if (NBUF == 2) {
            // the other buffer was last read by MFMA(tile-1), which every wave finished before the previous barrier
            if (tile + 1 < t_end) store(tile + 1, lt + (((tile - t_begin) & 1) ^ 1) * TILEB);
            if (tile + 2 < t_end) load(tile + 2);
        }

        // ---- epilogue: C[co][px]; lane = pixel r (+ half h), register i -> co = (i&3) + 8*(i>>2) + 4*h
        if constexpr (STAGED) {
            // bf16 storage, single tile buffer: the accumulators are rounded into an LDS staging tile (it aliases the
            // input tile, which every wave has finished reading), then ALL threads store it with 16-byte fully
            // coalesced stores and accumulate the statistics of those rounded values in registers (8 fixed channels per
            // lane, flushed per sample).  Replaces 8 scattered 8-byte stores + two LDS transposes per lane and tile.
            __syncthreads();
            char* stg = lt + (wv * 32 + r) * 128 + 8 * h;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& a = ct == 0 ? acc0 : acc1;
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
                    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
                    const f32x2_ lo = {a[4 * g], a[4 * g + 1]}, hi = {a[4 * g + 2], a[4 * g + 3]};
                    u32x2_ o;
                    o[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(lo, bf16x2_));
                    o[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(hi, bf16x2_));
                    *reinterpret_cast<u32x2_*>(stg + (((4 * ct + g) ^ (r & 7)) << 4)) = o;
                }
            __syncthreads();
            if (stat_partial && b != st_b) {
                if (st_b >= 0) st_flush(st_b);
                st_b = b;
            }
            const int c8 = threadIdx.x & 7;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int px = (threadIdx.x + k * 256) >> 3;
                typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
                u32x4_ v = *reinterpret_cast<const u32x4_*>(lt + px * 128 + ((c8 ^ (px & 7)) << 4));
                const int oy = y0 + (px >> 5), ox = x0 + (px & 31);
                const bool ok = (oy < H) & (ox < W);
                if (ok && !(P4C_EXP & 4)) *reinterpret_cast<u32x4_*>(out + (((int64_t)b * H + oy) * W + ox) * out_cs + mb * 64 + 8 * c8) = v;
                if (stat_partial && !(P4C_EXP & 8)) {
                    const unsigned int keep = ok ? 0xffffffffu : 0u;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned int w = v[q] & keep;
                        const float lo = __builtin_bit_cast(float, w << 16), hi = __builtin_bit_cast(float, w & 0xffff0000u);
                        st1[2 * q] += lo; st2[2 * q] = __builtin_fmaf(lo, lo, st2[2 * q]);
                        st1[2 * q + 1] += hi; st2[2 * q + 1] = __builtin_fmaf(hi, hi, st2[2 * q + 1]);
                    }
                }
            }
            __syncthreads();   // the staging tile is overwritten by the next tile's input
            continue;
        }
        if (NBUF == 1 && stat_partial) __syncthreads();  // every wave is done reading the tile the scratch aliases
        const int gx = x0 + r, gy = y0 + wv;
        const bool valid = (gy < H) && (gx < W);
        T* orow = out + (((int64_t)b * H + gy) * W + gx) * out_cs + mb * 64 + 4 * h;
        float* tw = sscr + wv * (32 * 33);
        float a1s = 0.f, a2s = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& a = ct == 0 ? acc0 : acc1;
                const f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                if (valid) store4(orow + ct * 32 + 8 * g, v);
                if (stat_partial) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) tw[(8 * g + 4 * h + j) * 33 + r] = valid ? v[j] : 0.f;
                }
            }
            if (stat_partial) {
                // the wave wrote its own [32 co][32 px] block; lane l < 32 sums row l (stride 33: conflict-free)
                const float* row = tw + r * 33;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const float o = row[k];
                    s1 += o;
                    s2 += o * o;
                }
                if ((lane >> 5) == ct) { a1s = s1; a2s = s2; }  // lanes 0-31 keep ct=0, lanes 32-63 keep ct=1: co = lane
            }
        }
        if (stat_partial) {
            float* red = sscr + 4 * 32 * 33 + ((tile - t_begin) & 1) * 512;  // [parity][wave][stat][64]
            red[(wv * 2 + 0) * 64 + lane] = a1s;
            red[(wv * 2 + 1) * 64 + lane] = a2s;
        }
        __syncthreads();
        if (stat_partial && threadIdx.x < 128) {
            const float* red = sscr + 4 * 32 * 33 + ((tile - t_begin) & 1) * 512;
            const int t = threadIdx.x;
            stat_partial[(int64_t)canon * 128 + t] = (red[t] + red[128 + t]) + (red[256 + t] + red[384 + t]);
        }
    }
    if constexpr (STAGED) {
        if (stat_partial) {
            st_flush(st_b);
            int bf, y0, x0, cn;
            coords(t_begin, bf, y0, x0, cn);
            for (int bb = 0; bb < B; ++bb)   // samples this workgroup never touched read as zero (no memset pass)
                if (bb < bf || bb > st_b) {
                    float* dst = stat_partial + ((int64_t)bb * gridDim.x * 4 + blockIdx.x * 4 + wv) * 128;
                    dst[lane] = 0.f;
                    dst[64 + lane] = 0.f;
                }
        }
    }
}

// workgroup barrier that orders LDS traffic only: global loads (the loader's prefetch) and stores (the epilogue)
// stay in flight across it.  __syncthreads() would drain vmcnt(0) and expose a full memory latency per tile.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Per-wave flush of the running channel statistics: lane (r,h) holds, for register i of channel tile ct, the sum
// over ITS pixels of channel co = ct*32 + (i&3) + 8*(i>>2) + 4*h.  The 32 pixel-lanes are summed through a
// [32 co][33] LDS transpose (conflict-free) and lane l writes channel l of slot `dst` ([2][64] floats).
__device__ __forceinline__ void flush_stats(float (&s1)[2][16], float (&s2)[2][16], float* tw, float* dst, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float keep = 0.f;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int i = 0; i < 16; ++i) tw[((i & 3) + 8 * (i >> 2) + 4 * h) * 33 + r] = st == 0 ? s1[ct][i] : s2[ct][i];
            const float* row = tw + r * 33;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) s += row[k];
            if (h == ct) keep = s;  // lanes 0-31 -> channels 0-31, lanes 32-63 -> channels 32-63
        }
        dst[st * 64 + lane] = keep;
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) s1[ct][i] = s2[ct][i] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// conv_fwd_bf16_ws: the tile pipeline with ROLE-SPLIT waves (512 threads): waves 0-3 only run the matrix phase +
// epilogue, waves 4-7 only stage tiles (global -> registers -> normalise/ReLU/round -> LDS).  With one 4-wave
// workgroup per CU every staging instruction would sit in the MFMA waves' issue stream; with the split the two
// waves of each SIMD run a matrix-heavy and a memory-heavy stream side by side.  One LDS-only barrier per tile.
//   iteration i:  compute waves: MFMA(buf[i&1]) ; epilogue(i)
//                 loader waves : write regs(tile i+1) -> buf[~i&1] ; issue loads(tile i+2)
// Channel statistics are accumulated in registers over all tiles of a sample and flushed once per
// (sample, wave): stat_partial[b][G*4][2][64], zero-filled by the launcher.
template <typename T, int CI, int KS>
__global__ void __launch_bounds__(512, 2)
    conv_fwd_bf16_ws_kernel(const T* __restrict__ in, const __bf16* __restrict__ wp, const float* __restrict__ in_scale,
                            const float* __restrict__ in_shift, int in_relu, T* __restrict__ out, int out_cs,
                            float* __restrict__ stat_partial, int B, int H, int W) {
    constexpr int TH = 4;
    constexpr int HALO = KS / 2;
    constexpr int LH = TH + 2 * HALO, LW = BTW + 2 * HALO;
    constexpr int ROWB = CI * 2 + 16;
    constexpr int NTAPS = KS * KS;
    constexpr int NKS = CI / 16;
    constexpr int WBYTES = NTAPS * CI * 64 * 2;
    constexpr int TILEB = (LH * LW * ROWB + 15) / 16 * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lw = smem;
    char* lt = smem + WBYTES;
    float* sscr = reinterpret_cast<float*>(smem + WBYTES + 2 * TILEB);  // [4 waves][32][33] statistics transpose
    // 1x1 convs on bf16 storage without statistics (the output conv and its data gradient): the outputs go through two
    // LDS staging tiles and the loader waves write them with 16-byte coalesced stores (see conv3x3_bf16_ring_kernel)
    constexpr bool CAN_STAGE = KS == 1 && std::is_same<T, __bf16>::value;
    const bool staged = CAN_STAGE && stat_partial == nullptr && (out_cs & 7) == 0;
    char* lstg = smem + WBYTES + 2 * TILEB + 4 * 32 * 33 * (int)sizeof(float);

    const int mb = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool loader = wv >= 4;
    const int ltid = threadIdx.x - 256;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * B;

    const int t_begin = (int)((int64_t)ntiles * blockIdx.x / gridDim.x);
    const int t_end = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
    auto coords = [&](int t, int& b, int& y0, int& x0) {
        const int ty = t % tiles_y;
        const int rest = t / tiles_y;
        const int tx = rest % tiles_x;
        b = rest / tiles_x;
        y0 = ty * TH;
        x0 = tx * BTW;
    };
    if (t_begin >= t_end) return;

    const char* wsrc = reinterpret_cast<const char*>(wp) + (int64_t)mb * WBYTES;
    constexpr int WIT = (WBYTES + 512 * 16 - 1) / (512 * 16);
    // weights -> LDS once per workgroup: all loads of a thread are issued before its stores
    auto copy_weights = [&]() {
        f32x4 wr[WIT];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = (threadIdx.x + it * 512) * 16;
            if (i < WBYTES) wr[it] = *reinterpret_cast<const f32x4*>(wsrc + i);
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = (threadIdx.x + it * 512) * 16;
            if (i < WBYTES) *reinterpret_cast<f32x4*>(lw + i) = wr[it];
        }
    };

    // The two roles run SEPARATE loops (same number of barriers) so that their register sets do not add up.
    if (loader) {
        // two register images (A, B) alternate: while tile i+1 is transformed and written to LDS from one of them,
        // the loads of tiles i+2 and i+3 are already in flight in the other / re-issued into the freed one, so
        // about two tiles of global loads are outstanding per CU at any time (memory-level parallelism).
        BTile<T, CI, LH, LW> ta, tb;
        auto load = [&](BTile<T, CI, LH, LW>& tr, int t) {
            if (t >= t_end) return;
            if (P4C_EXP & 2) { if (t > t_begin + 1) return; }
            int b, y0, x0;
            coords(t, b, y0, x0);
            btile_load<T, CI, LH, LW, HALO>(tr, in, b, y0, x0, H, W, CI, ltid);
        };
        auto store = [&](const BTile<T, CI, LH, LW>& tr, int t, char* buf) {
            if (t >= t_end) return;
            if (P4C_EXP & 1) { if (t > t_begin + 1) return; }
            int b, y0, x0;
            coords(t, b, y0, x0);
            btile_store<T, CI, LH, LW, HALO, ROWB>(tr, in_scale, in_shift, in_relu, buf, b, y0, x0, H, W, CI, ltid);
        };
        auto drain = [&](int t) __attribute__((always_inline)) {   // staged outputs of tile t -> HBM
            if (!staged || t < t_begin) return;
            int b, y0, x0;
            coords(t, b, y0, x0);
            const char* stg = lstg + ((t - t_begin) & 1) * (TH * BTW * 128);
            const int c8 = ltid & 7;
            typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int px = (ltid + k * 256) >> 3;
                const u32x4_ v = *reinterpret_cast<const u32x4_*>(stg + px * 128 + ((c8 ^ (px & 7)) << 4));
                const int oy = y0 + (px >> 5), ox = x0 + (px & 31);
                if ((oy < H) & (ox < W))
                    *reinterpret_cast<u32x4_*>(reinterpret_cast<__bf16*>(out) + (((int64_t)b * H + oy) * W + ox) * out_cs + mb * 64 + 8 * c8) = v;
            }
        };
        load(ta, t_begin);      // first tiles' loads fly while the weights are copied
        load(tb, t_begin + 1);
        copy_weights();
        store(ta, t_begin, lt);
        load(ta, t_begin + 2);
        lds_barrier();
        // iteration `tile`: stage tile+1 into the buffer the compute waves are NOT reading
        for (int tile = t_begin; tile < t_end; tile += 2) {
            P4C_STAMP(1000 + 4 * (tile - t_begin) + 0);
            drain(tile - 1);
            store(tb, tile + 1, lt + TILEB);   // (tile - t_begin) even -> compute reads buffer 0
            P4C_STAMP(1000 + 4 * (tile - t_begin) + 1);
            load(tb, tile + 3);
            P4C_STAMP(1000 + 4 * (tile - t_begin) + 2);
            lds_barrier();
            P4C_STAMP(1000 + 4 * (tile - t_begin) + 3);
            if (tile + 1 >= t_end) break;
            P4C_STAMP(1000 + 4 * (tile + 1 - t_begin) + 0);
            drain(tile);
            store(ta, tile + 2, lt);           // compute reads buffer 1
            P4C_STAMP(1000 + 4 * (tile + 1 - t_begin) + 1);
            load(ta, tile + 4);
            P4C_STAMP(1000 + 4 * (tile + 1 - t_begin) + 2);
            lds_barrier();
            P4C_STAMP(1000 + 4 * (tile + 1 - t_begin) + 3);
        }
        drain(t_end - 1);
        return;
    }

    copy_weights();
    lds_barrier();
    float s1[2][16], s2[2][16];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int i = 0; i < 16; ++i) s1[ct][i] = s2[ct][i] = 0.f;
    int cur_b = -1;
    float* tw = sscr + wv * (32 * 33);
    const int nslot = gridDim.x * 4;
    const char* wl = lw + (h * 64 + r) * 16;

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int par = (tile - t_begin) & 1;
        int b, y0, x0;
        coords(tile, b, y0, x0);
        if (stat_partial && b != cur_b) {
            if (cur_b >= 0) flush_stats(s1, s2, tw, stat_partial + ((int64_t)cur_b * nslot + blockIdx.x * 4 + wv) * 128, lane);
            cur_b = b;
        }
        P4C_STAMP(4 * (tile - t_begin) + 0);
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
        const char* pl = lt + par * TILEB + (wv * LW + r) * ROWB + 16 * h;
        if (!(P4C_EXP & 16)) {
        // fully unrolled tap stream with operand double buffering at TAP granularity: the 3*NKS ds_read_b128 of tap
        // t+1 are issued before the 2*NKS MFMAs of tap t (>= 256 cycles of matrix work cover the LDS read latency;
        // a one-step look-ahead of 64 cycles does not: measured 65 cycles per MFMA instead of 32)
        bf16x8 fa0[2][NKS], fa1[2][NKS], fb[2][NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            fa0[0][ks] = *reinterpret_cast<const bf16x8*>(wl + ks * 2048);
            fa1[0][ks] = *reinterpret_cast<const bf16x8*>(wl + ks * 2048 + 512);
            fb[0][ks] = *reinterpret_cast<const bf16x8*>(pl + 32 * ks);
        }
#pragma unroll
        for (int tap = 0; tap < NTAPS; ++tap) {
            const int cb = tap & 1, nb = cb ^ 1;
            if (tap + 1 < NTAPS) {
                const int ky = (tap + 1) / KS, kx = (tap + 1) % KS;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    fa0[nb][ks] = *reinterpret_cast<const bf16x8*>(wl + ((tap + 1) * NKS + ks) * 2048);
                    fa1[nb][ks] = *reinterpret_cast<const bf16x8*>(wl + ((tap + 1) * NKS + ks) * 2048 + 512);
                    fb[nb][ks] = *reinterpret_cast<const bf16x8*>(pl + (ky * LW + kx) * ROWB + 32 * ks);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the look-ahead reads ABOVE this tap's MFMAs (hipcc sinks them otherwise)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa0[cb][ks], fb[cb][ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa1[cb][ks], fb[cb][ks], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        } else { acc0[0] = pl[0]; acc1[3] = pl[16]; }
        P4C_STAMP(4 * (tile - t_begin) + 1);
        // ---- epilogue: C[co][px]; lane = pixel r (+ half h), register i -> co = (i&3) + 8*(i>>2) + 4*h
        const int gx = x0 + r, gy = y0 + wv;
        const bool valid = (gy < H) && (gx < W);
        const float vkeep = valid ? 1.f : 0.f;
        T* orow = out + (((int64_t)b * H + gy) * W + gx) * out_cs + mb * 64 + 4 * h;
        if (CAN_STAGE && staged) {
            char* stg = lstg + par * (TH * BTW * 128) + (wv * 32 + r) * 128 + 8 * h;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x16& a = ct == 0 ? acc0 : acc1;
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
                    typedef unsigned int u32x2_ __attribute__((ext_vector_type(2)));
                    const f32x2_ lo = {a[4 * g], a[4 * g + 1]}, hi = {a[4 * g + 2], a[4 * g + 3]};
                    u32x2_ o;
                    o[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(lo, bf16x2_));
                    o[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(hi, bf16x2_));
                    *reinterpret_cast<u32x2_*>(stg + (((4 * ct + g) ^ (r & 7)) << 4)) = o;
                }
        } else
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x16& a = ct == 0 ? acc0 : acc1;
                const f32x4 v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
                if (valid && (!(P4C_EXP & 4) || v.x == 123.456f)) store4(orow + ct * 32 + 8 * g, v);
                if (stat_partial && !(P4C_EXP & 8)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float o = v[j] * vkeep;
                        s1[ct][4 * g + j] += o;
                        s2[ct][4 * g + j] += o * o;
                    }
                }
            }
        P4C_STAMP(4 * (tile - t_begin) + 2);
        lds_barrier();
        P4C_STAMP(4 * (tile - t_begin) + 3);
    }
    if (stat_partial) {
        flush_stats(s1, s2, tw, stat_partial + ((int64_t)cur_b * nslot + blockIdx.x * 4 + wv) * 128, lane);
        // samples this workgroup never touched: its slots must read as zero (no memset pass needed)
        int bf, y0, x0;
        coords(t_begin, bf, y0, x0);
        for (int b = 0; b < B; ++b)
            if (b < bf || b > cur_b) {
                float* dst = stat_partial + ((int64_t)b * nslot + blockIdx.x * 4 + wv) * 128;
                dst[lane] = 0.f;
                dst[64 + lane] = 0.f;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// conv3x3_bf16_ring: 3x3 conv, 64 -> 64 channels, bf16 activations in HBM (the hot kernel of the bf16 plan).
// 512 threads: waves 0-3 run the matrix phase, waves 4-7 the memory side (global -> registers -> normalise/ReLU ->
// LDS ring, and LDS staging -> HBM + channel statistics).  On a SIMD the scarce resource is VECTOR ISSUE: an MFMA
// holds it for 8 of its 32 cycles, every other vector instruction of EITHER wave for ~4, so the two roles overlap
// only while their non-MFMA instructions fit the 24 free cycles per MFMA (stage-removal runs, tools/diagnostics/conv_exp.py:
// the roles' times ADDED up).  The design therefore minimises instructions, not bytes:
//   * WEIGHTS STATIONARY IN REGISTERS: a compute wave owns 32 output channels x 2 tile rows and keeps its
//     9 taps x 64 input channels of weights (36 A operands, 144 VGPRs) for the whole launch; per MFMA it reads ONE
//     B operand from LDS (was 1.5 with LDS-resident weights, which saturated the LDS pipe at 97 %);
//   * input rows live in a CONTIGUOUS LDS ring per 32-pixel strip: walking down a strip a tile re-uses the two halo
//     rows its predecessor staged (4 rows fetched and normalised instead of 6); a tile's 6 rows are contiguous, so
//     every B read is `per-tile base VGPR + immediate`;
//   * rows are 128 bytes with the 16-byte slot index XOR-swizzled by the pixel column: conflict-free B reads and
//     loader writes without padding;
//   * the compute waves round their accumulators to bf16 into an LDS staging tile; the memory-side waves drain it
//     with 16-byte fully coalesced stores and accumulate the statistics of exactly those rounded values;
//   * the memory side is table driven (per-lane offsets computed once; buffer descriptors with a per-tile scalar
//     offset), with a slower bounds-checked path for tiles that touch the image border; both paths issue the same
//     number of memory operations, which keeps every s_waitcnt an exact count.
// LDS: ring 22 x 34 x 128 B = 94 KB | staging 2 x 16 KB | normalisation rows 4 KB.
namespace ring {
constexpr int TH = 4, LW = BTW + 2, NROW = 22, ROWB = 128, RROW = LW * ROWB;
constexpr int RINGB = NROW * RROW, STGB = TH * BTW * 128;
constexpr int MAXB = 32;                      // samples whose normalisation rows fit (512 B each)
constexpr int SMEM = RINGB + 2 * STGB + MAXB * 128 * 4;
constexpr int NIMG = 7;                       // register slots per lane of a staged tile (6 rows fresh, 4 otherwise)
constexpr int ROWSLOTS = LW * 8;              // 16-byte slots per halo row
}  // namespace ring

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int ring_swz(int col) { return (col >> 1) & 7; }

// relu(v*scale+shift) on the 2 bf16 channels packed in one word (fp32 arithmetic, one rounding), or parts of it.
// MODE 0: copy, 1: ReLU, 2: scale/shift + ReLU, 3: scale/shift.
template <int MODE>
__device__ __forceinline__ unsigned int xform2(unsigned int w, f32x2 sc, f32x2 sh) {
    if (MODE >= 2) {
        // scalar fma pair on purpose: beside MFMAs a v_pk_fma_f32 costs more issue time than two v_fma_f32
        const float lo = __builtin_fmaf(__builtin_bit_cast(float, w << 16), sc.x, sh.x);
        const float hi = __builtin_fmaf(__builtin_bit_cast(float, w & 0xffff0000u), sc.y, sh.y);
        const f32x2 v = {lo, hi};
        w = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
    }
    if (MODE == 1 || MODE == 2) {
        // bf16 ReLU on the packed pair: negative floats are negative int16 (sign bit), max(.,0) clears them
        const s16x2 z = {0, 0};
        w = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
    }
    return w;
}

// raw buffer descriptor over [base, base+bytes): loads beyond `bytes` return 0 and stores are dropped, which turns
// every image-border test of the memory side into an offset select (no divergent branches, no zero-fill code)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr int OOB = 0x7fffffff;

template <int MODE, bool BST = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
    conv3x3_bf16_ring_kernel(const __bf16* __restrict__ in, const __bf16* __restrict__ wp, const float* __restrict__ in_scale,
                             const float* __restrict__ in_shift, __bf16* __restrict__ out, int out_cs,
                             float* __restrict__ stat_partial, int B, int H, int W, BatchFin fin, RingBwdStats bst) {
    using namespace ring;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lring = smem;
    char* lstg = smem + RINGB;
    float* lnorm = reinterpret_cast<float*>(smem + RINGB + 2 * STGB);  // [B][2][64] scale, shift (MODE >= 2)

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool loader = wv >= 4;
    const int ltid = threadIdx.x - 256;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * B;
    const int t_begin = (int)((int64_t)ntiles * blockIdx.x / gridDim.x);
    const int t_end = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
    if (t_begin >= t_end) return;
    // Tile cursor: tiles run down a 32-pixel strip first (ty fastest), then across, then over samples; every role
    // walks its own cursors forward one tile at a time (no divisions in the loops).  `p` is the tile's first ring
    // row: a tile that continues down its strip and still fits shares its rows 0..1 with the predecessor's rows
    // 4..5 (p += 4); otherwise it is "fresh": all 6 rows are staged, behind the predecessor if that fits, else at the
    // start of the ring.  Either way the rows written for tile t+1 never overlap the 6 rows being read for tile t.
    struct Cur { int t, b, tx, ty, p; bool fresh; };
    auto cur_init = [&]() __attribute__((always_inline)) {
        Cur c;
        c.t = t_begin;
        c.ty = t_begin % tiles_y;
        const int rest = t_begin / tiles_y;
        c.tx = rest % tiles_x;
        c.b = rest / tiles_x;
        c.p = 0;
        c.fresh = true;
        return c;
    };
    auto cur_next = [&](Cur& c) __attribute__((always_inline)) {
        ++c.t;
        if (++c.ty == tiles_y) {
            c.ty = 0;
            if (++c.tx == tiles_x) { c.tx = 0; ++c.b; }
        }
        if (c.ty != 0 && c.p + 10 <= NROW) {
            c.p += 4;
            c.fresh = false;
        } else {
            c.fresh = true;
            c.p = (c.p + 12 <= NROW) ? c.p + 6 : 0;
        }
    };

    if (loader) {
        if (P4C_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (P4C_PRIO == 3) __builtin_amdgcn_s_setprio(3);
        const int c8 = ltid & 7;
        const unsigned int sample_bytes = (unsigned int)H * W * 64 * 2;
        const unsigned int out_sample_bytes = (unsigned int)H * W * out_cs * 2;
        // per-lane tables: slot `it` of a staged block (row-major over the block's rows, 272 slots per row)
        int gofs[NIMG], lofs[NIMG], rrcol[NIMG];
#pragma unroll
        for (int it = 0; it < NIMG; ++it) {
            const int idx = ltid + it * 256;
            const int rr = idx / ROWSLOTS, col = (idx - rr * ROWSLOTS) >> 3;
            gofs[it] = (rr * W + col) * 128 + 16 * c8;
            lofs[it] = rr * RROW + col * ROWB + ((c8 ^ ring_swz(col)) << 4);
            rrcol[it] = (rr << 8) | col;
        }
        int dofs[4], sofs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int px = (ltid + k * 256) >> 3;
            dofs[k] = (((px >> 5) * W + (px & 31)) * out_cs + 8 * c8) * 2;
            sofs[k] = px * 128 + ((c8 ^ ((px >> 1) & 7)) << 4);
        }
        struct Img { u32x4 s[NIMG]; };
        Img ta, tb;
        // A tile stages 6 halo rows when fresh, else 4 (its rows 2..5).  The instruction stream is the SAME either
        // way and on the fast and the border path (7 buffer loads per lane; slots that are not needed get an
        // out-of-range offset and cost no traffic): with a fixed number of memory operations per trip every
        // s_waitcnt is an exact count and the prefetches / output stores in flight are never drained by accident.
        Cur lc = cur_init();  // load cursor: the next tile a prefetch is issued for
        auto load = [&](Img& im) __attribute__((always_inline)) {
            int b = 0, y0 = 0, x0 = 0, nact = 0, j0 = 0;
            if (lc.t < t_end && !((P4C_EXP & 2) && lc.t > t_begin + 1)) {
                b = lc.b; y0 = lc.ty * TH; x0 = lc.tx * BTW;
                nact = (lc.fresh ? 6 : 4) * ROWSLOTS;
                j0 = lc.fresh ? 0 : 2;
            }
            cur_next(lc);
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(in + (int64_t)b * H * W * 64, sample_bytes);
            const int gy0 = y0 - 1 + j0, gx0 = x0 - 1;
            if (gy0 >= 0 && y0 + 5 <= H && gx0 >= 0 && x0 + 33 <= W) {
                const int so = (gy0 * W + gx0) * 128;   // interior: one scalar offset, per-lane table entries
#pragma unroll
                for (int it = 0; it < NIMG; ++it)
                    im.s[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (ltid + it * 256 < nact) ? gofs[it] : OOB, so, 0);
            } else {
#pragma unroll
                for (int it = 0; it < NIMG; ++it) {
                    const int gy = gy0 + (rrcol[it] >> 8), gx = gx0 + (rrcol[it] & 255);
                    const bool ok = (ltid + it * 256 < nact) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
                    im.s[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (gy * W + gx) * 128 + 16 * c8 : OOB, 0, 0);
                }
            }
        };
        // per-sample normalisation of this lane's 8 channels
        f32x2 sc[4], sh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sc[k] = sh[k] = f32x2{0.f, 0.f};
        int sc_b = -1;
        Cur sc_ = cur_init();  // store cursor: the next tile to stage into the ring
        auto store = [&](const Img& im) __attribute__((always_inline)) {
            const int t = sc_.t, b = sc_.b, y0 = sc_.ty * TH, x0 = sc_.tx * BTW, sp = sc_.p;
            const bool fresh = sc_.fresh;
            cur_next(sc_);
            if (t >= t_end || ((P4C_EXP & 1) && t > t_begin + 1)) return;
            if (MODE >= 2 && b != sc_b) {
                sc_b = b;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sc[k] = *reinterpret_cast<const f32x2*>(lnorm + b * 128 + 8 * c8 + 2 * k);
                    sh[k] = *reinterpret_cast<const f32x2*>(lnorm + b * 128 + 64 + 8 * c8 + 2 * k);
                }
            }
            const int j0 = fresh ? 0 : 2, nact = (fresh ? 6 : 4) * ROWSLOTS;
            char* dst = lring + (sp + j0) * RROW;
            const int gy0 = y0 - 1 + j0, gx0 = x0 - 1;
            const bool interior = gy0 >= 0 && y0 + 5 <= H && gx0 >= 0 && x0 + 33 <= W;
            if (MODE == 0 || interior) {
#pragma unroll
                for (int it = 0; it < NIMG; ++it) {
                    u32x4 o = im.s[it];
                    if (MODE != 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = xform2<MODE>(o[k], sc[k], sh[k]);
                    }
                    if (ltid + it * 256 < nact) *reinterpret_cast<u32x4*>(dst + lofs[it]) = o;
                }
            } else {
#pragma unroll
                for (int it = 0; it < NIMG; ++it) {
                    // zero padding applies to the NORMALISED activation: out-of-image slots are cleared after the transform
                    const int gy = gy0 + (rrcol[it] >> 8), gx = gx0 + (rrcol[it] & 255);
                    const unsigned int keep = (((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W)) ? 0xffffffffu : 0u;
                    u32x4 o = im.s[it];
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = xform2<MODE>(o[k], sc[k], sh[k]) & keep;
                    if (ltid + it * 256 < nact) *reinterpret_cast<u32x4*>(dst + lofs[it]) = o;
                }
            }
        };
        // channel statistics of this lane's 8 channels over the pixels it drains
        float a1[8], a2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a1[q] = a2[q] = 0.f;
        int cur_b = -1;
        const int nslot = gridDim.x * 4, lwv = wv - 4;
        auto flush = [&](int b) __attribute__((always_inline)) {
            float* dst = stat_partial + ((int64_t)b * nslot + blockIdx.x * 4 + lwv) * 128;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float u = a1[q], v = a2[q];
                u += __shfl_xor(u, 8); v += __shfl_xor(v, 8);
                u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
                u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
                if (BST) v *= bst.rstd[b * 64 + 8 * c8 + q];   // sums of g * (y - mean) -> sums of g * xhat
                if (lane < 8) { dst[8 * c8 + q] = u; dst[64 + 8 * c8 + q] = v; }
                a1[q] = a2[q] = 0.f;
            }
        };
        // BST: the y tile of the next tile to drain (prefetched one trip ahead) and the lane's normalisation constants
        u32x4 yq[4];
        f32x2 bsc[4], bsh[4], bmu[4];
        int yofs[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int px = (ltid + k * 256) >> 3;
            yofs[k] = (((px >> 5) * W + (px & 31)) * 64 + 8 * c8) * 2;
            yq[k] = u32x4{0u, 0u, 0u, 0u};
            bsc[k] = bsh[k] = bmu[k] = f32x2{0.f, 0.f};
        }
        auto yload = [&](const Cur& c) __attribute__((always_inline)) {
            const bool act = c.t < t_end;
            const int b = act ? c.b : 0, y0 = c.ty * TH, x0 = c.tx * BTW;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const __bf16*>(bst.y) + (int64_t)b * H * W * 64, sample_bytes);
            if (act && y0 + 4 <= H && x0 + 32 <= W) {
                const int so = (y0 * W + x0) * 128;
#pragma unroll
                for (int k = 0; k < 4; ++k) yq[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, yofs[k], so, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int px = (ltid + k * 256) >> 3;
                    const int gy = y0 + (px >> 5), gx = x0 + (px & 31);
                    const bool valid = act & (gy < H) & (gx < W);
                    yq[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, valid ? ((gy * W + gx) * 64 + 8 * c8) * 2 : OOB, 0, 0);
                }
            }
        };
        Cur dc = cur_init();  // drain cursor: the next tile whose staged output goes to HBM
        auto drain = [&](bool live) __attribute__((always_inline)) {
            const int b = dc.b, y0 = dc.ty * TH, x0 = dc.tx * BTW;
            const char* stg = lstg + ((dc.t - t_begin + (live ? 0 : 1)) & 1) * STGB;
            if (live) cur_next(dc);
            if (live && stat_partial && b != cur_b) {
                if (cur_b >= 0 && !fin.slots) flush(cur_b);   // (batch statistics: one sum over all samples, nothing to flush)
                cur_b = b;
                if (BST) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        bsc[k] = *reinterpret_cast<const f32x2*>(lnorm + b * 192 + 8 * c8 + 2 * k);
                        bsh[k] = *reinterpret_cast<const f32x2*>(lnorm + b * 192 + 64 + 8 * c8 + 2 * k);
                        bmu[k] = *reinterpret_cast<const f32x2*>(lnorm + b * 192 + 128 + 8 * c8 + 2 * k);
                    }
                }
            }
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(out + (int64_t)b * H * W * out_cs, out_sample_bytes);
            u32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const u32x4*>(stg + sofs[k]);
            if (live && y0 + 4 <= H && x0 + 32 <= W) {
                const int so = (y0 * W + x0) * out_cs * 2;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    __builtin_amdgcn_raw_buffer_store_b128(v[k], rs, (P4C_EXP & 4) ? OOB : dofs[k], so, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int px = (ltid + k * 256) >> 3;
                    const int gy = y0 + (px >> 5), gx = x0 + (px & 31);
                    const bool valid = (gy < H) & (gx < W) & live;
                    if (!valid) v[k] = u32x4{0u, 0u, 0u, 0u};   // keeps the statistics below unmasked
                    __builtin_amdgcn_raw_buffer_store_b128(v[k], rs, (valid && !(P4C_EXP & 4)) ? ((gy * W + gx) * out_cs + 8 * c8) * 2 : OOB, 0, 0);
                }
            }
            if (BST) {
                if (live) {
                    // pass 1 of the normalisation backward on the gradient tile just stored (its rounded values, what a separate
                    // pass would read back): g = dA where the forward ReLU was alive, sums of g and of g * (y - mean)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float ylo = __builtin_bit_cast(float, yq[k][q] << 16);
                            const float yhi = __builtin_bit_cast(float, yq[k][q] & 0xffff0000u);
                            float lo = __builtin_bit_cast(float, v[k][q] << 16);
                            float hi = __builtin_bit_cast(float, v[k][q] & 0xffff0000u);
                            lo = __builtin_fmaf(ylo, bsc[q].x, bsh[q].x) > 0.f ? lo : 0.f;
                            hi = __builtin_fmaf(yhi, bsc[q].y, bsh[q].y) > 0.f ? hi : 0.f;
                            a1[2 * q] += lo; a2[2 * q] = __builtin_fmaf(lo, ylo - bmu[q].x, a2[2 * q]);
                            a1[2 * q + 1] += hi; a2[2 * q + 1] = __builtin_fmaf(hi, yhi - bmu[q].y, a2[2 * q + 1]);
                        }
                    yload(dc);   // dc already points at the next tile to drain
                }
            } else if (stat_partial && !(P4C_EXP & 8)) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = __builtin_bit_cast(float, v[k][q] << 16);
                        const float hi = __builtin_bit_cast(float, v[k][q] & 0xffff0000u);
                        a1[2 * q] += lo; a2[2 * q] = __builtin_fmaf(lo, lo, a2[2 * q]);
                        a1[2 * q + 1] += hi; a2[2 * q + 1] = __builtin_fmaf(hi, hi, a2[2 * q + 1]);
                    }
            }
        };

        P4C_STAMP_RT(3100);
        load(ta);
        load(tb);
        if (MODE >= 2) {
            // normalisation rows of every sample -> LDS: the per-sample reload in store() is then an LDS read, which
            // does not tie the staging code to the global-memory counter (vmcnt) of the prefetches in flight
            for (int i = ltid; i < B * 64; i += 256) {
                const int b = i >> 6, c = i & 63;
                lnorm[b * 128 + c] = in_scale[i];
                lnorm[b * 128 + 64 + c] = in_shift[i];
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (BST) {   // (MODE 0 launches only: lnorm is free) scale, shift, mean of the gradient's layer, 192 floats per sample
            for (int i = ltid; i < B * 64; i += 256) {
                const int b = i >> 6, c = i & 63;
                lnorm[b * 192 + c] = bst.scale[i];
                lnorm[b * 192 + 64 + c] = bst.shift[i];
                lnorm[b * 192 + 128 + c] = bst.mean[i];
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            yload(dc);   // the first tile's y rows
        }
        store(ta);
        load(ta);
        P4C_STAMP_RT(3101);
        lds_barrier();
        P4C_STAMP_RT(3102);
        for (int tile = t_begin; tile < t_end; tile += 2) {
            // compute reads tile `tile`; drain tile-1, stage tile+1, prefetch tile+3.  The first trip has nothing to
            // drain but still issues the 4 (dropped, out-of-range) stores: same operation count on every trip.
            drain(tile > t_begin);
            store(tb);
            load(tb);
            lds_barrier();
            if (tile + 1 >= t_end) break;
            drain(true);
            store(ta);
            load(ta);
            lds_barrier();
        }
        P4C_STAMP_RT(3103);
        drain(true);
        P4C_STAMP_RT(3104);
        if (fin.slots) {
            // ---- BatchNorm finished in place (kernels.hpp: BatchFin).  The compute waves have left (their last barrier was the
            // one that released the final staged tile), so barriers from here on are among the four loader waves only.
            float* lred = lnorm;                                          // [4 waves][128], then doubles [8][128] -- all dead by now
            unsigned int* lflag = reinterpret_cast<unsigned int*>(lnorm + 4 * 128 + 8 * 256);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float u = a1[q], v = a2[q];
                u += __shfl_xor(u, 8); v += __shfl_xor(v, 8);
                u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
                u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
                if (lane < 8) { lred[lwv * 128 + 8 * c8 + q] = u; lred[lwv * 128 + 64 + 8 * c8 + q] = v; }
            }
            lds_barrier();
            if (lwv == 0) {
                // The slot goes out with device-scope (write-through) stores and the ticket follows once they are acknowledged: no
                // release FENCE -- at the end of a kernel that has just written its output map a fence means writing back every
                // dirty line of this XCD's L2 (measured: +10 us per launch, more than the finalize launch it was to replace).
#pragma unroll
                for (int j = lane; j < 128; j += 64)   // fixed order over the waves
                    __hip_atomic_store(fin.slots + (int64_t)blockIdx.x * 128 + j,
                                       (lred[j] + lred[128 + j]) + (lred[256 + j] + lred[384 + j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) *lflag = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            lds_barrier();
            if (*lflag != gridDim.x - 1) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // invalidate only: the other workgroups' slots are read from memory
            // 256 threads: 32 column quads x 8 slot groups; slots summed in increasing order within a group, groups in order
            const int cq = ltid & 31, sg = ltid >> 5;
            double acc4[4] = {0.0, 0.0, 0.0, 0.0};
            for (int s0 = sg; s0 < (int)gridDim.x; s0 += 8 * 8) {
                p4c_f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sl = s0 + 8 * u;
                    v[u] = sl < (int)gridDim.x ? *(reinterpret_cast<const p4c_f32x4*>(fin.slots + (int64_t)sl * 128) + cq)
                                               : p4c_f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc4[0] += v[u].x; acc4[1] += v[u].y; acc4[2] += v[u].z; acc4[3] += v[u].w; }
            }
            double* dred = reinterpret_cast<double*>(lnorm + 4 * 128);   // [8][128]
            lds_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) dred[sg * 128 + 4 * cq + k] = acc4[k];
            lds_barrier();
            if (ltid < 64) {
                const int c = ltid;
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int g8 = 0; g8 < 8; ++g8) { s1 += dred[g8 * 128 + c]; s2 += dred[g8 * 128 + 64 + c]; }
                const double n = fin.count;
                const double mean = s1 / n;
                double var = s2 / n - mean * mean;
                if (var < 0.0) var = 0.0;
                const float rstd = (float)(1.0 / sqrt(var + (double)fin.eps));
                if (fin.running_mean) {   // torch semantics: biased variance normalises, the unbiased one is tracked
                    fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)mean;
                    const double unbiased = n > 1.0 ? var * n / (n - 1.0) : var;
                    fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)unbiased;
                }
                const float sc = fin.gamma[c] * rstd, sh = fin.beta[c] - (float)mean * sc;
                for (int b = 0; b < fin.B; ++b) {
                    fin.scale[b * 64 + c] = sc; fin.shift[b * 64 + c] = sh; fin.mean[b * 64 + c] = (float)mean; fin.rstd[b * 64 + c] = rstd;
                }
            }
            if (ltid == 0) __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (stat_partial) {
            flush(cur_b);
            const int bf = (t_begin / tiles_y) / tiles_x;
            for (int b = 0; b < B; ++b)   // samples this workgroup never touched read as zero (no memset pass)
                if (b < bf || b > cur_b) {
                    float* dst = stat_partial + ((int64_t)b * nslot + blockIdx.x * 4 + lwv) * 128;
                    dst[lane] = 0.f;
                    dst[64 + lane] = 0.f;
                }
        }
        return;
    }

    // ---------------------------------------------------------------- compute waves
    // wave wv: output channels 32*ct .. +31 (ct = wv >> 1), tile rows 2*rp and 2*rp+1 (rp = wv & 1)
    const int ct = wv >> 1, rp = wv & 1;
    if (P4C_PRIO == 2) __builtin_amdgcn_s_setprio(1);
    P4C_STAMP_RT(3110);
    bf16x8 A[9][4];
    {
        const char* wsrc = reinterpret_cast<const char*>(wp) + (h * 64 + ct * 32 + r) * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) A[tap][ks] = *reinterpret_cast<const bf16x8*>(wsrc + (tap * 4 + ks) * 2048);
    }
    if (MODE >= 2 || BST) lds_barrier();  // pairs with the loaders' barrier after the normalisation rows are in LDS
    lds_barrier();
    int boff[3][4];  // B-operand byte offset inside a ring row: pixel column r+kx, channel slot 2ks+h (swizzled)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) boff[kx][ks] = 2 * rp * RROW + (r + kx) * ROWB + (((2 * ks + h) ^ ring_swz(r + kx)) << 4);
    int soff[4];     // staging offsets of this lane's 4 channel quads (row 0 of the wave; row 1 = +32 pixels)
#pragma unroll
    // (slot XOR (r >> 1) & 7: a row is 128 B = 32 banks, so rows r and r + 2 meet in the same banks; keyed on r & 7 the 8-byte
    // writes of a wave fell on 16 distinct slots only -- a 4-way conflict, the 128 conflict cycles per tile of the round-1
    // counters -- keyed on r >> 1 sixteen consecutive rows cover all 64 banks: two passes, the minimum for 512 bytes)
    for (int g = 0; g < 4; ++g) soff[g] = (2 * rp * 32 + r) * 128 + 8 * h + (((4 * ct + g) ^ ((r >> 1) & 7)) << 4);
    Cur cc = cur_init();
    P4C_STAMP_RT(3111);
    for (int tile = t_begin; tile < t_end; ++tile) {
        P4C_STAMP_RT(3120 + 2 * (tile - t_begin));
        const char* tb0 = lring + cc.p * RROW;
        cur_next(cc);
        const char* ba[3][4];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ba[kx][ks] = tb0 + boff[kx][ks];
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
        // epilogue target: C[co][px]; lane = pixel r (+ half h), register quad g -> channels 32ct + 8g + 4h .. +3, i.e. half h
        // of 16-byte slot 4ct + g of the pixel in the staging tile (slot XOR-swizzled by (pixel >> 1) & 7)
        char* stg = lstg + ((tile - t_begin) & 1) * STGB;
        if (!(P4C_EXP & 16)) {
            // The wave's two output rows use the four input rows j = 0..3 of the ring: row j is tap row ky = j of output row 0
            // and tap row ky = j - 1 of output row 1, so each of the 48 operands (j, kx, 16-channel slice) is read ONCE and feeds
            // one MFMA (j = 0, 3) or two (j = 1, 2): 48 B reads for 72 MFMAs instead of 72.  Order: the single-use rows first,
            // alternating j = 0 / 3 (accumulators alternate), then j = 1 and j = 2.  Reads run LEAD operands ahead of their MFMAs
            // through a ring of LEAD + 1 buffers, one read per MFMA gap at most (bursts of reads beside MFMAs cost 48 instead of
            // 36 cycles per MFMA, tools/diagnostics/mfma_rate.hip).
#ifndef P4C_RING_LEAD
#define P4C_RING_LEAD 8
#endif
            constexpr int LEAD = P4C_RING_LEAD, NBUF_B = LEAD + 1;
            bf16x8 fb[NBUF_B];
#ifndef P4C_RING_ORDER
#define P4C_RING_ORDER 1
#endif
            // operand order.  0 (round 1): the single-use rows first, j = 0 / 3 alternating, then j = 1, j = 2.  1: rows in order
            // j = 0, 1, 2, 3 -- the first output row (acc0) is complete after row 2, so its conversion + staging stores are issued
            // between the last twelve MFMAs (row 3 feeds acc1 only) instead of after them; only acc1's epilogue stays exposed.
            auto q_i = [](int q) __attribute__((always_inline)) {
                return P4C_RING_ORDER ? q % 12 : (q < 24 ? (q >> 1) : (q < 36 ? q - 24 : q - 36));
            };
            auto q_j = [](int q) __attribute__((always_inline)) {
                return P4C_RING_ORDER ? q / 12 : (q < 24 ? ((q & 1) ? 3 : 0) : (q < 36 ? 1 : 2));
            };
            auto issue = [&](int q) __attribute__((always_inline)) {
                const int i = q_i(q), j = q_j(q);
                fb[q % NBUF_B] = *reinterpret_cast<const bf16x8*>(ba[i >> 2][i & 3] + j * RROW);
            };
            auto stage = [&](const f32x16& a, int row, int g) __attribute__((always_inline)) {
                const f32x2 lo = {a[4 * g], a[4 * g + 1]}, hi = {a[4 * g + 2], a[4 * g + 3]};
                u32x2 o;
                o[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(lo, bf16x2));
                o[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(hi, bf16x2));
                *reinterpret_cast<u32x2*>(stg + soff[g] + row * 32 * 128) = o;
            };
#pragma unroll
            for (int q = 0; q < LEAD; ++q) issue(q);
#pragma unroll
            for (int q = 0; q < 48; ++q) {
                const int i = q_i(q), j = q_j(q);
                const int kx = i >> 2, ks = i & 3;
                if (q + LEAD < 48) issue(q + LEAD);
                __builtin_amdgcn_sched_barrier(0);
                if (j <= 2) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j * 3 + kx][ks], fb[q % NBUF_B], acc0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (j >= 1) {
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[(j - 1) * 3 + kx][ks], fb[q % NBUF_B], acc1, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (P4C_RING_ORDER && q >= 38 && q <= 44 && ((q - 38) & 1) == 0) {   // acc0 is final since q = 35: one channel quad per gap
                    stage(acc0, 0, (q - 38) >> 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (!P4C_RING_ORDER) {
#pragma unroll
                for (int g = 0; g < 4; ++g) stage(acc0, 0, g);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) stage(acc1, 1, g);
        } else { acc0[0] = ba[0][0][0]; acc1[3] = ba[1][1][16]; }
        P4C_STAMP_RT(3121 + 2 * (tile - t_begin));
        lds_barrier();
    }
    P4C_STAMP_RT(3112);
}

template <int MODE, bool BST = false>
static int launch_ring_mode(const __bf16* in, const __bf16* wp, const float* in_scale, const float* in_shift, __bf16* out,
                            int out_cs, float* stat_partial, int B, int H, int W, int G, hipStream_t stream, const BatchFin& fin,
                            const RingBwdStats& bst = RingBwdStats{}) {
    P4C_TRY(ensure_dyn_smem((const void*)conv3x3_bf16_ring_kernel<MODE, BST>, ring::SMEM));
    hipLaunchKernelGGL((conv3x3_bf16_ring_kernel<MODE, BST>), dim3(G), dim3(512), ring::SMEM, stream, in, wp, in_scale, in_shift,
                       out, out_cs, stat_partial, B, H, W, fin, bst);
    return P4C_OK;
}

static int launch_conv3x3_bf16_ring(const __bf16* in, const __bf16* wp, const float* in_scale, const float* in_shift,
                                    int in_relu, __bf16* out, int out_cs, float* stat_partial, int B, int H, int W,
                                    hipStream_t stream, const BatchFin* finp, const RingBwdStats* bst, int* nblk_out) {
    BatchFin fin{};
    if (finp && stat_partial) { fin = *finp; fin.slots = stat_partial; }
    if (bst) {
        P4C_CHECK_ARG(!in_scale && !in_relu && stat_partial && !finp && out_cs == 64 && B <= RING_BWD_STATS_MAXB && nblk_out,
                      "conv3x3_bf16_ring: backward statistics need a plain 64-channel launch with a partial buffer and B <= %d",
                      RING_BWD_STATS_MAXB);
        const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + 3) / 4;
        const int64_t ntiles = (int64_t)tiles_x * tiles_y * B;
        const int G = ntiles < num_cus() ? (int)ntiles : num_cus();
        *nblk_out = 4 * G;
        prof_begin(P4C_PROF_CONV3X3_C64, (int64_t)B * H * W, stream);
        const int rc = launch_ring_mode<0, true>(in, wp, nullptr, nullptr, out, out_cs, stat_partial, B, H, W, G, stream, fin, *bst);
        prof_end(P4C_PROF_CONV3X3_C64, stream);
        if (rc != P4C_OK) return rc;
        P4C_CHECK_LAUNCH("conv3x3_bf16_ring(bwd stats)");
        return P4C_OK;
    }
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + 3) / 4;
    int64_t ntiles = (int64_t)tiles_x * tiles_y * B;
    int G = num_cus();
    if (ntiles < G) G = (int)ntiles;
    prof_begin(P4C_PROF_CONV3X3_C64, (int64_t)B * H * W, stream);
    int rc;
    if (in_scale)
        rc = in_relu ? launch_ring_mode<2>(in, wp, in_scale, in_shift, out, out_cs, stat_partial, B, H, W, G, stream, fin)
                     : launch_ring_mode<3>(in, wp, in_scale, in_shift, out, out_cs, stat_partial, B, H, W, G, stream, fin);
    else
        rc = in_relu ? launch_ring_mode<1>(in, wp, in_scale, in_shift, out, out_cs, stat_partial, B, H, W, G, stream, fin)
                     : launch_ring_mode<0>(in, wp, in_scale, in_shift, out, out_cs, stat_partial, B, H, W, G, stream, fin);
    prof_end(P4C_PROF_CONV3X3_C64, stream);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_LAUNCH("conv3x3_bf16_ring");
    return P4C_OK;
}

// ---------------------------------------------------------------------------------------------
// conv_wgrad_bf16: persistent; grid = G workgroups, tiles of 8 x 32 pixels.
//   dW[tap][ci][co] += sum_px In[px + tap][ci] * dOut[px][co]
// MFMA per tap and 16-pixel K step: A[i=ci][k=px], B[k=px][j=co]; both operands are transposing LDS reads
// (ds_read_b64_tr_b16: 4 pixel rows x 16 channels per 16-lane group) of the [pixel][channel] bf16 images.
// Row strides of 64 B (mod 256) make the 4 rows x 2 groups of a half-wave cover all 64 banks once.
template <typename T, int CI, int KS>
__global__ void __launch_bounds__(256, 1)
    conv_wgrad_bf16_kernel(const T* __restrict__ in, const float* __restrict__ in_scale,
                           const float* __restrict__ in_shift, int in_relu, const T* __restrict__ dout,
                           float* __restrict__ partial, int B, int H, int W, int in_cs, int ci_off, int part_cip) {
    constexpr int TH = 8;
    constexpr int HALO = KS / 2;
    constexpr int LH = TH + 2 * HALO, LW = BTW + 2 * HALO;
    constexpr int ROWA = (CI * 2) % 256 == 64 ? CI * 2 : CI * 2 + 64;  // 64 -> 64, 128 -> 192
    constexpr int ROWD = 192;                                            // 64 channels * 2 B + 64
    constexpr int NTAPS = KS * KS;
    constexpr int UNITS = (CI / 32) * 2;
    constexpr int KSPLIT = 4 / UNITS;
    constexpr int KSTEPS = TH * BTW / 16 / KSPLIT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lin = smem;
    char* ldo = smem + LH * LW * ROWA;

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int unit = wv % UNITS, ksl = wv / UNITS;
    const int cit = unit >> 1, cot = unit & 1;
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * B;
    const T* inb = in + ci_off;
    const float* scb = in_scale ? in_scale + ci_off : nullptr;
    const float* shb = in_shift ? in_shift + ci_off : nullptr;

    // transposing-read lane geometry: lane 4q+p of a 16-lane group addresses pixel row q, channels 4p..4p+3
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int a_col = (cit * 32 + tg * 16 + tp * 4) * 2;  // byte offset of the lane's 4 channels in a pixel row
    const int b_col = (cot * 32 + tg * 16 + tp * 4) * 2;

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    auto coords = [&](int t, int& b, int& y0, int& x0) {
        const int ty = t % tiles_y;
        const int rest = t / tiles_y;
        b = rest / tiles_x;
        y0 = ty * TH;
        x0 = (rest - b * tiles_x) * BTW;
    };
    const int t_begin = (int)((int64_t)ntiles * blockIdx.x / gridDim.x);
    const int t_end = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);

    // ---- staging of the two tiles (input with halo, output gradient): global -> registers -> (normalise) -> LDS.
    // bf16 storage: table driven.  Each lane's slots are the same for every tile, so their global / LDS offsets are
    // computed once; per tile an interior tile needs one scalar offset per tensor and no bounds tests (buffer
    // descriptors return 0 beyond the sample, border tiles take a bounds-checked path with the same number of loads).
    constexpr bool TAB = std::is_same<T, __bf16>::value;
    constexpr int C8 = CI / 8;
    constexpr int TOT_IN = LH * LW * C8, NI = (TOT_IN + 255) / 256, ND = TH * BTW * 8 / 256;
    BTile<T, TAB ? 8 : CI, TAB ? 1 : LH, TAB ? 1 : LW> tr;   // generic (fp32 storage) path only
    BTile<T, TAB ? 8 : 64, TAB ? 1 : TH, TAB ? 1 : BTW> td;
    u32x4 ri[TAB ? NI : 1], rd[TAB ? ND : 1];
    int gi[TAB ? NI : 1], li[TAB ? NI : 1], pi[TAB ? NI : 1], gd[TAB ? ND : 1], ld[TAB ? ND : 1];
    const int c8i = threadIdx.x % C8, c8d = threadIdx.x & 7;
    if (TAB) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int idx = threadIdx.x + it * 256, pix = idx / C8;
            const int ly = pix / LW, lx = pix - ly * LW;
            gi[it] = ((ly * W + lx) * in_cs + 8 * c8i) * 2;
            li[it] = pix * ROWA + 16 * c8i;
            pi[it] = (ly << 8) | lx;
        }
#pragma unroll
        for (int it = 0; it < ND; ++it) {
            const int pix = (threadIdx.x + it * 256) >> 3;
            gd[it] = (((pix >> 5) * W + (pix & 31)) * 64 + 8 * c8d) * 2;
            ld[it] = pix * ROWD + 16 * c8d;
        }
    }
    f32x2 nsc[4], nsh[4];   // normalisation of this lane's 8 channels for the tile in flight (fetched with the tile)
#pragma unroll
    for (int k = 0; k < 4; ++k) nsc[k] = nsh[k] = f32x2{0.f, 0.f};
    auto stage_load = [&](int t) __attribute__((always_inline)) {
        int b, y0, x0;
        coords(t, b, y0, x0);
        if constexpr (TAB) {
            if (scb) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    nsc[k] = *reinterpret_cast<const f32x2*>(scb + (int64_t)b * in_cs + 8 * c8i + 2 * k);
                    nsh[k] = *reinterpret_cast<const f32x2*>(shb + (int64_t)b * in_cs + 8 * c8i + 2 * k);
                }
            }
            const __amdgpu_buffer_rsrc_t rsi = make_rsrc(inb + (int64_t)b * H * W * in_cs, (unsigned)(((int64_t)H * W * in_cs - ci_off) * 2));
            const __amdgpu_buffer_rsrc_t rsd = make_rsrc(dout + (int64_t)b * H * W * 64, (unsigned)((int64_t)H * W * 64 * 2));
            const int gy0 = y0 - HALO, gx0 = x0 - HALO;
            if (gy0 >= 0 && y0 + TH + HALO <= H && gx0 >= 0 && x0 + BTW + HALO <= W) {
                const int so = (gy0 * W + gx0) * in_cs * 2, sd = (y0 * W + x0) * 64 * 2;
#pragma unroll
                for (int it = 0; it < NI; ++it)
                    ri[it] = __builtin_amdgcn_raw_buffer_load_b128(rsi, (threadIdx.x + it * 256 < TOT_IN) ? gi[it] : OOB, so, 0);
#pragma unroll
                for (int it = 0; it < ND; ++it) rd[it] = __builtin_amdgcn_raw_buffer_load_b128(rsd, gd[it], sd, 0);
            } else {
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int gy = gy0 + (pi[it] >> 8), gx = gx0 + (pi[it] & 255);
                    const bool ok = (threadIdx.x + it * 256 < TOT_IN) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
                    ri[it] = __builtin_amdgcn_raw_buffer_load_b128(rsi, ok ? ((gy * W + gx) * in_cs + 8 * c8i) * 2 : OOB, 0, 0);
                }
#pragma unroll
                for (int it = 0; it < ND; ++it) {
                    const int pix = (threadIdx.x + it * 256) >> 3;
                    const int gy = y0 + (pix >> 5), gx = x0 + (pix & 31);
                    rd[it] = __builtin_amdgcn_raw_buffer_load_b128(rsd, ((gy < H) & (gx < W)) ? ((gy * W + gx) * 64 + 8 * c8d) * 2 : OOB, 0, 0);
                }
            }
        } else {
            btile_load<T, CI, LH, LW, HALO>(tr, inb, b, y0, x0, H, W, in_cs);
            btile_load<T, 64, TH, BTW, 0>(td, dout, b, y0, x0, H, W, 64);
        }
    };
    auto stage_store = [&](int t) __attribute__((always_inline)) {
        int b, y0, x0;
        coords(t, b, y0, x0);
        if constexpr (TAB) {
            const int mode = scb ? (in_relu ? 2 : 3) : (in_relu ? 1 : 0);
            f32x2 sc[4], sh[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { sc[k] = nsc[k]; sh[k] = nsh[k]; }
            const int gy0 = y0 - HALO, gx0 = x0 - HALO;
            const bool interior = gy0 >= 0 && y0 + TH + HALO <= H && gx0 >= 0 && x0 + BTW + HALO <= W;
            auto put = [&](auto mode_tag) __attribute__((always_inline)) {
                constexpr int M = decltype(mode_tag)::value;
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    u32x4 o = ri[it];
                    if (M != 0) {
                        unsigned int keep = 0xffffffffu;
                        if (!interior) {   // zero padding applies to the NORMALISED activation
                            const int gy = gy0 + (pi[it] >> 8), gx = gx0 + (pi[it] & 255);
                            keep = (((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W)) ? 0xffffffffu : 0u;
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) o[k] = xform2<M>(o[k], sc[k], sh[k]) & keep;
                    }
                    if (threadIdx.x + it * 256 < TOT_IN) *reinterpret_cast<u32x4*>(lin + li[it]) = o;
                }
            };
            if (mode == 0) put(std::integral_constant<int, 0>());
            else if (mode == 1) put(std::integral_constant<int, 1>());
            else if (mode == 2) put(std::integral_constant<int, 2>());
            else put(std::integral_constant<int, 3>());
#pragma unroll
            for (int it = 0; it < ND; ++it) *reinterpret_cast<u32x4*>(ldo + ld[it]) = rd[it];
        } else {
            btile_store<T, CI, LH, LW, HALO, ROWA>(tr, scb, shb, in_relu, lin, b, y0, x0, H, W, in_cs);
            btile_store<T, 64, TH, BTW, 0, ROWD>(td, nullptr, nullptr, 0, ldo, b, y0, x0, H, W, 64);
        }
    };

    int tile = t_begin;
    if (tile < t_end) stage_load(tile);
    P4C_STAMP(3000); P4C_STAMP_RT(3001);
    for (; tile < t_end; ++tile) {
        P4C_STAMP(2000 + 4 * (tile - t_begin) + 0);
        __syncthreads();
        if (!(P4C_EXP & 1) || tile == t_begin) stage_store(tile);
        __syncthreads();
        if (tile + 1 < t_end && !(P4C_EXP & 2)) stage_load(tile + 1);
        // K loop over 16-pixel steps, software pipelined: the 2 + 2*NTAPS transposing LDS reads of step kk+1 are issued
        // before the NTAPS MFMAs of step kk (two operand sets alternate), so the matrix pipe never waits on LDS latency
        struct Ops { s16x4 b[2]; s16x4 a[NTAPS][2]; };
        auto fetch = [&](Ops& o, int kk) __attribute__((always_inline)) {
            const int kstep = ksl * KSTEPS + kk;        // 16 pixels: row kstep>>1, columns (kstep&1)*16 ..
            const int row = kstep >> 1, col0 = (kstep & 1) * 16;
            const int px = col0 + 8 * h + tq;           // this lane's pixel for the first 4-row block (+4 for the second)
            const char* bp = ldo + (row * BTW + px) * ROWD + b_col;
            if (!(P4C_EXP & 64) || kk < 2) {
            o.b[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bp));
            o.b[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bp + 4 * ROWD));
            }
            const char* ap0 = lin + (row * LW + px) * ROWA + a_col;
            if (!(P4C_EXP & 32) || kk < 2)
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int ky = t / KS, kx = t - ky * KS;
                const char* ap = ap0 + (ky * LW + kx) * ROWA;
                o.a[t][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(ap));
                o.a[t][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(ap + 4 * ROWA));
            }
        };
        auto mma = [&](const Ops& o) __attribute__((always_inline)) {
            union { s16x4 s[2]; bf16x8 v; } ub;
            ub.s[0] = o.b[0]; ub.s[1] = o.b[1];
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                union { s16x4 s[2]; bf16x8 v; } u;
                u.s[0] = o.a[t][0]; u.s[1] = o.a[t][1];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, ub.v, acc[t], 0, 0, 0);
            }
        };
        P4C_STAMP(2000 + 4 * (tile - t_begin) + 1);
        static_assert(KSTEPS % 2 == 0, "the pipelined K loop handles steps in pairs");
        constexpr int KLOOP = (P4C_EXP & 16) ? 2 : KSTEPS;
        Ops o0, o1;
        fetch(o0, 0);
#pragma unroll 1
        for (int kk = 0; kk < KLOOP; kk += 2) {
            fetch(o1, kk + 1);
            __builtin_amdgcn_sched_barrier(0);
            mma(o0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < KLOOP) fetch(o0, kk + 2);
            __builtin_amdgcn_sched_barrier(0);
            mma(o1);
            __builtin_amdgcn_sched_barrier(0);
        }
        P4C_STAMP(2000 + 4 * (tile - t_begin) + 2);
    }
    P4C_STAMP(3002); P4C_STAMP_RT(3003);
    // C[ci][co]: lane = co (r), register i -> ci = (i&3) + 8*(i>>2) + 4*h.  One partial per (workgroup, k-slice).
    float* pbase = partial + ((int64_t)blockIdx.x * KSPLIT + ksl) * NTAPS * part_cip * 64;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = ci_off + cit * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            pbase[((int64_t)t * part_cip + ci) * 64 + cot * 32 + r] = acc[t][i];
        }
}

// ---------------------------------------------------------------------------------------------
// conv3x3_wgrad_bf16_ws: weight gradient of the 3x3 conv, 64 input channels (one 64-channel chunk) x 64 output
// channels, bf16 storage -- the role-split form of conv_wgrad_bf16_kernel.  In the single-role kernel the staging of a
// tile (wait, 19 LDS writes, barrier, 19 buffer loads per lane: ~4.7k cycles) is serial with its 8k-cycle matrix phase.
// Here waves 4-7 stage tile t+1 into the other LDS buffer (and prefetch tiles t+2, t+3 into registers) while waves 0-3
// run the K loop of tile t; one LDS-only barrier per tile.  Tiles are 4 x 32 pixels so that two (input + halo, output
// gradient) images fit: 2 x (6x34 + 4x32) pixels x 192 B = 125 KB.
namespace wgws {
constexpr int TH = 4, ROWA = 192, ROWD = 192;
constexpr int MAXB = 32;
// KS = 3: 4 x 32 output pixels + halo;  KS = 1 (the 1x1 output convolution): no halo, one tap
template <int KS> struct Geo {
    static constexpr int HALO = KS / 2, NTAPS = KS * KS;
    static constexpr int LH = TH + 2 * HALO, LW = BTW + 2 * HALO;
    static constexpr int INB = LH * LW * ROWA, DOB = TH * BTW * ROWD, BUFB = INB + DOB;
    static constexpr int SMEM = 2 * BUFB + MAXB * 128 * 4;
    static constexpr int NI = (LH * LW * 8 + 255) / 256, ND = TH * BTW * 8 / 256;   // 7 (4) + 4 slots of 16 B per lane and tile
    static constexpr int TOT_IN = LH * LW * 8;
};
}  // namespace wgws

// NB: `dout` holds dA and the kernel forms dY = alpha * (dA where the forward ReLU was alive) + beta * y + delta while the staging waves load it
// (kernels.hpp: NormBwdCoef -- pass 2 of the normalisation backward without a launch or a dY map of its own).
template <int MODE, bool NB, int KS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
    conv3x3_wgrad_bf16_ws_kernel(const __bf16* __restrict__ in, const float* __restrict__ in_scale,
                                 const float* __restrict__ in_shift, const __bf16* __restrict__ dout,
                                 float* __restrict__ partial, int B, int H, int W, int in_cs, int ci_off, int part_cip, NormBwdCoef nb,
                                 int dout_cs) {
    // (dout_cs: channels per pixel of `dout`, 64 or a smaller multiple of 8 -- "compact" gradients of a convolution with fewer than 64
    // output channels; likewise in_cs - ci_off may be below 64: absent channel octets are staged as zeros)
    using namespace wgws;
    using G = Geo<KS>;
    constexpr int NTAPS = G::NTAPS, HALO = G::HALO, LH = G::LH, LW = G::LW, INB = G::INB, BUFB = G::BUFB, NI = G::NI, ND = G::ND,
                  TOT_IN = G::TOT_IN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lnorm = reinterpret_cast<float*>(smem + 2 * BUFB);

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool loader = wv >= 4;
    const int ltid = threadIdx.x - 256;
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * B;
    const int t_begin = (int)((int64_t)ntiles * blockIdx.x / gridDim.x);
    const int t_end = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
    const __bf16* inb = in + ci_off;

    if (loader) {
        if (t_begin >= t_end) return;
        const int c8 = ltid & 7;
        const bool in_ch = 8 * c8 < in_cs - ci_off, do_ch = 8 * c8 < dout_cs;
        struct Cur { int t, b, tx, ty; };
        auto cur_init = [&]() __attribute__((always_inline)) {
            Cur c;
            c.t = t_begin;
            c.ty = t_begin % tiles_y;
            const int rest = t_begin / tiles_y;
            c.tx = rest % tiles_x;
            c.b = rest / tiles_x;
            return c;
        };
        auto cur_next = [&](Cur& c) __attribute__((always_inline)) {
            ++c.t;
            if (++c.ty == tiles_y) {
                c.ty = 0;
                if (++c.tx == tiles_x) { c.tx = 0; ++c.b; }
            }
        };
        // per-lane tables (slot -> offsets), same for every tile
        int gi[NI], li[NI], pi[NI], gd[ND], ld[ND];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int pix = (ltid + it * 256) >> 3;
            const int ly = pix / LW, lx = pix - ly * LW;
            gi[it] = in_ch ? ((ly * W + lx) * in_cs + 8 * c8) * 2 : OOB;
            li[it] = pix * ROWA + 16 * c8;
            pi[it] = (ly << 8) | lx;
        }
#pragma unroll
        for (int it = 0; it < ND; ++it) {
            const int pix = (ltid + it * 256) >> 3;
            gd[it] = do_ch ? (((pix >> 5) * W + (pix & 31)) * dout_cs + 8 * c8) * 2 : OOB;
            ld[it] = INB + pix * ROWD + 16 * c8;
        }
        struct Img { u32x4 a[NI]; u32x4 d[ND]; u32x4 y[NB ? ND : 1]; };
        Img ta, tb;
        Cur lc = cur_init(), sc_ = cur_init();
        // same number of memory operations on every path (see conv3x3_bf16_ring_kernel): exact s_waitcnt counts
        auto load = [&](Img& im) __attribute__((always_inline)) {
            const bool live = lc.t < t_end && !((P4C_EXP & 2) && lc.t > t_begin + 1);
            const int b = live ? lc.b : 0, y0 = live ? lc.ty * TH : 0, x0 = live ? lc.tx * BTW : 0;
            cur_next(lc);
            const __amdgpu_buffer_rsrc_t rsi = make_rsrc(inb + (int64_t)b * H * W * in_cs, (unsigned)(((int64_t)H * W * in_cs - ci_off) * 2));
            const __amdgpu_buffer_rsrc_t rsd = make_rsrc(dout + (int64_t)b * H * W * dout_cs, (unsigned)((int64_t)H * W * dout_cs * 2));
            const __amdgpu_buffer_rsrc_t rsy = make_rsrc((NB ? reinterpret_cast<const __bf16*>(nb.y) : dout) + (int64_t)b * H * W * 64,
                                                         (unsigned)((int64_t)H * W * 64 * 2));
            const int gy0 = y0 - HALO, gx0 = x0 - HALO;
            if (live && gy0 >= 0 && y0 + TH + HALO <= H && gx0 >= 0 && x0 + BTW + HALO <= W) {
                const int so = (gy0 * W + gx0) * in_cs * 2, sd = (y0 * W + x0) * dout_cs * 2;
#pragma unroll
                for (int it = 0; it < NI; ++it)
                    im.a[it] = __builtin_amdgcn_raw_buffer_load_b128(rsi, (ltid + it * 256 < TOT_IN) ? gi[it] : OOB, so, 0);
#pragma unroll
                for (int it = 0; it < ND; ++it) im.d[it] = __builtin_amdgcn_raw_buffer_load_b128(rsd, gd[it], sd, 0);
                if (NB) {
#pragma unroll
                    for (int it = 0; it < ND; ++it) im.y[it] = __builtin_amdgcn_raw_buffer_load_b128(rsy, gd[it], sd, 0);
                }
            } else {
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int gy = gy0 + (pi[it] >> 8), gx = gx0 + (pi[it] & 255);
                    const bool ok = live & (ltid + it * 256 < TOT_IN) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W) & in_ch;
                    im.a[it] = __builtin_amdgcn_raw_buffer_load_b128(rsi, ok ? ((gy * W + gx) * in_cs + 8 * c8) * 2 : OOB, 0, 0);
                }
#pragma unroll
                for (int it = 0; it < ND; ++it) {
                    const int pix = (ltid + it * 256) >> 3;
                    const int gy = y0 + (pix >> 5), gx = x0 + (pix & 31);
                    im.d[it] = __builtin_amdgcn_raw_buffer_load_b128(rsd, (live & (gy < H) & (gx < W) & do_ch) ? ((gy * W + gx) * dout_cs + 8 * c8) * 2 : OOB, 0, 0);
                    if (NB) im.y[it] = __builtin_amdgcn_raw_buffer_load_b128(rsy, (live & (gy < H) & (gx < W)) ? ((gy * W + gx) * 64 + 8 * c8) * 2 : OOB, 0, 0);
                }
            }
        };
        f32x2 sc[4], sh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) sc[k] = sh[k] = f32x2{0.f, 0.f};
        int sc_b = -1;
        f32x2 nal[4], nbe[4], nde[4], nsc[4], nsh[4];   // NB: alpha, beta, delta, and the forward scale / shift (ReLU mask) of this lane's 8 gradient channels, current sample
#pragma unroll
        for (int k = 0; k < 4; ++k) nal[k] = nbe[k] = nde[k] = nsc[k] = nsh[k] = f32x2{0.f, 0.f};
        int nb_b = -1;
        auto store = [&](const Img& im, char* buf) __attribute__((always_inline)) {
            const int t = sc_.t, b = sc_.b, y0 = sc_.ty * TH, x0 = sc_.tx * BTW;
            cur_next(sc_);
            if (t >= t_end || ((P4C_EXP & 1) && t > t_begin + 1)) return;
            if (NB && b != nb_b) {   // (a workgroup's tiles change sample a few times per launch at most)
                nb_b = b;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int ch = 8 * c8 + 2 * k;
                    const f32x2 ga = *reinterpret_cast<const f32x2*>(nb.gamma + ch), rs = *reinterpret_cast<const f32x2*>(nb.rstd + b * 64 + ch);
                    const f32x2 mu = *reinterpret_cast<const f32x2*>(nb.mean + b * 64 + ch);
                    const f32x2 q1 = *reinterpret_cast<const f32x2*>(nb.k1 + b * 64 + ch), q2 = *reinterpret_cast<const f32x2*>(nb.k2 + b * 64 + ch);
                    nsc[k] = *reinterpret_cast<const f32x2*>(nb.scale + b * 64 + ch);
                    nsh[k] = *reinterpret_cast<const f32x2*>(nb.shift + b * 64 + ch);
                    nal[k] = rs * ga;
                    nbe[k] = -(rs * rs) * q2;
                    nde[k] = rs * rs * q2 * mu - rs * q1;
                }
            }
            if (MODE >= 2 && b != sc_b) {
                sc_b = b;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    sc[k] = *reinterpret_cast<const f32x2*>(lnorm + b * 128 + 8 * c8 + 2 * k);
                    sh[k] = *reinterpret_cast<const f32x2*>(lnorm + b * 128 + 64 + 8 * c8 + 2 * k);
                }
            }
            const int gy0 = y0 - HALO, gx0 = x0 - HALO;
            const bool interior = gy0 >= 0 && y0 + TH + HALO <= H && gx0 >= 0 && x0 + BTW + HALO <= W;
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                u32x4 o = im.a[it];
                if (MODE != 0) {
                    unsigned int keep = 0xffffffffu;
                    if (!interior) {   // zero padding applies to the NORMALISED activation
                        const int gy = gy0 + (pi[it] >> 8), gx = gx0 + (pi[it] & 255);
                        keep = (((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W)) ? 0xffffffffu : 0u;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = xform2<MODE>(o[k], sc[k], sh[k]) & keep;
                }
                if (ltid + it * 256 < TOT_IN) *reinterpret_cast<u32x4*>(buf + li[it]) = o;
            }
#pragma unroll
            for (int it = 0; it < ND; ++it) {
                u32x4 o = im.d[it];
                if (NB) {
                    // (pixels of the tile outside the image: g and y load as zeros and dY would be delta -- they must stay zero)
                    const int pix = (ltid + it * 256) >> 3;
                    const unsigned int keep = ((y0 + (pix >> 5) < H) & (x0 + (pix & 31) < W)) ? 0xffffffffu : 0u;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned int g2 = o[k], y2 = im.y[it][k];
                        const float ylo = __builtin_bit_cast(float, y2 << 16), yhi = __builtin_bit_cast(float, y2 & 0xffff0000u);
                        const float glo = __builtin_fmaf(ylo, nsc[k].x, nsh[k].x) > 0.f ? __builtin_bit_cast(float, g2 << 16) : 0.f;
                        const float ghi = __builtin_fmaf(yhi, nsc[k].y, nsh[k].y) > 0.f ? __builtin_bit_cast(float, g2 & 0xffff0000u) : 0.f;
                        const float lo = __builtin_fmaf(nal[k].x, glo, __builtin_fmaf(nbe[k].x, ylo, nde[k].x));
                        const float hi = __builtin_fmaf(nal[k].y, ghi, __builtin_fmaf(nbe[k].y, yhi, nde[k].y));
                        const f32x2 v2 = {lo, hi};
                        o[k] = __builtin_bit_cast(unsigned int, __builtin_convertvector(v2, bf16x2)) & keep;
                    }
                }
                *reinterpret_cast<u32x4*>(buf + ld[it]) = o;
            }
        };

        load(ta);
        load(tb);
        if (MODE >= 2) {
            for (int i = ltid; i < B * 64; i += 256) {
                const int b = i >> 6, c = i & 63;
                const bool have = ci_off + c < in_cs;   // (a half-empty remainder chunk has no channels beyond in_cs)
                lnorm[b * 128 + c] = have ? in_scale[(int64_t)b * in_cs + ci_off + c] : 0.f;
                lnorm[b * 128 + 64 + c] = have ? in_shift[(int64_t)b * in_cs + ci_off + c] : 0.f;
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        store(ta, smem);
        load(ta);
        lds_barrier();
        for (int tile = t_begin; tile < t_end; tile += 2) {
            store(tb, smem + BUFB);     // compute reads buffer 0
            load(tb);
            lds_barrier();
            if (tile + 1 >= t_end) break;
            store(ta, smem);            // compute reads buffer 1
            load(ta);
            lds_barrier();
        }
        return;
    }

    // ---------------------------------------------------------------- compute waves: one (32 ci x 32 co) unit each
    const int h = lane >> 5, r = lane & 31;
    const int cit = wv >> 1, cot = wv & 1;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int a_col = (cit * 32 + tg * 16 + tp * 4) * 2, b_col = (cot * 32 + tg * 16 + tp * 4) * 2;
    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    if (t_begin < t_end) {
        if (MODE >= 2) lds_barrier();
        lds_barrier();
        // The K loop walks INPUT rows, not output rows: input row j of the tile feeds the taps ky of the gradient rows j - ky, so
        // its three column-shifted operands are read once (6 transposed reads) and used by up to nine MFMAs, and the four
        // gradient-row operands of a half tile stay in registers: 88 LDS reads per tile for 72 MFMAs instead of 160 (the reads, not
        // the matrix pipe, bounded this phase: 58 cycles per MFMA at 2.2 reads each).
        struct AOps { s16x4 a[KS][2]; };
        for (int tile = t_begin; tile < t_end; ++tile) {
            const char* lin = smem + ((tile - t_begin) & 1) * BUFB;
            const char* ldo = lin + INB;
            auto fetch_a = [&](AOps& o, int j, int col0) __attribute__((always_inline)) {
                const char* ap0 = lin + (j * LW + col0 + 8 * h + tq) * ROWA + a_col;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    o.a[kx][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(ap0 + kx * ROWA));
                    o.a[kx][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(ap0 + (kx + 4) * ROWA));
                }
            };
#pragma unroll
            for (int half = 0; half < ((P4C_EXP & 16) ? 0 : 2); ++half) {
                const int col0 = half * 16;
                s16x4 bo[TH][2];
#pragma unroll
                for (int rr = 0; rr < TH; ++rr) {
                    const char* bp = ldo + (rr * BTW + col0 + 8 * h + tq) * ROWD + b_col;
                    bo[rr][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bp));
                    bo[rr][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bp + 4 * ROWD));
                }
                AOps ao[2];
                fetch_a(ao[0], 0, col0);
#pragma unroll
                for (int j = 0; j < LH; ++j) {
                    if (j + 1 < LH) fetch_a(ao[(j + 1) & 1], j + 1, col0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ky = 0; ky < KS; ++ky) {
                        const int rr = j - ky;
                        if (rr < 0 || rr >= TH) continue;
                        union { s16x4 s[2]; bf16x8 v; } ub;
                        ub.s[0] = bo[rr][0]; ub.s[1] = bo[rr][1];
#pragma unroll
                        for (int kx = 0; kx < KS; ++kx) {
                            union { s16x4 s[2]; bf16x8 v; } u;
                            u.s[0] = ao[j & 1].a[kx][0]; u.s[1] = ao[j & 1].a[kx][1];
                            acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, ub.v, acc[ky * KS + kx], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            lds_barrier();
        }
    }
    // C[ci][co]: lane = co (r), register i -> ci = (i&3) + 8*(i>>2) + 4*h.  One partial per workgroup.
    float* pbase = partial + (int64_t)blockIdx.x * NTAPS * part_cip * 64;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = ci_off + cit * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (ci < part_cip) pbase[((int64_t)t * part_cip + ci) * 64 + cot * 32 + r] = acc[t][i];   // (a 32-channel remainder runs as a half-empty chunk)
        }
}

template <int MODE, bool NB, int KS = 3>
static int launch_wgrad_ws_mode(const __bf16* in, const float* in_scale, const float* in_shift, const __bf16* dout, float* partial,
                                int G, int B, int H, int W, int in_cs, int ci_off, int part_cip, hipStream_t stream, const NormBwdCoef& nb,
                                int dout_cs = 64) {
    P4C_TRY(ensure_dyn_smem((const void*)conv3x3_wgrad_bf16_ws_kernel<MODE, NB, KS>, wgws::Geo<KS>::SMEM));
    hipLaunchKernelGGL((conv3x3_wgrad_bf16_ws_kernel<MODE, NB, KS>), dim3(G), dim3(512), wgws::Geo<KS>::SMEM, stream, in, in_scale,
                       in_shift, dout, partial, B, H, W, in_cs, ci_off, part_cip, nb, dout_cs);
    return P4C_OK;
}

static int launch_conv3x3_wgrad_bf16_ws(const __bf16* in, const float* in_scale, const float* in_shift, int in_relu,
                                        const __bf16* dout, float* partial, int G, int B, int H, int W, int in_cs, int ci_off,
                                        int part_cip, hipStream_t stream, const NormBwdCoef* nbp = nullptr, int ks = 3, int dout_cs = 64) {
    const int tiles = ((H + 3) / 4) * ((W + BTW - 1) / BTW) * B;
    if (tiles < G) G = tiles;
    const NormBwdCoef nb = nbp ? *nbp : NormBwdCoef{};
    if (nbp && dout_cs != 64) return fail(P4C_ERR_INVALID, "conv_wgrad_bf16: NormBwdCoef needs 64-channel gradients");
    if (ks == 1) {   // the 1x1 output convolution: same staging, one tap
        if (nbp) return fail(P4C_ERR_INVALID, "conv_wgrad_bf16: NormBwdCoef is for the 3x3 blocks");
#define P4C_WG1(M) launch_wgrad_ws_mode<M, false, 1>(in, in_scale, in_shift, dout, partial, G, B, H, W, in_cs, ci_off, part_cip, stream, nb, dout_cs)
        P4C_TRY(in_scale ? (in_relu ? P4C_WG1(2) : P4C_WG1(3)) : (in_relu ? P4C_WG1(1) : P4C_WG1(0)));
#undef P4C_WG1
        P4C_CHECK_LAUNCH("conv1x1_wgrad_bf16_ws");
        return P4C_OK;
    }
    const int tag = (in_cs == 64) ? P4C_PROF_WGRAD3X3_C64 : 0;
    if (tag) prof_begin(tag, (int64_t)B * H * W, stream);
    int rc;
#define P4C_WG(M)                                                                                                                  \
    (nbp ? launch_wgrad_ws_mode<M, true>(in, in_scale, in_shift, dout, partial, G, B, H, W, in_cs, ci_off, part_cip, stream, nb)   \
         : launch_wgrad_ws_mode<M, false>(in, in_scale, in_shift, dout, partial, G, B, H, W, in_cs, ci_off, part_cip, stream, nb, dout_cs))
    if (in_scale)
        rc = in_relu ? P4C_WG(2) : P4C_WG(3);
    else
        rc = in_relu ? P4C_WG(1) : P4C_WG(0);
#undef P4C_WG
    if (tag) prof_end(tag, stream);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_LAUNCH("conv3x3_wgrad_bf16_ws");
    return P4C_OK;
}

// ---------------------------------------------------------------------------------------------
template <typename T, int CI, int KS, int NBUF>
static int launch_conv_fwd_bf16(const T* in, const __bf16* wp, const float* in_scale, const float* in_shift, int in_relu,
                                T* out, int out_cs, float* stat_partial, int B, int H, int W, int m_blocks,
                                hipStream_t stream) {
    constexpr int TH = 4;
    constexpr int HALO = KS / 2;
    constexpr int LH = TH + 2 * HALO, LW = BTW + 2 * HALO;
    constexpr int TILEB = (LH * LW * (CI * 2 + 16) + 15) / 16 * 16;
    const size_t scratch = (4 * 32 * 33 + 2 * 4 * 2 * 64) * sizeof(float);
    const size_t smem = (size_t)KS * KS * CI * 64 * 2 + (NBUF == 2 ? (size_t)2 * TILEB + scratch + (KS == 1 ? 2 * 4 * 32 * 128 : 0) : (TILEB > scratch ? TILEB : scratch));
    if (NBUF == 2)
        P4C_TRY(ensure_dyn_smem((const void*)conv_fwd_bf16_ws_kernel<T, CI, KS>, (int)smem));
    else
        P4C_TRY(ensure_dyn_smem((const void*)conv_fwd_bf16_kernel<T, CI, KS, NBUF>, (int)smem));
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + TH - 1) / TH;
    int64_t ntiles = (int64_t)tiles_x * tiles_y * B;
    int G = num_cus();
    if (ntiles < G) G = (int)ntiles;
    const int tag = (CI == 64 && KS == 3 && m_blocks == 1) ? P4C_PROF_CONV3X3_C64 : 0;
    if (tag) prof_begin(tag, (int64_t)B * H * W, stream);
    if (NBUF == 2)
        hipLaunchKernelGGL((conv_fwd_bf16_ws_kernel<T, CI, KS>), dim3(G, m_blocks), dim3(512), smem, stream, in, wp, in_scale,
                           in_shift, in_relu, out, out_cs, stat_partial, B, H, W);
    else
        hipLaunchKernelGGL((conv_fwd_bf16_kernel<T, CI, KS, NBUF>), dim3(G, m_blocks), dim3(256), smem, stream, in, wp,
                           in_scale, in_shift, in_relu, out, out_cs, stat_partial, B, H, W);
    if (tag) prof_end(tag, stream);
    P4C_CHECK_LAUNCH("conv_fwd_bf16");
    return P4C_OK;
}

template <typename T, int CI, int KS>
static int launch_conv_wgrad_bf16(const T* in, const float* in_scale, const float* in_shift, int in_relu, const T* dout,
                                  float* partial, int G, int B, int H, int W, int in_cs, int ci_off, int part_cip,
                                  hipStream_t stream) {
    constexpr int HALO = KS / 2;
    constexpr int LH = 8 + 2 * HALO, LW = BTW + 2 * HALO;
    constexpr int ROWA = (CI * 2) % 256 == 64 ? CI * 2 : CI * 2 + 64;
    const size_t smem = (size_t)LH * LW * ROWA + (size_t)8 * BTW * 192;
    auto kern = conv_wgrad_bf16_kernel<T, CI, KS>;
    P4C_TRY(ensure_dyn_smem((const void*)kern, (int)smem));
    const int tag = (CI == 64 && KS == 3 && in_cs == 64) ? P4C_PROF_WGRAD3X3_C64 : 0;
    if (tag) prof_begin(tag, (int64_t)B * H * W, stream);
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), smem, stream, in, in_scale, in_shift, in_relu, dout, partial, B, H, W,
                       in_cs, ci_off, part_cip);
    if (tag) prof_end(tag, stream);
    P4C_CHECK_LAUNCH("conv_wgrad_bf16");
    return P4C_OK;
}

// rows per sample of the statistics partial buffer written by conv_fwd_bf16
int conv_bf16_stat_slots(int CI, int storage, int B, int H, int W, int ks) {
    if (conv_bf16_is_rows(storage, CI, ks, 1, 64, B, H, W)) return conv_rows_stat_slots(B, H, W);   // per (workgroup, loader wave)
    const int tiles = ((H + 3) / 4) * ((W + BTW - 1) / BTW);
    if (CI > 64 && storage != P4C_BF16) return tiles;  // per-tile partials (single-buffer kernel, fp32 storage)
    int64_t ntiles = (int64_t)tiles * B;
    const int G = ntiles < num_cus() ? (int)ntiles : num_cus();
    return G * 4;               // per (workgroup, compute wave) partials
}

template <typename T>
static int conv_fwd_bf16_t(const T* in, int CI, const void* wp, int ks, const float* in_scale, const float* in_shift,
                           int in_relu, T* out, int out_cs, float* stat_partial, int B, int H, int W, int m_blocks,
                           hipStream_t stream) {
    const __bf16* w = (const __bf16*)wp;
#define P4C_CASE(ci, k, nb)                                                                                          \
    if (CI == ci && ks == k)                                                                                         \
        return launch_conv_fwd_bf16<T, ci, k, nb>(in, w, in_scale, in_shift, in_relu, out, out_cs, stat_partial, B, H, \
                                                  W, m_blocks, stream);
    P4C_CASE(32, 3, 2) P4C_CASE(64, 3, 2) P4C_CASE(96, 3, 1) P4C_CASE(32, 1, 2) P4C_CASE(64, 1, 2) P4C_CASE(96, 1, 1)
#undef P4C_CASE
    return fail(P4C_ERR_UNSUPPORTED, "conv_fwd_bf16: unsupported (CI=%d, ks=%d)", CI, ks);
}

static bool is_tile_ring(int storage, int CI, int ks, int m_blocks, int out_cs, int B, int H, int W) {
    return storage == P4C_BF16 && CI == 64 && ks == 3 && m_blocks == 1 && out_cs % 8 == 0 && B <= ring::MAXB &&
           (int64_t)H * W * out_cs * 2 < (int64_t)1 << 31;  // per-sample byte offsets of the buffer descriptors are 32-bit
}

bool conv_bf16_is_ring(int storage, int CI, int ks, int m_blocks, int out_cs, int B, int H, int W) {
    return conv_bf16_is_rows(storage, CI, ks, m_blocks, out_cs, B, H, W) || is_tile_ring(storage, CI, ks, m_blocks, out_cs, B, H, W);
}

bool conv_bf16_bwd_stats_ok(int storage, int B, int H, int W) {
    if (conv_bf16_is_rows(storage, 64, 3, 1, 64, B, H, W)) return conv_rows_stat_slots(B, H, W) <= NORM_BWD_MAX_BLOCKS;
    return is_tile_ring(storage, 64, 3, 1, 64, B, H, W) && B <= RING_BWD_STATS_MAXB && 4 * num_cus() <= NORM_BWD_MAX_BLOCKS;
}

int conv_fwd_bf16(const void* in, int storage, int CI, const void* wp, int ks, const float* in_scale,
                  const float* in_shift, int in_relu, void* out, int out_cs, float* stat_partial, int B, int H, int W,
                  int m_blocks, hipStream_t stream, const BatchFin* fin, const RingBwdStats* bst, int* nblk_out, const NormBwdCoef* nb) {
    if (conv_bf16_is_rows(storage, CI, ks, m_blocks, out_cs, B, H, W))
        return launch_conv3x3_bf16_rows(in, wp, ks, in_scale, in_shift, in_relu, out, out_cs, stat_partial, B, H, W, stream, fin, bst,
                                        nblk_out, nb);
    if (nb) return fail(P4C_ERR_INVALID, "conv_fwd_bf16: NormBwdCoef needs the row kernel");
    if (is_tile_ring(storage, CI, ks, m_blocks, out_cs, B, H, W))
        return launch_conv3x3_bf16_ring((const __bf16*)in, (const __bf16*)wp, in_scale, in_shift, in_relu, (__bf16*)out, out_cs,
                                        stat_partial, B, H, W, stream, fin, bst, nblk_out);
    if (fin || bst) return fail(P4C_ERR_INVALID, "conv_fwd_bf16: in-kernel statistics need the ring kernel");
    if (storage == P4C_BF16)
        return conv_fwd_bf16_t<__bf16>((const __bf16*)in, CI, wp, ks, in_scale, in_shift, in_relu, (__bf16*)out, out_cs,
                                       stat_partial, B, H, W, m_blocks, stream);
    return conv_fwd_bf16_t<float>((const float*)in, CI, wp, ks, in_scale, in_shift, in_relu, (float*)out, out_cs, stat_partial,
                                  B, H, W, m_blocks, stream);
}

int wgrad_reduce(const float* partial, int nslots, int ks, int CI_pad, int ci_lo, int ci_hi, int CO, int CI, float* grad,
                 hipStream_t stream);

template <typename T>
static int conv_wgrad_bf16_t(const T* in, int CI, int ks, const float* in_scale, const float* in_shift, int in_relu,
                             const T* dout, float* partial, int G, int B, int H, int W, int CO, int CIreal, float* grad,
                             hipStream_t stream, const NormBwdCoef* nb) {
    if (diag_skip(32)) return P4C_OK;
    if (CI % 32 != 0 || CI <= 0 || CI > 256) return fail(P4C_ERR_UNSUPPORTED, "conv_wgrad_bf16: unsupported CI=%d", CI);
    if (ks != 1 && ks != 3) return fail(P4C_ERR_UNSUPPORTED, "conv_wgrad_bf16: unsupported ks=%d", ks);
    const int tiles = ((H + 7) / 8) * ((W + BTW - 1) / BTW) * B;
    if (tiles < G) G = tiles;
    for (int off = 0; off < CI;) {
        const int chunk = (CI - off >= 64) ? 64 : 32;
        int rc;
        if (std::is_same<T, __bf16>::value && conv_wgrad_rows_ok(P4C_BF16, CI, off, 64, ks, G, B, H, W)) {
            // maps at least 64 pixels wide: the row-streaming kernel (conv_wgrad_rows.hip), one launch per chunk -- the 96-channel
            // first convolution as a full chunk and a THIN one (its real channels beyond 64: one octet at 69 inputs); same partial
            // layout, same reduction
            int nslots = 0;
            if (off < CIreal) {
                P4C_TRY(launch_conv3x3_wgrad_bf16_rows(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, stream, nb, &nslots, CI, off,
                                                       CIreal, CI));
                if (!diag_skip(8)) P4C_TRY(wgrad_reduce(partial, nslots, 3, CI, off, off + chunk, CO, CIreal, grad, stream));
            }
            off += chunk;
            continue;
        }
        // (P4C_WGWS_REM=1, A/B switch: the 32-channel remainder of a 96-channel input -- the first convolution -- on the role-split
        // kernel as a half-empty 64-channel chunk, absent channel octets staged as zeros; it then takes NormBwdCoef.  Measured:
        // neutral without NormBwdCoef, 0.04 ms per step SLOWER with it -- block 0's weight gradient is the tail of the backward and
        // both of its launches then read y as well -- so the tiled kernel and the norm_bwd_apply launch stay the default)
        const bool rem_ws = chunk == 32 && ks == 3 && off > 0 && diag_env("P4C_WGWS_REM") != nullptr;
        const bool ws_ok = (chunk == 64 || rem_ws) && (ks == 3 || diag_env("P4C_NO_WGWS_1X1") == nullptr) && std::is_same<T, __bf16>::value &&
                           B <= wgws::MAXB && diag_env("P4C_NO_WGWS") == nullptr;
        if (nb && !ws_ok) return fail(P4C_ERR_INVALID, "conv_wgrad_bf16: NormBwdCoef needs the role-split 3x3 kernel (64-channel chunk, bf16)");
        if (ws_ok)
            rc = launch_conv3x3_wgrad_bf16_ws((const __bf16*)in, in_scale, in_shift, in_relu, (const __bf16*)dout, partial, G, B, H, W,
                                              CI, off, CI, stream, nb, ks);
        else if (chunk == 64 && ks == 3)
            rc = launch_conv_wgrad_bf16<T, 64, 3>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        else if (chunk == 32 && ks == 3)
            rc = launch_conv_wgrad_bf16<T, 32, 3>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        else if (chunk == 64)
            rc = launch_conv_wgrad_bf16<T, 64, 1>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        else
            rc = launch_conv_wgrad_bf16<T, 32, 1>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        if (rc != P4C_OK) return rc;
        rc = diag_skip(8) ? P4C_OK : wgrad_reduce(partial, G * ((chunk == 64 || ws_ok) ? 1 : 2), ks, CI, off, off + chunk, CO, CIreal, grad, stream);
        if (rc != P4C_OK) return rc;
        off += chunk;
    }
    return P4C_OK;
}

// every chunk of this weight gradient runs on the role-split kernel, i.e. the launch takes NormBwdCoef
bool conv_wgrad_bf16_takes_nb(int storage, int CI, int ks, int B) {
    if (storage != P4C_BF16 || ks != 3 || B > wgws::MAXB || diag_env("P4C_NO_WGWS")) return false;
    return CI % 64 == 0 || (CI % 64 == 32 && CI > 64 && diag_env("P4C_WGWS_REM") != nullptr);
}

int conv_wgrad_bf16(const void* in, int storage, int CI, int ks, const float* in_scale, const float* in_shift, int in_relu,
                    const void* dout, float* partial, int G, int B, int H, int W, int CO, int CIreal, float* grad,
                    hipStream_t stream, const NormBwdCoef* nb) {
    if (storage == P4C_BF16)
        return conv_wgrad_bf16_t<__bf16>((const __bf16*)in, CI, ks, in_scale, in_shift, in_relu, (const __bf16*)dout, partial, G,
                                         B, H, W, CO, CIreal, grad, stream, nb);
    if (nb) return fail(P4C_ERR_INVALID, "conv_wgrad_bf16: NormBwdCoef needs bf16 storage");
    return conv_wgrad_bf16_t<float>((const float*)in, CI, ks, in_scale, in_shift, in_relu, (const float*)dout, partial, G, B, H,
                                    W, CO, CIreal, grad, stream, nullptr);
}

// Weight gradient of a plain convolution with compact channel counts (conv_rows.hip: CPT): in (B,H,W,in_cs), dout (B,H,W,dout_cs),
// both <= 64 channels (multiples of 8); one role-split launch + the fixed-order reduction.  partial: conv_wgrad workspace of CI_pad 64.
int conv_wgrad_bf16_compact(const void* in, int in_cs, int ks, const void* dout, int dout_cs, float* partial, int G, int B, int H, int W,
                            int CO, int CIreal, float* grad, hipStream_t stream) {
    if (in_cs <= 0 || dout_cs <= 0 || in_cs > 64 || dout_cs > 64 || in_cs % 8 || dout_cs % 8 || (ks != 1 && ks != 3) || B > wgws::MAXB)
        return fail(P4C_ERR_UNSUPPORTED, "conv_wgrad_bf16_compact: unsupported shape (%d / %d channels, ks %d, B %d)", in_cs, dout_cs, ks, B);
    P4C_TRY(launch_conv3x3_wgrad_bf16_ws((const __bf16*)in, nullptr, nullptr, 0, (const __bf16*)dout, partial, G, B, H, W, in_cs, 0, 64, stream,
                                         nullptr, ks, dout_cs));
    const int tiles = ((H + 3) / 4) * ((W + BTW - 1) / BTW) * B;
    return wgrad_reduce(partial, tiles < G ? tiles : G, ks, 64, 0, 64, CO, CIreal, grad, stream);
}

#ifdef P4C_STAMPS
}  // namespace p4c
extern "C" int p4c_debug_set_stamps(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(p4c::g_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
namespace p4c {
#endif

int prep_weights_bf16(const float* w, int CO, int CI, int ks, int transpose_flip, int M_pad, int K_pad, void* out,
                      hipStream_t stream) {
    const int total = M_pad * K_pad * ks * ks;
    hipLaunchKernelGGL(prep_weights_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, CO, CI, ks * ks,
                       transpose_flip, M_pad, K_pad, (__bf16*)out);
    P4C_CHECK_LAUNCH("prep_weights_bf16");
    return P4C_OK;
}

// ---------------------------------------------------------------------------------------------
// prep_weights_batch: every weight tensor of a network re-laid in ONE launch (blockIdx.y = job); the job table
// travels in the kernel arguments.  bf16 = 0 writes the fp32 stream of conv_f32.hip (k = 8q + 4h + s).
__global__ void __launch_bounds__(256) prep_weights_batch_kernel(PrepBatch pb) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < pb.n_zero) pb.zero_words[threadIdx.x] = 0u;
    const PrepJob& jb = pb.job[blockIdx.y];
    const int total = jb.M_pad * jb.K_pad * jb.ntaps;
    const int kv = pb.bf16 ? 8 : 4;           // k values per lane slot
    const int ksteps = jb.K_pad / (2 * kv);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int t = i;
        const int j = t % kv; t /= kv;
        const int m_l = t & 63; t >>= 6;
        const int h = t & 1; t >>= 1;
        const int ks = t % ksteps; t /= ksteps;
        const int tap = t % jb.ntaps;
        const int mb = t / jb.ntaps;
        const int k = 2 * kv * ks + kv * h + j, m = 64 * mb + m_l;
        float v = 0.0f;
        if (!jb.transpose_flip) {
            if (m < jb.CO && k < jb.CI) v = jb.w[((int64_t)m * jb.CI + k) * jb.ntaps + tap];
        } else {
            if (k < jb.CO && m < jb.CI) v = jb.w[((int64_t)k * jb.CI + m) * jb.ntaps + (jb.ntaps - 1 - tap)];
        }
        if (pb.bf16) reinterpret_cast<__bf16*>(jb.out)[i] = (__bf16)v;
        else reinterpret_cast<float*>(jb.out)[i] = v;
    }
}

int prep_weights_batch(const PrepBatch& pb, hipStream_t stream) {
    if (pb.n <= 0) return P4C_OK;
    int maxtotal = 0;
    for (int i = 0; i < pb.n; ++i) {
        const int t = pb.job[i].M_pad * pb.job[i].K_pad * pb.job[i].ntaps;
        if (t > maxtotal) maxtotal = t;
    }
    int bx = (maxtotal + 256 * 4 - 1) / (256 * 4);
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(prep_weights_batch_kernel, dim3(bx, pb.n), dim3(256), 0, stream, pb);
    P4C_CHECK_LAUNCH("prep_weights_batch");
    return P4C_OK;
}

}  // namespace p4c
