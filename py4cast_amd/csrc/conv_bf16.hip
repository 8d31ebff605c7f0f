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
#include "common.hpp"

namespace p4c {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BTW = 32;  // tile width in pixels

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
// Register image of a (LH x LW) halo tile of an fp32 NHWC tensor, 8 channels (32 B) per slot.
template <int CI, int LH, int LW>
struct BTile {
    static constexpr int C8 = CI / 8;
    static constexpr int TOTAL = LH * LW * C8;
    static constexpr int ITERS = (TOTAL + 255) / 256;
    f32x4 lo[ITERS], hi[ITERS];
};

template <int CI, int LH, int LW, int HALO>
__device__ __forceinline__ void btile_load(BTile<CI, LH, LW>& t, const float* __restrict__ in, int b, int y0, int x0, int H,
                                           int W, int in_cs) {
    constexpr int C8 = CI / 8;
#pragma unroll
    for (int it = 0; it < BTile<CI, LH, LW>::ITERS; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int pix = idx / C8, c8 = idx - pix * C8;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
        if (idx < BTile<CI, LH, LW>::TOTAL && gy >= 0 && gy < H && gx >= 0 && gx < W) {
            const float* p = in + (((int64_t)b * H + gy) * W + gx) * in_cs + 8 * c8;
            lo = *reinterpret_cast<const f32x4*>(p);
            hi = *reinterpret_cast<const f32x4*>(p + 4);
        }
        t.lo[it] = lo;
        t.hi[it] = hi;
    }
}

// transform (norm + relu of the producer layer), round to bf16, write [pixel][channel] rows of ROWB bytes
template <int CI, int LH, int LW, int HALO, int ROWB>
__device__ __forceinline__ void btile_store(const BTile<CI, LH, LW>& t, const float* __restrict__ scale,
                                            const float* __restrict__ shift, int relu, char* lds, int b, int y0, int x0,
                                            int H, int W, int sc_cs) {
    constexpr int C8 = CI / 8;
    constexpr bool FIXED = (256 % C8) == 0;  // the thread's channel octet is the same in every iteration
    f32x4 sc_lo = {1, 1, 1, 1}, sc_hi = {1, 1, 1, 1}, sh_lo = {0, 0, 0, 0}, sh_hi = {0, 0, 0, 0};
    if (FIXED && scale) {
        const int c8 = threadIdx.x % C8;
        const float* sc = scale + (int64_t)b * sc_cs + 8 * c8;
        const float* sh = shift + (int64_t)b * sc_cs + 8 * c8;
        sc_lo = *reinterpret_cast<const f32x4*>(sc); sc_hi = *reinterpret_cast<const f32x4*>(sc + 4);
        sh_lo = *reinterpret_cast<const f32x4*>(sh); sh_hi = *reinterpret_cast<const f32x4*>(sh + 4);
    }
#pragma unroll
    for (int it = 0; it < BTile<CI, LH, LW>::ITERS; ++it) {
        const int idx = threadIdx.x + it * 256;
        if (idx >= BTile<CI, LH, LW>::TOTAL) break;
        const int pix = idx / C8, c8 = idx - pix * C8;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        f32x4 lo = t.lo[it], hi = t.hi[it];
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
            if (scale) {
                if (!FIXED) {
                    const float* sc = scale + (int64_t)b * sc_cs + 8 * c8;
                    const float* sh = shift + (int64_t)b * sc_cs + 8 * c8;
                    sc_lo = *reinterpret_cast<const f32x4*>(sc); sc_hi = *reinterpret_cast<const f32x4*>(sc + 4);
                    sh_lo = *reinterpret_cast<const f32x4*>(sh); sh_hi = *reinterpret_cast<const f32x4*>(sh + 4);
                }
                lo = lo * sc_lo + sh_lo;
                hi = hi * sc_hi + sh_hi;
            }
            if (relu) {
                lo.x = fmaxf(lo.x, 0.f); lo.y = fmaxf(lo.y, 0.f); lo.z = fmaxf(lo.z, 0.f); lo.w = fmaxf(lo.w, 0.f);
                hi.x = fmaxf(hi.x, 0.f); hi.y = fmaxf(hi.y, 0.f); hi.z = fmaxf(hi.z, 0.f); hi.w = fmaxf(hi.w, 0.f);
            }
        }
        *reinterpret_cast<bf16x8*>(lds + pix * ROWB + 16 * c8) = pack8(lo, hi);
    }
}

// ---------------------------------------------------------------------------------------------
// conv_fwd_bf16: persistent, grid = G workgroups x M_pad/64 (blockIdx.y), 256 threads.
// Tile TH x 32 pixels x 64 output channels; wave w owns rows [w*RW, (w+1)*RW), RW = TH/4.
// LDS: weights [tap][kstep][h][64][8] bf16, then the halo tile with rows padded by one 16-byte slot
// (row stride = odd number of slots -> the 16 lanes of a ds_read_b128 group hit 16 distinct slots).
template <int CI, int KS, int TH>
__global__ void __launch_bounds__(256, 1)
    conv_fwd_bf16_kernel(const float* __restrict__ in, const __bf16* __restrict__ wp, const float* __restrict__ in_scale,
                         const float* __restrict__ in_shift, int in_relu, float* __restrict__ out, int out_cs,
                         float* __restrict__ stat_partial, int B, int H, int W) {
    constexpr int HALO = KS / 2;
    constexpr int LH = TH + 2 * HALO, LW = BTW + 2 * HALO;
    constexpr int ROWB = CI * 2 + 16;
    constexpr int RW = TH / 4;
    constexpr int NTAPS = KS * KS;
    constexpr int NKS = CI / 16;
    constexpr int WBYTES = NTAPS * CI * 64 * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lw = smem;
    char* lt = smem + WBYTES;

    const int mb = blockIdx.y;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * B;

    // weights -> LDS once per workgroup (linear copy, 16 B per thread and step)
    {
        const char* src = reinterpret_cast<const char*>(wp) + (int64_t)mb * WBYTES;
        for (int i = threadIdx.x * 16; i < WBYTES; i += 256 * 16)
            *reinterpret_cast<f32x4*>(lw + i) = *reinterpret_cast<const f32x4*>(src + i);
    }

    BTile<CI, LH, LW> tr;
    // each workgroup walks a contiguous run of tiles ordered down a 32-pixel-wide strip (ty fastest): the two halo
    // rows shared with the previous tile were fetched by this same CU a moment ago (L2 hits, not HBM re-reads)
    const int t_begin = (int)((int64_t)ntiles * blockIdx.x / gridDim.x);
    const int t_end = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
    int tile = t_begin;
    auto coords = [&](int t, int& b, int& y0, int& x0, int& canon) {
        const int ty = t % tiles_y;
        const int rest = t / tiles_y;
        const int tx = rest % tiles_x;
        b = rest / tiles_x;
        y0 = ty * TH;
        x0 = tx * BTW;
        canon = (b * tiles_y + ty) * tiles_x + tx;
    };
    if (tile < t_end) {
        int b, y0, x0, cn;
        coords(tile, b, y0, x0, cn);
        btile_load<CI, LH, LW, HALO>(tr, in, b, y0, x0, H, W, CI);
    }
    for (; tile < t_end; ++tile) {
        int b, y0, x0, canon;
        coords(tile, b, y0, x0, canon);
        __syncthreads();  // previous tile (and its statistics scratch) fully consumed; weights landed
        btile_store<CI, LH, LW, HALO, ROWB>(tr, in_scale, in_shift, in_relu, lt, b, y0, x0, H, W, CI);
        __syncthreads();
        if (tile + 1 < t_end) {  // next tile's loads fly during this tile's MFMAs
            int nb, ny, nx, ncn;
            coords(tile + 1, nb, ny, nx, ncn);
            btile_load<CI, LH, LW, HALO>(tr, in, nb, ny, nx, H, W, CI);
        }

        f32x16 acc[2][RW];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int pt = 0; pt < RW; ++pt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[ct][pt][i] = 0.f;

        const char* wl = lw + (h * 64 + r) * 16;
#pragma unroll 1
        for (int tap = 0; tap < NTAPS; ++tap) {
            const int ky = tap / KS, kx = tap - ky * KS;
            const char* pl = lt + ((wv * RW + ky) * LW + (r + kx)) * ROWB + 16 * h;
            const char* wt = wl + tap * NKS * 2048;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(wt + ks * 2048);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(wt + ks * 2048 + 512);
#pragma unroll
                for (int pt = 0; pt < RW; ++pt) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(pl + pt * LW * ROWB + 32 * ks);
                    acc[0][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bv, acc[0][pt], 0, 0, 0);
                    acc[1][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bv, acc[1][pt], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: C[co][px]; lane = pixel r (+ half h), register i -> co = (i&3) + 8*(i>>2) + 4*h
        const int gx = x0 + r;
        if (stat_partial) __syncthreads();  // tile reads done: its LDS is reused for the statistics transpose
        float* tw = reinterpret_cast<float*>(lt) + wv * (64 * 33);
        float a1s = 0.f, a2s = 0.f;
#pragma unroll
        for (int pt = 0; pt < RW; ++pt) {
            const int gy = y0 + wv * RW + pt;
            const bool valid = (gy < H) && (gx < W);
            float* orow = out + (((int64_t)b * H + gy) * W + gx) * out_cs + mb * 64 + 4 * h;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]};
                    if (valid) *reinterpret_cast<f32x4*>(orow + ct * 32 + 8 * g) = v;
                    if (stat_partial) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) tw[(ct * 32 + 8 * g + 4 * h + j) * 33 + r] = valid ? v[j] : 0.f;
                    }
                }
            if (stat_partial) {
                // the wave wrote its own [64][32] block; lane l sums row co = l (stride 33: conflict-free)
                const float* row = tw + lane * 33;
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    const float o = row[k];
                    a1s += o;
                    a2s += o * o;
                }
            }
        }
        if (stat_partial) {
            float* red = reinterpret_cast<float*>(lt) + 4 * 64 * 33;  // [wave][stat][64]
            red[(wv * 2 + 0) * 64 + lane] = a1s;
            red[(wv * 2 + 1) * 64 + lane] = a2s;
            __syncthreads();
            if (threadIdx.x < 128) {
                const int t = threadIdx.x;
                stat_partial[(int64_t)canon * 128 + t] = (red[t] + red[128 + t]) + (red[256 + t] + red[384 + t]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv_wgrad_bf16: persistent; grid = G workgroups, tiles of 8 x 32 pixels.
//   dW[tap][ci][co] += sum_px In[px + tap][ci] * dOut[px][co]
// MFMA per tap and 16-pixel K step: A[i=ci][k=px], B[k=px][j=co]; both operands are transposing LDS reads
// (ds_read_b64_tr_b16: 4 pixel rows x 16 channels per 16-lane group) of the [pixel][channel] bf16 images.
// Row strides of 64 B (mod 256) make the 4 rows x 2 groups of a half-wave cover all 64 banks once.
template <int CI, int KS>
__global__ void __launch_bounds__(256, 1)
    conv_wgrad_bf16_kernel(const float* __restrict__ in, const float* __restrict__ in_scale,
                           const float* __restrict__ in_shift, int in_relu, const float* __restrict__ dout,
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
    const float* inb = in + ci_off;
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

    BTile<CI, LH, LW> tr;
    BTile<64, TH, BTW> td;
    auto coords = [&](int t, int& b, int& y0, int& x0) {
        const int ty = t % tiles_y;
        const int rest = t / tiles_y;
        b = rest / tiles_x;
        y0 = ty * TH;
        x0 = (rest - b * tiles_x) * BTW;
    };
    const int t_begin = (int)((int64_t)ntiles * blockIdx.x / gridDim.x);
    const int t_end = (int)((int64_t)ntiles * (blockIdx.x + 1) / gridDim.x);
    int tile = t_begin;
    if (tile < t_end) {
        int b, y0, x0;
        coords(tile, b, y0, x0);
        btile_load<CI, LH, LW, HALO>(tr, inb, b, y0, x0, H, W, in_cs);
        btile_load<64, TH, BTW, 0>(td, dout, b, y0, x0, H, W, 64);
    }
    for (; tile < t_end; ++tile) {
        int b, y0, x0;
        coords(tile, b, y0, x0);
        __syncthreads();
        btile_store<CI, LH, LW, HALO, ROWA>(tr, scb, shb, in_relu, lin, b, y0, x0, H, W, in_cs);
        btile_store<64, TH, BTW, 0, ROWD>(td, nullptr, nullptr, 0, ldo, b, y0, x0, H, W, 64);
        __syncthreads();
        if (tile + 1 < t_end) {
            int nb, ny, nx;
            coords(tile + 1, nb, ny, nx);
            btile_load<CI, LH, LW, HALO>(tr, inb, nb, ny, nx, H, W, in_cs);
            btile_load<64, TH, BTW, 0>(td, dout, nb, ny, nx, H, W, 64);
        }
#pragma unroll 1
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int kstep = ksl * KSTEPS + kk;        // 16 pixels: row kstep>>1, columns (kstep&1)*16 ..
            const int row = kstep >> 1, col0 = (kstep & 1) * 16;
            const int px = col0 + 8 * h + tq;           // this lane's pixel for the first 4-row block (+4 for the second)
            // B operand: dOut[px][co]
            const char* bp = ldo + (row * BTW + px) * ROWD + b_col;
            const s16x4 b_lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bp));
            const s16x4 b_hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(bp + 4 * ROWD));
            bf16x8 bv;
            {
                union { s16x4 s[2]; bf16x8 v; } u;
                u.s[0] = b_lo; u.s[1] = b_hi;
                bv = u.v;
            }
            const char* ap0 = lin + (row * LW + px) * ROWA + a_col;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int ky = t / KS, kx = t - ky * KS;
                const char* ap = ap0 + (ky * LW + kx) * ROWA;
                union { s16x4 s[2]; bf16x8 v; } u;
                u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(ap));
                u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(ap + 4 * ROWA));
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(u.v, bv, acc[t], 0, 0, 0);
            }
        }
    }
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
template <int CI, int KS, int TH>
static int launch_conv_fwd_bf16(const float* in, const __bf16* wp, const float* in_scale, const float* in_shift, int in_relu,
                                float* out, int out_cs, float* stat_partial, int B, int H, int W, int m_blocks,
                                hipStream_t stream) {
    constexpr int HALO = KS / 2;
    constexpr int LH = TH + 2 * HALO, LW = BTW + 2 * HALO;
    size_t tile_b = (size_t)LH * LW * (CI * 2 + 16);
    const size_t stat_b = (4 * 64 * 33 + 4 * 2 * 64) * sizeof(float);
    if (tile_b < stat_b) tile_b = stat_b;
    const size_t smem = (size_t)KS * KS * CI * 64 * 2 + tile_b;
    auto kern = conv_fwd_bf16_kernel<CI, KS, TH>;
    static bool attr_set = false;
    if (!attr_set) {
        P4C_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const int tiles_x = (W + BTW - 1) / BTW, tiles_y = (H + TH - 1) / TH;
    int64_t ntiles = (int64_t)tiles_x * tiles_y * B;
    int G = num_cus();
    if (ntiles < G) G = (int)ntiles;
    const int tag = (CI == 64 && KS == 3 && m_blocks == 1) ? P4C_PROF_CONV3X3_C64 : 0;
    if (tag) prof_begin(tag, (int64_t)B * H * W, stream);
    hipLaunchKernelGGL(kern, dim3(G, m_blocks), dim3(256), smem, stream, in, wp, in_scale, in_shift, in_relu, out, out_cs,
                       stat_partial, B, H, W);
    if (tag) prof_end(tag, stream);
    P4C_CHECK_LAUNCH("conv_fwd_bf16");
    return P4C_OK;
}

template <int CI, int KS>
static int launch_conv_wgrad_bf16(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                                  const float* dout, float* partial, int G, int B, int H, int W, int in_cs, int ci_off,
                                  int part_cip, hipStream_t stream) {
    constexpr int HALO = KS / 2;
    constexpr int LH = 8 + 2 * HALO, LW = BTW + 2 * HALO;
    constexpr int ROWA = (CI * 2) % 256 == 64 ? CI * 2 : CI * 2 + 64;
    const size_t smem = (size_t)LH * LW * ROWA + (size_t)8 * BTW * 192;
    auto kern = conv_wgrad_bf16_kernel<CI, KS>;
    static bool attr_set = false;
    if (!attr_set) {
        P4C_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const int tag = (CI == 64 && KS == 3 && in_cs == 64) ? P4C_PROF_WGRAD3X3_C64 : 0;
    if (tag) prof_begin(tag, (int64_t)B * H * W, stream);
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), smem, stream, in, in_scale, in_shift, in_relu, dout, partial, B, H, W,
                       in_cs, ci_off, part_cip);
    if (tag) prof_end(tag, stream);
    P4C_CHECK_LAUNCH("conv_wgrad_bf16");
    return P4C_OK;
}

// tiles of the bf16 forward kernel (statistics partial buffers are indexed by tile)
int conv_bf16_tile_h(int CI) { return CI <= 64 ? 8 : 4; }

int conv_fwd_bf16(const float* in, int CI, const void* wp, int ks, const float* in_scale, const float* in_shift,
                  int in_relu, float* out, int out_cs, float* stat_partial, int B, int H, int W, int m_blocks,
                  hipStream_t stream) {
    const __bf16* w = (const __bf16*)wp;
#define P4C_CASE(ci, k, th)                                                                                          \
    if (CI == ci && ks == k)                                                                                         \
        return launch_conv_fwd_bf16<ci, k, th>(in, w, in_scale, in_shift, in_relu, out, out_cs, stat_partial, B, H, W, \
                                               m_blocks, stream);
    P4C_CASE(32, 3, 8) P4C_CASE(64, 3, 8) P4C_CASE(96, 3, 4) P4C_CASE(32, 1, 8) P4C_CASE(64, 1, 8) P4C_CASE(96, 1, 4)
#undef P4C_CASE
    return fail(P4C_ERR_UNSUPPORTED, "conv_fwd_bf16: unsupported (CI=%d, ks=%d)", CI, ks);
}

int wgrad_reduce(const float* partial, int nslots, int ks, int CI_pad, int ci_lo, int ci_hi, int CO, int CI, float* grad,
                 hipStream_t stream);

int conv_wgrad_bf16(const float* in, int CI, int ks, const float* in_scale, const float* in_shift, int in_relu,
                    const float* dout, float* partial, int G, int B, int H, int W, int CO, int CIreal, float* grad,
                    hipStream_t stream) {
    if (CI % 32 != 0 || CI <= 0 || CI > 256) return fail(P4C_ERR_UNSUPPORTED, "conv_wgrad_bf16: unsupported CI=%d", CI);
    if (ks != 1 && ks != 3) return fail(P4C_ERR_UNSUPPORTED, "conv_wgrad_bf16: unsupported ks=%d", ks);
    const int tiles = ((H + 7) / 8) * ((W + BTW - 1) / BTW) * B;
    if (tiles < G) G = tiles;
    for (int off = 0; off < CI;) {
        const int chunk = (CI - off >= 64) ? 64 : 32;
        int rc;
        if (chunk == 64 && ks == 3)
            rc = launch_conv_wgrad_bf16<64, 3>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        else if (chunk == 32 && ks == 3)
            rc = launch_conv_wgrad_bf16<32, 3>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        else if (chunk == 64)
            rc = launch_conv_wgrad_bf16<64, 1>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        else
            rc = launch_conv_wgrad_bf16<32, 1>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        if (rc != P4C_OK) return rc;
        rc = wgrad_reduce(partial, G * (chunk == 64 ? 1 : 2), ks, CI, off, off + chunk, CO, CIreal, grad, stream);
        if (rc != P4C_OK) return rc;
        off += chunk;
    }
    return P4C_OK;
}

int prep_weights_bf16(const float* w, int CO, int CI, int ks, int transpose_flip, int M_pad, int K_pad, void* out,
                      hipStream_t stream) {
    const int total = M_pad * K_pad * ks * ks;
    hipLaunchKernelGGL(prep_weights_bf16_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, CO, CI, ks * ks,
                       transpose_flip, M_pad, K_pad, (__bf16*)out);
    P4C_CHECK_LAUNCH("prep_weights_bf16");
    return P4C_OK;
}

}  // namespace p4c
