// fp32 convolution kernels of the HalfUNet path on gfx950 matrix cores:
//   conv_fwd_f32   : 3x3 / 1x1 "same" convolution, NHWC, implicit GEMM on v_mfma_f32_32x32x2_f32
//                    (exact fp32 products, fp32 accumulate), with the previous layer's
//                    normalisation + ReLU applied while the input tile is staged (the normalised
//                    activation is never materialised) and the per-channel sum / sum-of-squares of
//                    the output produced in the epilogue (BatchNorm / GroupNorm statistics).
//                    The same kernel evaluates the data gradient (weights flipped + transposed by
//                    prep_weights).
//   conv_wgrad_f32 : weight gradient, persistent workgroups accumulate dW[tap][ci][co] in registers
//                    over their share of pixel tiles (K = pixels), one partial per workgroup,
//                    reduced by wgrad_reduce (deterministic, no float atomics).
//
// Layout: activations (B,H,W,C) with C contiguous (the reference's NamedTensor layout, features
// last), C a multiple of 32.  GEMM orientation is "weights x pixels": A = W (M = output channel),
// B = input pixels (N = pixel), so an accumulator lane owns one pixel and 4 consecutive output
// channels per register quad -> 16-byte NHWC stores straight from registers.
//
// K ordering: an MFMA 32x32x2 step consumes 2 channels (one per lane half h).  Channels are taken
// in groups of 8: sub-step s of group q uses channel 8q + 4h + s, so each lane reads its 4 channels
// of a pixel as ONE 16-byte LDS read (ds_read_b128) and its 4 weights as one 16-byte global load.
#include "kernels.hpp"

namespace p4c {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TW = 32;  // tile width in pixels = one MFMA N-tile

// ---------------------------------------------------------------------------------------------
// prep_weights: canonical torch weight w[CO][CI][KS][KS] -> MFMA A-operand stream
//   out[mb][tap][q][h][m_l(64)][s(4)],  k = 8q + 4h + s,  m = 64 mb + m_l
//   transpose_flip = 0: out = w[m][k][tap]               (forward:  M = co, K = ci)
//   transpose_flip = 1: out = w[k][m][ntaps-1-tap]       (data grad: M = ci, K = co, taps flipped)
// zero-filled outside the real channel counts.
__global__ void prep_weights_kernel(const float* __restrict__ w, int CO, int CI, int ntaps, int transpose_flip,
                                    int M_pad, int K_pad, float* __restrict__ out) {
    const int total = M_pad * K_pad * ntaps;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int t = i;
        const int s = t & 3; t >>= 2;
        const int m_l = t & 63; t >>= 6;
        const int h = t & 1; t >>= 1;
        const int q = t % (K_pad / 8); t /= (K_pad / 8);
        const int tap = t % ntaps;
        const int mb = t / ntaps;
        const int k = 8 * q + 4 * h + s, m = 64 * mb + m_l;
        float v = 0.0f;
        if (!transpose_flip) {
            if (m < CO && k < CI) v = w[((int64_t)m * CI + k) * ntaps + tap];
        } else {
            if (k < CO && m < CI) v = w[((int64_t)k * CI + m) * ntaps + (ntaps - 1 - tap)];
        }
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Stage a (LH x LW) halo tile of the NHWC input into LDS, applying v = relu?(v*scale[c]+shift[c]).
// Rows of CI floats are padded to CIS = CI+4 so that the per-pixel 16-byte reads of one
// ds_read_b128 lane group fall on 16 distinct bank quads.
// Split in two phases (T14, "issue early / write late"): tile_load() issues every global load of the
// thread into registers without waiting; tile_store() transforms and writes them to LDS.  Callers put
// independent work (the previous tile's MFMA phase) between the two.
template <int CI, int LH, int LW>
struct TileRegs {
    static constexpr int C4 = CI / 4;
    static constexpr int TOTAL = LH * LW * C4;
    static constexpr int ITERS = (TOTAL + 255) / 256;
    f32x4 v[ITERS];
};

template <int CI, int LH, int LW, int HALO>
__device__ __forceinline__ void tile_load(TileRegs<CI, LH, LW>& t, const float* __restrict__ in, int b, int y0, int x0, int H,
                                          int W, int in_cs) {
    constexpr int C4 = CI / 4;
#pragma unroll
    for (int it = 0; it < TileRegs<CI, LH, LW>::ITERS; ++it) {
        const int idx = threadIdx.x + it * 256;
        const int pix = idx / C4, c4 = idx - pix * C4;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (idx < TileRegs<CI, LH, LW>::TOTAL && gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = *reinterpret_cast<const f32x4*>(in + (((int64_t)b * H + gy) * W + gx) * in_cs + 4 * c4);
        t.v[it] = v;
    }
}

template <int CI, int LH, int LW, int HALO>
__device__ __forceinline__ void tile_store(const TileRegs<CI, LH, LW>& t, const float* __restrict__ scale,
                                           const float* __restrict__ shift, int relu, float* lds, int b, int y0, int x0,
                                           int H, int W, int sc_cs) {
    constexpr int CIS = CI + 4;
    constexpr int C4 = CI / 4;
#pragma unroll
    for (int it = 0; it < TileRegs<CI, LH, LW>::ITERS; ++it) {
        const int idx = threadIdx.x + it * 256;
        if (idx >= TileRegs<CI, LH, LW>::TOTAL) break;
        const int pix = idx / C4, c4 = idx - pix * C4;
        const int ly = pix / LW, lx = pix - ly * LW;
        const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
        f32x4 v = t.v[it];
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {  // zero padding stays zero: it is applied AFTER norm+relu
            if (scale) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (int64_t)b * sc_cs + 4 * c4);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + (int64_t)b * sc_cs + 4 * c4);
                v = v * sc + sh;
            }
            if (relu) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
        }
        *reinterpret_cast<f32x4*>(lds + pix * CIS + 4 * c4) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// conv_fwd_f32: grid (tiles_x, tiles_y*B, M_pad/64), 256 threads; tile = TH x 32 pixels x 64 outputs.
// wave w owns rows [w*RW, (w+1)*RW) of the tile (RW = TH/4), i.e. RW pixel-tiles x 2 channel-tiles.
template <int CI, int KS, int TH>
__global__ void __launch_bounds__(256)
    conv_fwd_f32_kernel(const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ in_scale,
                        const float* __restrict__ in_shift, int in_relu, const float* __restrict__ bias,
                        float* __restrict__ out, int out_cs, float* __restrict__ stat_partial, int H, int W) {
    constexpr int HALO = KS / 2;
    constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
    constexpr int CIS = CI + 4;
    constexpr int RW = TH / 4;
    constexpr int NTAPS = KS * KS;
    constexpr int NQ = CI / 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tiles_y = (H + TH - 1) / TH;
    const int b = blockIdx.y / tiles_y, ty = blockIdx.y - b * tiles_y;
    const int y0 = ty * TH, x0 = blockIdx.x * TW;
    const int mb = blockIdx.z;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;

    {
        TileRegs<CI, LH, LW> tr;
        tile_load<CI, LH, LW, HALO>(tr, in, b, y0, x0, H, W, CI);
        tile_store<CI, LH, LW, HALO>(tr, in_scale, in_shift, in_relu, lds, b, y0, x0, H, W, CI);
    }
    __syncthreads();

    f32x16 acc[2][RW];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < RW; ++pt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][pt][i] = 0.f;

    // weight stream: [tap][q] groups are contiguous (512 floats each); the next group's two 16-byte loads are
    // issued before the current group's 8*RW MFMAs so that the L2 latency hides under the matrix pipe
    const float* wbase = wp + (int64_t)mb * NTAPS * CI * 64 + (h * 64 + r) * 4;
    constexpr int NJ = NTAPS * NQ;
    f32x4 a0 = *reinterpret_cast<const f32x4*>(wbase);
    f32x4 a1 = *reinterpret_cast<const f32x4*>(wbase + 128);
#pragma unroll 1
    for (int tap = 0; tap < NTAPS; ++tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
        const float* lt = lds + ((wv * RW + ky) * LW + (r + kx)) * CIS + 4 * h;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            int jn = tap * NQ + q + 1;
            jn = jn < NJ ? jn : NJ - 1;
            const f32x4 n0 = *reinterpret_cast<const f32x4*>(wbase + jn * 512);
            const f32x4 n1 = *reinterpret_cast<const f32x4*>(wbase + jn * 512 + 128);
            f32x4 bb[RW];
#pragma unroll
            for (int pt = 0; pt < RW; ++pt) bb[pt] = *reinterpret_cast<const f32x4*>(lt + pt * LW * CIS + 8 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int pt = 0; pt < RW; ++pt) {
                    acc[0][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s], bb[pt][s], acc[0][pt], 0, 0, 0);
                    acc[1][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s], bb[pt][s], acc[1][pt], 0, 0, 0);
                }
            a0 = n0;
            a1 = n1;
        }
    }

    // ---- epilogue: C[co][px]; lane = pixel r (+ half h), register i -> co = (i&3) + 8*(i>>2) + 4*h
    const int gx = x0 + r;
    if (stat_partial) __syncthreads();  // all waves are done reading the input tile: LDS is reused below
    float* tw = lds + wv * (64 * 33);   // per-wave [co][33] transpose buffer
#pragma unroll
    for (int pt = 0; pt < RW; ++pt) {
        const int gy = y0 + wv * RW + pt;
        const bool valid = (gy < H) && (gx < W);
        float* orow = out + (((int64_t)b * H + gy) * W + gx) * out_cs + mb * 64 + 4 * h;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]};
                if (bias) v += *reinterpret_cast<const f32x4*>(bias + mb * 64 + ct * 32 + 8 * g + 4 * h);
                if (valid) *reinterpret_cast<f32x4*>(orow + ct * 32 + 8 * g) = v;
                if (stat_partial) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = ct * 32 + 8 * g + 4 * h + j;
                        const float o = valid ? v[j] : 0.f;
                        tw[co * 33 + r] = o;
                    }
                }
            }
        }
    }
    if (stat_partial) {
        static_assert(RW == 1, "statistics epilogue assumes one pixel-tile per wave");
        // lane l now sums row co = l of its wave's [64][32] tile (stride 33: conflict-free), then 4 waves -> 1
        __syncthreads();
        float a1 = 0.f, a2 = 0.f;
        const float* row = tw + lane * 33;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float o = row[k];
            a1 += o;
            a2 += o * o;
        }
        float* red = lds + 4 * 64 * 33;  // [wave][stat][64]
        red[(wv * 2 + 0) * 64 + lane] = a1;
        red[(wv * 2 + 1) * 64 + lane] = a2;
        __syncthreads();
        if (threadIdx.x < 128) {
            const int t = threadIdx.x;
            const float v = (red[t] + red[128 + t]) + (red[256 + t] + red[384 + t]);
            const int64_t tile = ((int64_t)blockIdx.y) * gridDim.x + blockIdx.x;
            stat_partial[tile * 128 + t] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// conv_wgrad_f32: persistent; grid = G workgroups, each loops over pixel tiles (4 x 32).
//   dW[tap][ci][co] += sum_px In[px + tap][ci] * dOut[px][co]
// GEMM per tap: A[i=ci][k=px] (lane = ci, half h = pixel parity), B[k=px][j=co] (lane = co).
// A wave owns one (ci-tile, co-tile) unit = 9 accumulator tiles.  With fewer than 4 units (CI = 32) the
// pixel range of a tile is split between wave pairs (KSPLIT) and each pair writes its own partial.
// The next tile's global loads are issued before the current tile's MFMA phase and written to LDS after it
// (register double buffering, T14): HBM latency hides under the matrix pipe although only one workgroup
// fits a CU.
template <int CI, int KS>
__global__ void __launch_bounds__(256, 1)
    conv_wgrad_f32_kernel(const float* __restrict__ in, const float* __restrict__ in_scale,
                          const float* __restrict__ in_shift, int in_relu, const float* __restrict__ dout,
                          float* __restrict__ partial, int B, int H, int W, int in_cs, int ci_off, int part_cip) {
    constexpr int TH = 4;
    constexpr int HALO = KS / 2;
    constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
    constexpr int CIS = CI + 4;
    constexpr int DS = 64 + 4;
    constexpr int NTAPS = KS * KS;
    constexpr int UNITS = (CI / 32) * 2;          // 2 or 4
    constexpr int KSPLIT = 4 / UNITS;             // wave groups splitting the pixel range
    constexpr int KSTEPS = TH * TW / 2 / KSPLIT;  // MFMA k-steps (pixel pairs) per wave and tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* lin = lds;
    float* ldo = lds + LH * LW * CIS;

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int unit = wv % UNITS, ksl = wv / UNITS;
    const int cit = unit >> 1, cot = unit & 1;
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * B;
    const float* inb = in + ci_off;
    const float* scb = in_scale ? in_scale + ci_off : nullptr;
    const float* shb = in_shift ? in_shift + ci_off : nullptr;

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    TileRegs<CI, LH, LW> tr;
    f32x4 dreg[TH * TW * 16 / 256];
    auto load_tile = [&](int tile) {
        const int tx = tile % tiles_x;
        const int rest = tile / tiles_x;
        const int ty = rest % tiles_y, b = rest / tiles_y;
        tile_load<CI, LH, LW, HALO>(tr, inb, b, ty * TH, tx * TW, H, W, in_cs);
#pragma unroll
        for (int it = 0; it < TH * TW * 16 / 256; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int pix = idx >> 4, c4 = idx & 15;
            const int gy = ty * TH + (pix >> 5), gx = tx * TW + (pix & 31);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gy < H && gx < W) v = *reinterpret_cast<const f32x4*>(dout + (((int64_t)b * H + gy) * W + gx) * 64 + 4 * c4);
            dreg[it] = v;
        }
    };
    auto store_tile = [&](int tile) {
        const int tx = tile % tiles_x;
        const int rest = tile / tiles_x;
        const int ty = rest % tiles_y, b = rest / tiles_y;
        tile_store<CI, LH, LW, HALO>(tr, scb, shb, in_relu, lin, b, ty * TH, tx * TW, H, W, in_cs);
#pragma unroll
        for (int it = 0; it < TH * TW * 16 / 256; ++it) {
            const int idx = threadIdx.x + it * 256;
            *reinterpret_cast<f32x4*>(ldo + (idx >> 4) * DS + 4 * (idx & 15)) = dreg[it];
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();  // previous tile fully consumed
        store_tile(tile);
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);  // in flight during the MFMA phase
#pragma unroll 2
        for (int kp = 0; kp < KSTEPS; ++kp) {
            const int p = 2 * (ksl * KSTEPS + kp) + h;  // this lane half's pixel of the K-step
            const int row = p >> 5, col = p & 31;
            const float bv = ldo[p * DS + cot * 32 + r];
            const float* ain = lin + (row * LW + col) * CIS + cit * 32 + r;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int ky = t / KS, kx = t - ky * KS;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ain[(ky * LW + kx) * CIS], bv, acc[t], 0, 0, 0);
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

// grad[co][ci][tap] += sum_slots partial[slot][tap][ci_pad][64]  for ci in [ci_lo, ci_hi), real channels only.
// One block per (tap, ci, co group): QL lanes x 4 output channels (16-byte loads) x 256/QL slot slices, all of a thread's loads
// independent; fixed summation order (deterministic).  QL = 16: one block covers the 64 output channels; QL = 4: four blocks do
// (the 1x1 convolution has 64 (tap, ci) pairs only -- 64 workgroups walking 256 slots each took 41 us at 2 x 512 x 512).
template <int QL>
__device__ __forceinline__ void wgrad_reduce_block(const float* __restrict__ partial, int nslots, int ntaps, int CI_pad, int ci_lo,
                                                   int ci_hi, int CO, int CI, float* __restrict__ grad, int block) {
    constexpr int NS = 256 / QL, NCO = 4 * QL, NG = 64 / NCO;
    __shared__ float red[NS][NCO];
    const int nci = ci_hi - ci_lo;
    const int cg = block % NG, rest = block / NG;
    const int tap = rest / nci, ci = ci_lo + (rest - tap * nci);
    const int q = threadIdx.x % QL, sl = threadIdx.x / QL;
    const int64_t stride = (int64_t)ntaps * CI_pad * 64;
    const float* p = partial + ((int64_t)tap * CI_pad + ci) * 64 + cg * NCO + 4 * q;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int g = sl; g < nslots; g += NS) s += *reinterpret_cast<const f32x4*>(p + g * stride);
#pragma unroll
    for (int j = 0; j < 4; ++j) red[sl][4 * q + j] = s[j];
    __syncthreads();
    if (threadIdx.x < NCO) {
        const int co = cg * NCO + threadIdx.x;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < NS; ++k) t += red[k][threadIdx.x];
        if (co < CO && ci < CI) grad[((int64_t)co * CI + ci) * ntaps + tap] += t;
    }
}

template <int QL>
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, int nslots, int ntaps, int CI_pad,
                                                           int ci_lo, int ci_hi, int CO, int CI, float* __restrict__ grad) {
    wgrad_reduce_block<QL>(partial, nslots, ntaps, CI_pad, ci_lo, ci_hi, CO, CI, grad, blockIdx.x);
}

// every recorded job in one launch: block -> (job, block of that job) through the prefix table in the kernel arguments; each job
// is summed exactly as a wgrad_reduce_kernel<16> launch of its own would (same slices, same order).  The standalone wgrad_reduce
// below takes the <4> form for short jobs with many slots (pairs < 256, nslots >= 64: the 1x1 output convolution fed by
// out_conv_bwd's partials) -- 64 slot slices instead of 16, another order of the same fp32 sum: that one gradient is NOT bitwise
// the same between the batched plan and P4C_WGRAD_BATCH=0 / p4c_out_conv_bwd on its own (both fixed orders, both deterministic)
struct WgradBatchArgs {
    WgradReduceJob job[WGRAD_BATCH_MAX];
    int first[WGRAD_BATCH_MAX + 1];
    int n;
};
__global__ void __launch_bounds__(256) wgrad_reduce_batch_kernel(WgradBatchArgs a) {
    int j = 0;
    while (j + 1 < a.n && (int)blockIdx.x >= a.first[j + 1]) ++j;
    const WgradReduceJob& J = a.job[j];
    wgrad_reduce_block<16>(J.partial, J.nslots, J.ntaps, J.CI_pad, J.ci_lo, J.ci_hi, J.CO, J.CI, J.grad, (int)blockIdx.x - a.first[j]);
}

template <int CI, int KS, int TH>
static int launch_conv_fwd(const float* in, const float* wp, const float* in_scale, const float* in_shift, int in_relu,
                           const float* bias, float* out, int out_cs, float* stat_partial, int B, int H, int W,
                           int m_blocks, hipStream_t stream) {
    constexpr int HALO = KS / 2;
    constexpr int LH = TH + 2 * HALO, LW = TW + 2 * HALO;
    size_t smem = (size_t)LH * LW * (CI + 4) * sizeof(float);
    const size_t stat_smem = (4 * 64 * 33 + 4 * 2 * 64) * sizeof(float);  // statistics epilogue scratch
    if (smem < stat_smem) smem = stat_smem;
    auto kern = conv_fwd_f32_kernel<CI, KS, TH>;
    P4C_TRY(ensure_dyn_smem((const void*)kern, (int)smem));
    const int tiles_x = (W + TW - 1) / TW, tiles_y = (H + TH - 1) / TH;
    const int tag = (CI == 64 && KS == 3 && m_blocks == 1) ? P4C_PROF_CONV3X3_C64 : 0;
    if (tag) prof_begin(tag, (int64_t)B * H * W, stream);
    hipLaunchKernelGGL(kern, dim3(tiles_x, tiles_y * B, m_blocks), dim3(256), smem, stream, in, wp, in_scale, in_shift,
                       in_relu, bias, out, out_cs, stat_partial, H, W);
    if (tag) prof_end(tag, stream);
    P4C_CHECK_LAUNCH("conv_fwd_f32");
    return P4C_OK;
}

template <int CI, int KS>
static int launch_conv_wgrad(const float* in, const float* in_scale, const float* in_shift, int in_relu,
                             const float* dout, float* partial, int G, int B, int H, int W, int in_cs, int ci_off,
                             int part_cip, hipStream_t stream) {
    constexpr int HALO = KS / 2;
    constexpr int LH = 4 + 2 * HALO, LW = TW + 2 * HALO;
    const size_t smem = ((size_t)LH * LW * (CI + 4) + (size_t)4 * TW * 68) * sizeof(float);
    auto kern = conv_wgrad_f32_kernel<CI, KS>;
    P4C_TRY(ensure_dyn_smem((const void*)kern, (int)smem));
    const int tag = (CI == 64 && KS == 3 && in_cs == 64) ? P4C_PROF_WGRAD3X3_C64 : 0;
    if (tag) prof_begin(tag, (int64_t)B * H * W, stream);
    hipLaunchKernelGGL(kern, dim3(G), dim3(256), smem, stream, in, in_scale, in_shift, in_relu, dout, partial, B, H, W,
                       in_cs, ci_off, part_cip);
    if (tag) prof_end(tag, stream);
    P4C_CHECK_LAUNCH("conv_wgrad_f32");
    return P4C_OK;
}

// ---------------------------------------------------------------------------------------------
// host-side dispatchers used by the C ABI and by the HalfUNet plan
int conv_fwd_f32(const float* in, int CI, const float* wp, int ks, const float* in_scale, const float* in_shift,
                 int in_relu, const float* bias, float* out, int out_cs, float* stat_partial, int B, int H, int W,
                 int m_blocks, hipStream_t stream) {
#define P4C_CASE(ci, k)                                                                                              \
    if (CI == ci && ks == k)                                                                                         \
        return launch_conv_fwd<ci, k, 4>(in, wp, in_scale, in_shift, in_relu, bias, out, out_cs, stat_partial, B, H, \
                                         W, m_blocks, stream);
    P4C_CASE(32, 3) P4C_CASE(64, 3) P4C_CASE(96, 3) P4C_CASE(32, 1) P4C_CASE(64, 1) P4C_CASE(96, 1)
#undef P4C_CASE
    return fail(P4C_ERR_UNSUPPORTED, "conv_fwd_f32: unsupported (CI=%d, ks=%d): CI must be 32/64/96, ks 1/3", CI, ks);
}

static thread_local WgradCollect* g_wgrad_collect = nullptr;
void wgrad_collect_into(WgradCollect* c) { g_wgrad_collect = c; }

int wgrad_reduce_batch(const WgradCollect& c, hipStream_t stream) {
    if (c.n == 0) return P4C_OK;
    WgradBatchArgs a;
    a.n = c.n;
    int blocks = 0;
    for (int j = 0; j < c.n; ++j) {
        a.job[j] = c.job[j];
        a.first[j] = blocks;
        blocks += c.job[j].ntaps * (c.job[j].ci_hi - c.job[j].ci_lo);
    }
    a.first[c.n] = blocks;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks), dim3(256), 0, stream, a);
    P4C_CHECK_LAUNCH("wgrad_reduce_batch");
    return P4C_OK;
}

int wgrad_reduce(const float* partial, int nslots, int ks, int CI_pad, int ci_lo, int ci_hi, int CO, int CI, float* grad,
                        hipStream_t stream) {
    if (ci_hi > CI) ci_hi = CI;
    if (ci_hi <= ci_lo) return P4C_OK;
    if (WgradCollect* c = g_wgrad_collect) {
        if (c->n == WGRAD_BATCH_MAX) return fail(P4C_ERR_INVALID, "wgrad_reduce: more than %d deferred reductions in one batch", WGRAD_BATCH_MAX);
        c->job[c->n++] = WgradReduceJob{partial, grad, nslots, ks * ks, CI_pad, ci_lo, ci_hi, CO, CI};
        return P4C_OK;
    }
    const int pairs = ks * ks * (ci_hi - ci_lo);
    if (pairs < 256 && nslots >= 64)
        hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3(pairs * 4), dim3(256), 0, stream, partial, nslots, ks * ks, CI_pad, ci_lo,
                           ci_hi, CO, CI, grad);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<16>, dim3(pairs), dim3(256), 0, stream, partial, nslots, ks * ks, CI_pad, ci_lo, ci_hi,
                           CO, CI, grad);
    P4C_CHECK_LAUNCH("wgrad_reduce");
    return P4C_OK;
}

// grad[CO][CIreal][ks][ks] += weight gradient.  partial: wgrad_partial_floats(CI, ks, G) floats of scratch.
int conv_wgrad_f32(const float* in, int CI, int ks, const float* in_scale, const float* in_shift, int in_relu,
                   const float* dout, float* partial, int G, int B, int H, int W, int CO, int CIreal, float* grad,
                   hipStream_t stream) {
    // input channels are processed in chunks of 64 (+32): the 9 accumulator tiles per (ci,co) unit of a
    // 96-channel chunk would not fit the register file without spilling.
    if (CI % 32 != 0 || CI <= 0 || CI > 256) return fail(P4C_ERR_UNSUPPORTED, "conv_wgrad_f32: unsupported CI=%d", CI);
    if (ks != 1 && ks != 3) return fail(P4C_ERR_UNSUPPORTED, "conv_wgrad_f32: unsupported ks=%d", ks);
    for (int off = 0; off < CI;) {
        const int chunk = (CI - off >= 64) ? 64 : 32;
        int rc;
        if (chunk == 64 && ks == 3)
            rc = launch_conv_wgrad<64, 3>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        else if (chunk == 32 && ks == 3)
            rc = launch_conv_wgrad<32, 3>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        else if (chunk == 64)
            rc = launch_conv_wgrad<64, 1>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        else
            rc = launch_conv_wgrad<32, 1>(in, in_scale, in_shift, in_relu, dout, partial, G, B, H, W, CI, off, CI, stream);
        if (rc != P4C_OK) return rc;
        const int nslots = G * (chunk == 64 ? 1 : 2);
        rc = wgrad_reduce(partial, nslots, ks, CI, off, off + chunk, CO, CIreal, grad, stream);
        if (rc != P4C_OK) return rc;
        off += chunk;
    }
    return P4C_OK;
}

int prep_weights(const float* w, int CO, int CI, int ks, int transpose_flip, int M_pad, int K_pad, float* out,
                 hipStream_t stream) {
    const int total = M_pad * K_pad * ks * ks;
    hipLaunchKernelGGL(prep_weights_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w, CO, CI, ks * ks,
                       transpose_flip, M_pad, K_pad, out);
    P4C_CHECK_LAUNCH("prep_weights");
    return P4C_OK;
}

}  // namespace p4c
