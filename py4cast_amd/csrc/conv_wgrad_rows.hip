// conv3x3_wgrad_bf16_rows: weight gradient of the 3x3 convolution 64 -> 64 channels on bf16 NHWC maps (what cuDNN / MIOpen's wgrad
// does for mfai's HalfUNet under py4cast/lightning.py:591-596, differentiated) -- the ROW-STREAMING form (round 4), successor of the
// tile kernel conv3x3_wgrad_bf16_ws (conv_bf16.hip) wherever the map is at least 64 pixels wide:
//
//   dW[ky][kx][ci][co] = sum over pixels (y, x) of  X[y + ky - 1][x + kx - 1][ci] * dY[y][x][co]          (K = pixels)
//
//   * a workgroup owns a 64-pixel-wide STRIP SEGMENT of one sample (rows y0 .. y0+R-1 of dY, rows y0-1 .. y0+R of X) and walks
//     DOWN it, as conv3x3_bf16_rows does: every X row and every dY row is staged in LDS ONCE per workgroup (the tile kernel staged
//     6 x 34 input pixels per 4 x 32 output pixels: 1.59 x), halo = one column each side and one row at each end of a segment;
//   * 512 threads: waves 4-7 are the memory side (global -> registers -> normalise / ReLU of X, pass 2 of the normalisation
//     backward for dY (NormBwdCoef) -> LDS rings of 8 rows, four rows per interval, ONE workgroup barrier per four rows);
//     waves 0-3 the matrix side, wave = (32 ci x 32 co) x 9 taps = 9 accumulator tiles (144 registers);
//   * the K loop walks X rows: X row m meets dY rows m, m-1, m-2 (ky = 0, 1, 2).  Per 16-pixel K step the three column-shifted
//     X operands are read from LDS (6 transposed reads) and the step's operand of the NEW dY row (2 reads); the operands of
//     the two older dY rows stay in registers (3 rows x 4 steps x 4 registers): 32 ds_read_b64_tr_b16 per 36 MFMAs (0.89 per
//     MFMA; the tile kernel: 1.2), every read `row base + lane constant + immediate`;
//   * LDS pixels are 128 bytes (no padding): the two 64-byte channel halves of a pixel are swapped when bit 1 of its LDS column is
//     set, so the four pixels of a transposed read's 32-lane group fall in four different bank quarters for every column shift.
// Output: one fp32 partial [9][64][64] per workgroup, reduced in a fixed order by wgrad_reduce / wgrad_reduce_batch (conv_f32.hip).
// LDS: X ring 8 x 66 x 128 B = 66 KB | dY ring 8 x 64 x 128 B = 64 KB.
#include <stdlib.h>

#include "kernels.hpp"

namespace p4c {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace wr {
constexpr int SW = 64;             // strip width (pixels of dY)
constexpr int LW = SW + 2;         // X columns of a strip
constexpr int NR = 8;              // ring rows: 4 being read, 4 being staged
constexpr int PIXB = 128;
constexpr int XROW = LW * PIXB, DROW = SW * PIXB;
constexpr int XRING = NR * XROW, DRING = NR * DROW;
constexpr int SMEM = XRING + DRING;
constexpr int XSLOTS = LW * 8;     // 16-byte slots per X row
constexpr int NLX = 9;             // X slots per loader lane and interval: 4 x 528 = 2112 <= 9 x 256
constexpr int NLD = 8;             // dY slots: 4 x 512 = 8 x 256
constexpr int MIN_ROWS = 8;        // rows per segment at least (host)
// byte offset of channel octet c8 of LDS pixel column col
__device__ __forceinline__ int slot_off(int col, int c8) { return col * PIXB + ((((c8 >> 2) ^ (col >> 1)) & 1) << 6) + ((c8 & 3) << 4); }
}  // namespace wr

constexpr int OOB = 0x7fffffff;

// relu(v * scale + shift) on the 2 bf16 channels of a word.  MODE 0: copy, 1: ReLU, 2: scale / shift + ReLU, 3: scale / shift
template <int MODE>
__device__ __forceinline__ unsigned int xform2(unsigned int w, f32x2 sc, f32x2 sh) {
    if (MODE >= 2) {
        const float lo = __builtin_fmaf(__builtin_bit_cast(float, w << 16), sc.x, sh.x);
        const float hi = __builtin_fmaf(__builtin_bit_cast(float, w & 0xffff0000u), sc.y, sh.y);
        const f32x2 v = {lo, hi};
        w = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
    }
    if (MODE == 1 || MODE == 2) {
        const s16x2 z = {0, 0};
        w = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
    }
    return w;
}

// dY = alpha * g + beta * y + delta, g = dA where the forward ReLU was alive (kernels.hpp: NormBwdCoef), 2 channels per word
__device__ __forceinline__ unsigned int nb2(unsigned int a2, unsigned int y2, f32x2 al, f32x2 be, f32x2 de, f32x2 sc, f32x2 sh) {
    const float ylo = __builtin_bit_cast(float, y2 << 16), yhi = __builtin_bit_cast(float, y2 & 0xffff0000u);
    const float glo = __builtin_fmaf(ylo, sc.x, sh.x) > 0.f ? __builtin_bit_cast(float, a2 << 16) : 0.f;
    const float ghi = __builtin_fmaf(yhi, sc.y, sh.y) > 0.f ? __builtin_bit_cast(float, a2 & 0xffff0000u) : 0.f;
    const float lo = __builtin_fmaf(al.x, glo, __builtin_fmaf(be.x, ylo, de.x));
    const float hi = __builtin_fmaf(al.y, ghi, __builtin_fmaf(be.y, yhi, de.y));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

// x: (B,H,W,in_cs) with the chunk's channels at ci_off .. ci_off + 8 n_oct - 1 (n_oct <= 8 octets: absent ones are staged as zeros);
// partial: [workgroup][9][part_cip][64], the chunk writes channels ci_off .. of it (the 96-channel first convolution runs as a full
// chunk and a thin one into the same buffer)
struct WgRowsArgs {
    const __bf16* x; const float* x_scale; const float* x_shift; const __bf16* dout; float* partial;
    int H, W, rows_lo, rows_rem, in_cs, ci_off, n_oct, part_cip;
    NormBwdCoef nb;
};

struct BRow { s16x4 v[4][2]; };   // the four K-step operands of one dY row (two transposed reads each)

// One X row of the matrix phase.  bn / bm / bo: operands of dY rows m / m-1 / m-2.  On entry the step-0 operands of this row are in
// flight in fa[0] (and bn.v[0]); while step s runs its MFMAs the operands of step s+1 are read -- for s = 3 those of step 0 of the
// NEXT row (`pf`: same interval), whose new dY row lands in bo.v[0] (`pfb`: the next row has one): bo is the next row's bn.
template <bool H0, bool H1, bool H2>
__device__ __forceinline__ void wg_row(f32x16 (&acc)[9], BRow& bn, BRow& bm, BRow& bo, s16x4 (&fa)[2][3][2], const char* xr,
                                       const char* dr, const char* xr_next, const char* dr_next, bool pf, bool pfb,
                                       const int (&xoff)[3], int doff, bool active) {
    if (!active) return;   // (a thin chunk: this wave's 32 input channels do not exist -- wave-uniform)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bool last = s == 3;
        const char* xs = last ? xr_next : xr + (s + 1) * 2048;
        const char* ds = last ? dr_next : dr + (s + 1) * 2048;
        const int nx = (s + 1) & 1;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            // one transposed read per MFMA gap: operands of the next step
            if (j < 6) {
                if (!last || pf) fa[nx][j >> 1][j & 1] = tr_read(xs + xoff[j >> 1] + (j & 1) * 512);
            } else if (j < 8) {
                if (last) {
                    if (pfb) bo.v[0][j & 1] = tr_read(ds + doff + (j & 1) * 512);
                } else if (H0) {
                    bn.v[s + 1][j & 1] = tr_read(ds + doff + (j & 1) * 512);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int ky = j / 3, kx = j - 3 * ky;
            const bool on = ky == 0 ? H0 : (ky == 1 ? H1 : H2);
            if (on) {
                const BRow& b = ky == 0 ? bn : (ky == 1 ? bm : bo);
                union { s16x4 q[2]; bf16x8 v; } ua, ub;
                ua.q[0] = fa[s & 1][kx][0]; ua.q[1] = fa[s & 1][kx][1];
                ub.q[0] = b.v[s][0]; ub.q[1] = b.v[s][1];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int MODE, bool NB>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
    conv3x3_wgrad_bf16_rows_kernel(WgRowsArgs args) {
    using namespace wr;
    const __bf16* __restrict__ x = args.x;
    const __bf16* __restrict__ dout = args.dout;
    const int H = args.H, W = args.W, rows_lo = args.rows_lo, rows_rem = args.rows_rem;
    const NormBwdCoef& nb = args.nb;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xring = smem;
    char* dring = smem + XRING;

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int seg = blockIdx.x, strip = blockIdx.y, b = blockIdx.z;
    const int nseg = gridDim.x, nstrips = gridDim.y;
    const int y0 = seg * rows_lo + (seg < rows_rem ? seg : rows_rem);
    const int R = rows_lo + (seg < rows_rem ? 1 : 0);          // >= MIN_ROWS (host)
    const int x0 = strip * SW;
    const int K = (R + 5) >> 2;                                // intervals of four X rows: ceil((R + 2) / 4)

    if (wv >= 4) {
        // ------------------------------------------------------------ memory side
        const int ltid = threadIdx.x - 256, c8 = ltid & 7;
        const int in_cs = args.in_cs, xpb = in_cs * 2;   // bytes per pixel of x
        const bool in_ch = c8 < args.n_oct;               // this lane's channel octet exists in the chunk
        const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x + (int64_t)b * H * W * in_cs + args.ci_off,
                                                      (unsigned int)(((int64_t)H * W * in_cs - args.ci_off) * 2));
        const __amdgpu_buffer_rsrc_t rs_d = make_rsrc(dout + (int64_t)b * H * W * 64, (unsigned int)H * W * 128u);
        const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(NB ? reinterpret_cast<const __bf16*>(nb.y) + (int64_t)b * H * W * 64 : dout,
                                                      (unsigned int)H * W * 128u);
        int gx[NLX], lx[NLX], limx[NLX];
        int rr = 0, rem = ltid;
#pragma unroll
        for (int it = 0; it < NLX; ++it) {
            const int idx = ltid + it * 256;
            if (it > 0) { rem += 256; if (rem >= XSLOTS) { rem -= XSLOTS; ++rr; } }
            const int col = rem >> 3;
            gx[it] = in_ch ? (rr * W + col) * xpb + 16 * c8 : OOB;   // (an absent octet loads as zeros and is staged as such)
            lx[it] = rr * XROW + slot_off(col, c8);
            limx[it] = ((unsigned)(x0 - 1 + col) < (unsigned)W) ? idx : OOB;   // columns outside the image: never loaded, zeroed below
        }
        {
            const int left = x0 == 0 ? 1 : 0, right = x0 + SW >= W ? 1 : 0;   // (W is a multiple of 64: host)
            for (int rz = 0; rz < NR; ++rz)
                for (int j = ltid; j < (left + right) * 8; j += 256) {
                    const int col = (left && j < 8) ? 0 : LW - 1;
                    *reinterpret_cast<u32x4*>(xring + rz * XROW + col * PIXB + 16 * (j & 7)) = u32x4{0u, 0u, 0u, 0u};
                }
        }
        int gd[NLD], ld[NLD];
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int px = (ltid + j * 256) >> 3, row = px >> 6, col = px & 63;
            gd[j] = (row * W + col) * 128 + 16 * c8;
            ld[j] = row * DROW + slot_off(col, c8);
        }
        f32x2 sc[4], sh[4], al[4], be[4], dl[4], nsc[4], nsh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sc[k] = sh[k] = al[k] = be[k] = dl[k] = nsc[k] = nsh[k] = f32x2{0.f, 0.f};
            if (MODE >= 2 && in_ch) {
                sc[k] = *reinterpret_cast<const f32x2*>(args.x_scale + (int64_t)b * in_cs + args.ci_off + 8 * c8 + 2 * k);
                sh[k] = *reinterpret_cast<const f32x2*>(args.x_shift + (int64_t)b * in_cs + args.ci_off + 8 * c8 + 2 * k);
            }
            if (NB) {
                const int ch = 8 * c8 + 2 * k;
                const f32x2 ga = *reinterpret_cast<const f32x2*>(nb.gamma + ch), rs = *reinterpret_cast<const f32x2*>(nb.rstd + b * 64 + ch);
                const f32x2 mu = *reinterpret_cast<const f32x2*>(nb.mean + b * 64 + ch);
                const f32x2 q1 = *reinterpret_cast<const f32x2*>(nb.k1 + b * 64 + ch), q2 = *reinterpret_cast<const f32x2*>(nb.k2 + b * 64 + ch);
                nsc[k] = *reinterpret_cast<const f32x2*>(nb.scale + b * 64 + ch);
                nsh[k] = *reinterpret_cast<const f32x2*>(nb.shift + b * 64 + ch);
                al[k] = rs * ga;
                be[k] = -(rs * rs) * q2;
                dl[k] = rs * rs * q2 * mu - rs * q1;
            }
        }
        struct Img { u32x4 a[NLX]; u32x4 d[NLD]; u32x4 y[NB ? NLD : 1]; };
        Img ta;
        // interval k: X rows m = 4k .. 4k+3 (image rows y0 - 1 + m, m <= R + 1), dY rows n = 4k .. 4k+3 (image rows y0 + n, n < R)
        auto load = [&](Img& im, int k) __attribute__((always_inline)) {
            const int m0 = 4 * k;
            int nxr = R + 2 - m0;
            nxr = nxr > 4 ? 4 : (nxr < 0 ? 0 : nxr);
            int ndr = R - m0;
            ndr = ndr > 4 ? 4 : (ndr < 0 ? 0 : ndr);
            const int nactx = nxr * XSLOTS, nactd = ndr * 512;
            const int gy0 = y0 - 1 + m0, gx0 = x0 - 1;
            if (gy0 >= 0 && gy0 + nxr <= H) {
                const int so = (gy0 * W + gx0) * xpb;
#pragma unroll
                for (int it = 0; it < NLX; ++it)
                    im.a[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (limx[it] < nactx) ? gx[it] : OOB, so, 0);
            } else {
#pragma unroll
                for (int it = 0; it < NLX; ++it) {
                    const int idx = ltid + it * 256, r4 = idx / XSLOTS;
                    const int gy = gy0 + r4, gxx = gx0 + ((idx - r4 * XSLOTS) >> 3);
                    const bool ok = (idx < nactx) & ((unsigned)gy < (unsigned)H) & ((unsigned)gxx < (unsigned)W) & in_ch;
                    im.a[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (gy * W + gxx) * xpb + 16 * c8 : OOB, 0, 0);
                }
            }
            const int sd = ((y0 + m0) * W + x0) * 128;   // (dY rows of a segment are always inside the image)
#pragma unroll
            for (int j = 0; j < NLD; ++j) im.d[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_d, (ltid + j * 256 < nactd) ? gd[j] : OOB, sd, 0);
            if (NB) {
#pragma unroll
                for (int j = 0; j < NLD; ++j) im.y[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (ltid + j * 256 < nactd) ? gd[j] : OOB, sd, 0);
            }
        };
        auto store = [&](const Img& im, int k) __attribute__((always_inline)) {
            const int m0 = 4 * k;
            int nxr = R + 2 - m0;
            nxr = nxr > 4 ? 4 : (nxr < 0 ? 0 : nxr);
            int ndr = R - m0;
            ndr = ndr > 4 ? 4 : (ndr < 0 ? 0 : ndr);
            const int nactx = nxr * XSLOTS, nactd = ndr * 512;
            char* dstx = xring + (m0 & 7) * XROW;
            char* dstd = dring + (m0 & 7) * DROW;
            const int gy0 = y0 - 1 + m0;
            const bool interior = gy0 >= 0 && gy0 + nxr <= H;
#pragma unroll
            for (int it = 0; it < NLX; ++it) {
                u32x4 o = im.a[it];
                if (MODE != 0) {
                    unsigned int keep = 0xffffffffu;
                    if (!interior) {   // zero padding applies to the NORMALISED activation (rows outside the image; columns: limx)
                        const int idx = ltid + it * 256, r4 = idx / XSLOTS;
                        keep = ((unsigned)(gy0 + r4) < (unsigned)H) ? 0xffffffffu : 0u;
                    }
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) o[k2] = xform2<MODE>(o[k2], sc[k2], sh[k2]) & keep;
                }
                if (limx[it] < nactx) *reinterpret_cast<u32x4*>(dstx + lx[it]) = o;
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                u32x4 o = im.d[j];
                if (NB) {
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) o[k2] = nb2(o[k2], im.y[j][k2], al[k2], be[k2], dl[k2], nsc[k2], nsh[k2]);
                }
                if (ltid + j * 256 < nactd) *reinterpret_cast<u32x4*>(dstd + ld[j]) = o;
            }
        };
        load(ta, 0);
        store(ta, 0);
        load(ta, 1);
        lds_barrier();
        for (int k = 0; k < K; ++k) {
            store(ta, k + 1);
            load(ta, k + 2);
            lds_barrier();
        }
        return;
    }

    // ---------------------------------------------------------------- matrix side: wave = (32 ci x 32 co) x 9 taps
    const int h = lane >> 5, r = lane & 31;
    const int cit = wv >> 1, cot = wv & 1;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int cb = (tg * 16 + tp * 4) * 2;
    int xoff[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xoff[kx] = (8 * h + tq + kx) * PIXB + (((cit ^ ((tq + kx) >> 1)) & 1) << 6) + cb;
    const int doff = (8 * h + tq) * PIXB + (((cot ^ (tq >> 1)) & 1) << 6) + cb;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const bool active = cit * 4 < args.n_oct;   // this wave's 32 input channels exist (a thin chunk keeps waves 2, 3 idle)
    BRow s0, s1, s2;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) s0.v[s][q] = s1.v[s][q] = s2.v[s][q] = s16x4{0, 0, 0, 0};
    s16x4 fa[2][3][2];
    const int last = R + 1;
    int m = 0;
    auto xrow = [&](int mm) __attribute__((always_inline)) { return (const char*)xring + (mm & 7) * XROW; };
    auto drow = [&](int mm) __attribute__((always_inline)) { return (const char*)dring + (mm & 7) * DROW; };
    lds_barrier();   // the first interval's rows are staged
#pragma unroll
    for (int j = 0; j < 6; ++j) fa[0][j >> 1][j & 1] = tr_read(xrow(0) + xoff[j >> 1] + (j & 1) * 512);   // (idle waves: harmless reads)
#pragma unroll
    for (int j = 0; j < 2; ++j) s0.v[0][j] = tr_read(drow(0) + doff + j * 512);
    // BN / BM / BO: the dY operand sets of rows m / m-1 / m-2; BO receives step 0 of row m+1 (it is the next row's BN)
#define P4C_WROW(H0, H1, H2, BN, BM, BO)                                                                                  \
    {                                                                                                                     \
        const bool pf = ((m & 3) != 3) && (m != last);                                                                    \
        const bool nextb = m + 1 < R;                                                                                     \
        wg_row<H0, H1, H2>(acc, BN, BM, BO, fa, xrow(m), drow(m), xrow(m + 1), drow(m + 1), pf, pf && nextb, xoff, doff, active); \
        if (m == last) {                                                                                                  \
            lds_barrier();                                                                                                \
        } else if ((m & 3) == 3) {                                                                                        \
            lds_barrier();                                                                                                \
            _Pragma("unroll") for (int j = 0; j < 6; ++j) fa[0][j >> 1][j & 1] = tr_read(xrow(m + 1) + xoff[j >> 1] + (j & 1) * 512); \
            if (nextb) {                                                                                                  \
                _Pragma("unroll") for (int j = 0; j < 2; ++j) BO.v[0][j] = tr_read(drow(m + 1) + doff + j * 512);         \
            }                                                                                                             \
        }                                                                                                                 \
        ++m;                                                                                                              \
    }
    P4C_WROW(true, false, false, s0, s2, s1)     // X row 0 (image row y0 - 1): tap row 0 of dY row 0
    P4C_WROW(true, true, false, s1, s0, s2)      // X row 1
    while (m + 3 <= R) {
        P4C_WROW(true, true, true, s2, s1, s0)
        P4C_WROW(true, true, true, s0, s2, s1)
        P4C_WROW(true, true, true, s1, s0, s2)
    }
    // at most two left-over rows with a new dY row, then the two closing X rows: operand sets rotate by register moves (once per
    // workgroup) so that the names stay (s2, s1, s0)
#define P4C_WROTATE() { const BRow t_ = s0; s0 = s1; s1 = s2; s2 = t_; }
    while (m < R) {
        P4C_WROW(true, true, true, s2, s1, s0)
        P4C_WROTATE()
    }
    P4C_WROW(false, true, true, s2, s1, s0)      // X row R: tap rows 1 / 2 of the last two dY rows
    P4C_WROTATE()
    P4C_WROW(false, false, true, s2, s1, s0)     // X row R + 1
#undef P4C_WROTATE
#undef P4C_WROW
    // C[ci][co]: lane = co (r), register i -> ci = (i & 3) + 8 (i >> 2) + 4 h.  One partial per workgroup.
    const int wg_id = (b * nstrips + strip) * nseg + seg;
    const int part_cip = args.part_cip;
    float* pbase = args.partial + (int64_t)wg_id * 9 * part_cip * 64;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = args.ci_off + cit * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (ci < part_cip) pbase[(t * part_cip + ci) * 64 + cot * 32 + r] = acc[t][i];   // (idle waves: zeros for their existing padding channels)
        }
}

template <int MODE>
int launch_mode(const WgRowsArgs& a, bool nb, int nseg, int nstrips, int B, hipStream_t stream) {
    if (nb) {
        P4C_TRY(ensure_dyn_smem((const void*)conv3x3_wgrad_bf16_rows_kernel<MODE, true>, wr::SMEM));
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_rows_kernel<MODE, true>), dim3(nseg, nstrips, B), dim3(512), wr::SMEM, stream, a);
    } else {
        P4C_TRY(ensure_dyn_smem((const void*)conv3x3_wgrad_bf16_rows_kernel<MODE, false>, wr::SMEM));
        hipLaunchKernelGGL((conv3x3_wgrad_bf16_rows_kernel<MODE, false>), dim3(nseg, nstrips, B), dim3(512), wr::SMEM, stream, a);
    }
    return P4C_OK;
}

}  // namespace

// Strip / segment geometry: as many workgroups as fit in the partial buffer's G slots, segments of at least MIN_ROWS rows
static void wgrad_rows_geometry(int G, int B, int H, int W, int* nstrips_out, int* nseg_out) {
    const int nstrips = W / wr::SW;
    int nseg = G / (B * nstrips);
    if (const char* e = diag_env("P4C_WGROWS_NSEG")) nseg = atoi(e);   // (experiments; still clamped to the buffer)
    if (nseg > G / (B * nstrips)) nseg = G / (B * nstrips);
    if (nseg > H / wr::MIN_ROWS) nseg = H / wr::MIN_ROWS;
    if (nseg < 1) nseg = 1;
    *nstrips_out = nstrips;
    *nseg_out = nseg;
}

// in_cs: channels per pixel of x (a multiple of 8); the chunk starts at channel ci_off (a multiple of 32)
bool conv_wgrad_rows_ok(int storage, int in_cs, int ci_off, int dout_cs, int ks, int G, int B, int H, int W) {
    const char* e = diag_env("P4C_NO_WGRAD_ROWS");   // (read per call: A/B scripts and the parity tests switch it)
    if (e && e[0] == '1') return false;
    return storage == P4C_BF16 && in_cs >= 8 && in_cs % 8 == 0 && ci_off % 32 == 0 && ci_off < in_cs && dout_cs == 64 && ks == 3 &&
           W % wr::SW == 0 && H >= wr::MIN_ROWS && B * (W / wr::SW) <= G && (int64_t)H * W * in_cs * 2 < (int64_t)1 << 31;
}

// One chunk of at most 64 input channels: channels ci_off .. min(ci_off + 64, ci_real rounded up to 8) of x (B,H,W,in_cs).
// partial: [nslots][9][part_cip][64] floats, *nslots_out = workgroups launched (<= G; the same for every chunk of a shape)
int launch_conv3x3_wgrad_bf16_rows(const void* x, const float* x_scale, const float* x_shift, int x_relu, const void* dout, float* partial,
                                   int G, int B, int H, int W, hipStream_t stream, const NormBwdCoef* nbp, int* nslots_out, int in_cs,
                                   int ci_off, int ci_real, int part_cip) {
    int nstrips, nseg;
    wgrad_rows_geometry(G, B, H, W, &nstrips, &nseg);
    int hi = ci_real < in_cs ? ci_real : in_cs;   // channels beyond the real ones are zero padding of x: not loaded
    int n_oct = (hi - ci_off + 7) / 8;
    n_oct = n_oct > 8 ? 8 : (n_oct < 1 ? 1 : n_oct);
    const WgRowsArgs a{(const __bf16*)x, x_scale, x_shift, (const __bf16*)dout, partial, H, W, H / nseg, H % nseg, in_cs, ci_off, n_oct,
                       part_cip, nbp ? *nbp : NormBwdCoef{}};
    *nslots_out = B * nstrips * nseg;
    const bool nb = nbp != nullptr && nbp->y != nullptr;
    prof_begin(P4C_PROF_WGRAD3X3_C64, (int64_t)B * H * W, stream);
    int rc;
    if (x_scale)
        rc = x_relu ? launch_mode<2>(a, nb, nseg, nstrips, B, stream) : launch_mode<3>(a, nb, nseg, nstrips, B, stream);
    else
        rc = x_relu ? launch_mode<1>(a, nb, nseg, nstrips, B, stream) : launch_mode<0>(a, nb, nseg, nstrips, B, stream);
    prof_end(P4C_PROF_WGRAD3X3_C64, stream);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_LAUNCH("conv3x3_wgrad_bf16_rows");
    return P4C_OK;
}

}  // namespace p4c
