// conv3x3_bf16_rows: 3x3 convolution 64 -> 64 channels on bf16 NHWC activations, forward AND data gradient (flipped /
// transposed weights) -- the hot kernel of the bf16 HalfUNet plan (replaces mfai's cuDNN/MIOpen convolution call,
// py4cast/lightning.py:591-596).  Row-streaming form (round 3), successor of conv3x3_bf16_ring (conv_bf16.hip):
//
//   * a workgroup owns a 64-pixel-wide STRIP SEGMENT of one sample: output rows y0 .. y0+R-1, input rows y0-1 .. y0+R;
//   * 512 threads: waves 0-3 run the matrix phase (wave = 32 output channels x 32 pixel columns), waves 4-7 the memory
//     side (global -> registers -> normalise/ReLU -> LDS row ring; LDS staging -> HBM + channel statistics);
//   * each INPUT row is read from LDS exactly once: its 12 B operands (3 column shifts x 4 channel slices) feed three
//     rolling accumulators -- output row m-2 (tap row 2), m-1 (tap row 1), m (tap row 0) -- so there are 36 MFMAs per
//     12 LDS reads (the ring kernel: 72 per 48), no tile boundary, and the conversion of a finished output row to
//     bf16 is issued between the last MFMAs of the row that finishes it (the chain that finishes runs two operands ahead):
//     the matrix pipe never waits for an epilogue;
//   * weights stationary in registers (36 A operands = 144 VGPRs), accumulators rotate by name (row loop unrolled by 3);
//   * LDS pixels are padded to 144 bytes: every ds_read_b128 of a B operand is `row base + immediate` and the sixteen
//     pixels of a service group fall in sixteen different bank quads (no swizzle, no per-operand address registers);
//   * one workgroup barrier per FOUR input rows (an "interval"); the memory side stages the next interval's rows and
//     drains the previous interval's output rows meanwhile; its 9 loads / 9 LDS stores / 8 LDS reads / 8 stores per lane
//     and interval are the same instruction stream on every trip (out-of-range offsets instead of branches).
// LDS: input ring 8 rows x 66 px x 144 B = 74.3 KB | output staging 8 rows x 64 px x 128 B = 64 KB.
#include <stddef.h>
#include <stdlib.h>

#include "kernels.hpp"

namespace p4c {

#ifndef P4C_EXP
#define P4C_EXP 0  // diagnostic builds only: 1 no ring stores, 2 no global loads, 4 no output stores, 8 no statistics, 16 no matrix phase
#endif

#ifdef P4C_STAMPS  // diagnostic build only: s_memtime / s_memrealtime stamps of one compute and one loader wave of workgroup 7
__device__ unsigned long long* g_rows_stamps = nullptr;
__device__ __forceinline__ void rows_stamp(int slot) {
    if (g_rows_stamps && blockIdx.x == 7 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0) {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        g_rows_stamps[2 * slot] = t;
        g_rows_stamps[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
#define P4C_STAMP(slot) rows_stamp(slot)
#if P4C_STAMPS == 1
#define P4C_STAMP_ROW(slot) rows_stamp(slot)
#else
#define P4C_STAMP_ROW(slot)
#endif
#else
#define P4C_STAMP(slot)
#define P4C_STAMP_ROW(slot)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace rows {
constexpr int SW = 64;                 // strip width (output pixels)
constexpr int LW = SW + 2;             // input columns of a strip
constexpr int NR = 8;                  // ring rows: 4 being read, 4 being staged
constexpr int SROW = SW * 128;         // one staged output row
constexpr int NS = 8;
constexpr int STGB = NS * SROW;
constexpr int ROWSLOTS = LW * 8;       // 16-byte slots per input row
// (per loader lane and interval of IV rows -- IV = 4, or 2 where both normalisation-backward roles share a launch: input slots
//  NLD = ceil(IV x 528 / 256) = 9 | 5, output slots NST = IV x 512 / 256 = 8 | 4: constants of the kernel)
// LDS layout of an input pixel, by MFMA shape (MF = 32: v_mfma_f32_32x32x16_bf16, MF = 16: v_mfma_f32_16x16x32_bf16):
//   MF 32: 144 bytes (64 bf16 channels + 16 B pad): a B operand is `row base + immediate`; the 16 pixels of a ds_read_b128
//          service group (one channel slice) fall in 16 different bank quads;
//   MF 16: 128 bytes, 16-byte slots XOR-swizzled by (pixel >> 1) & 7: there a service group mixes two channel slices
//          ({pixels 0-3, 12-15} of slice g with {4-11} of slice g+1), which no padding separates but this swizzle does.
template <int MF> struct Lay {
    static constexpr int PIXB = MF == 32 ? 144 : 128;
    static constexpr int RROW = LW * PIXB;
    static constexpr int RINGB = NR * RROW;
    static constexpr int SMEM = RINGB + STGB;
    static __device__ __forceinline__ int slot_off(int col, int c8) {   // byte offset of channel octet c8 of pixel column col
        return MF == 32 ? col * PIXB + 16 * c8 : col * PIXB + ((c8 ^ ((col >> 1) & 7)) << 4);
    }
};
}  // namespace rows

constexpr int OOB = 0x7fffffff;

// relu(v*scale+shift) on the 2 bf16 channels packed in one word (fp32 arithmetic, one rounding), or parts of it.
// MODE 0: copy, 1: ReLU, 2: scale/shift + ReLU, 3: scale/shift.  (MODE 4 of the kernel -- pass 2 of a normalisation backward,
// kernels.hpp: NormBwdCoef -- has two operands: nb2 below.)
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
        const s16x2 z = {0, 0};   // negative floats are negative int16: max(., 0) clears them
        w = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
    }
    return w;
}

// dY = alpha * g + beta * y + delta with g = dA where the forward ReLU was alive (y * sc + sh > 0), on the 2 channels packed in the
// words a2 (dA) / y2 (fp32 arithmetic, one rounding to bf16)
__device__ __forceinline__ unsigned int nb2(unsigned int a2, unsigned int y2, f32x2 al, f32x2 be, f32x2 de, f32x2 sc, f32x2 sh) {
    const float ylo = __builtin_bit_cast(float, y2 << 16), yhi = __builtin_bit_cast(float, y2 & 0xffff0000u);
    const float glo = __builtin_fmaf(ylo, sc.x, sh.x) > 0.f ? __builtin_bit_cast(float, a2 << 16) : 0.f;
    const float ghi = __builtin_fmaf(yhi, sc.y, sh.y) > 0.f ? __builtin_bit_cast(float, a2 & 0xffff0000u) : 0.f;
    const float lo = __builtin_fmaf(al.x, glo, __builtin_fmaf(be.x, ylo, de.x));
    const float hi = __builtin_fmaf(al.y, ghi, __builtin_fmaf(be.y, yhi, de.y));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}

// raw buffer descriptor over [base, base+bytes): loads beyond `bytes` return 0 and stores are dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

// workgroup barrier that orders LDS traffic only: global loads / stores stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One input row of the matrix phase.  P / Q / S: accumulators of output rows m-2 / m-1 / m (tap rows 2 / 1 / 0); S starts
// from zero here, P is complete after this row and goes to the staging row `stg` as bf16.  fb: ring of six B operands;
// on entry operands 0..2 of this row are in flight in fb[0..2], on exit (pf) those of the next row are.
// Step s: read operand s+3, P uses operand s, Q and S operand s-2 (P runs two operands ahead so that its last MFMA
// is six MFMAs before the end of the row: the conversion below never waits for the matrix pipe).
template <bool HP, bool HQ, bool HS>
__device__ __forceinline__ void conv_row(const bf16x8 (&A)[9][4], f32x16& P, f32x16& Q, f32x16& S, bf16x8 (&fb)[6],
                                         const char* rb, const char* rbn, bool pf, char* stg, const int (&soff)[4]) {
    using namespace rows;
    constexpr int PIXB = Lay<32>::PIXB;
    auto stage = [&](int g) __attribute__((always_inline)) {
        // C[co][px]: lane = pixel column r (+ half h), register quad g -> channels 32ct + 8g + 4h .. +3
        const f32x2 lo = {P[4 * g], P[4 * g + 1]}, hi = {P[4 * g + 2], P[4 * g + 3]};
        u32x2 o;
        o[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(lo, bf16x2));
        o[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(hi, bf16x2));
        *reinterpret_cast<u32x2*>(stg + soff[g]) = o;
    };
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int o = s + 3;
        if (!(P4C_EXP & 16)) {
            if (o < 12) {
                fb[o % 6] = *reinterpret_cast<const bf16x8*>(rb + (o >> 2) * PIXB + (o & 3) * 32);
            } else if (o < 15) {
                if (pf) fb[o % 6] = *reinterpret_cast<const bf16x8*>(rbn + ((o - 12) >> 2) * PIXB + ((o - 12) & 3) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (HP && s < 12) {
                P = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[6 + (s >> 2)][s & 3], fb[s % 6], P, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (s >= 2) {
            const int q = s - 2;
            if (HQ && !(P4C_EXP & 16)) {
                Q = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[3 + (q >> 2)][q & 3], fb[q % 6], Q, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (HP && s >= 12) { stage(2 * (s - 12)); __builtin_amdgcn_sched_barrier(0); }
            if (HS && !(P4C_EXP & 16)) {
                if (q == 0) {
                    f32x16 z;
#pragma unroll
                    for (int i = 0; i < 16; ++i) z[i] = 0.f;
                    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q >> 2][q & 3], fb[q % 6], z, 0, 0, 0);
                } else {
                    S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[q >> 2][q & 3], fb[q % 6], S, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (HP && s >= 12) { stage(2 * (s - 12) + 1); __builtin_amdgcn_sched_barrier(0); }
        }
    }
}

// The same row on v_mfma_f32_16x16x32_bf16 (on real data the chip is power-limited and this shape sustains ~1.15x the rate of
// 32x32x16: tools/diagnostics/mfma_power.hip).  The wave's 32 channels x 32 pixels are 2 x 2 blocks of 16 x 16; K = 64 input
// channels in two steps of 32.  Operand o = (kx, ks, nb): column shift, channel half, pixel block -- 12 per row as before --
// feeds 2 (channel blocks) x 3 (tap rows) MFMAs.  boff[kx][ks]: the lane's LDS address of that operand in the CURRENT ring
// row (pixel block nb = +2048 bytes); moved on to the next row (`delta`) once the last read of this row has been issued.
struct Acc16 { f32x4 v[2][2]; };   // [channel block][pixel block]

template <bool HP, bool HQ, bool HS>
__device__ __forceinline__ void conv_row16(const bf16x8 (&A)[9][2][2], Acc16& P, Acc16& Q, Acc16& S, bf16x8 (&fb)[6],
                                           int (&boff)[3][2], int delta, bool pf, const char* lring, char* stg,
                                           const int (&soff)[2]) {
    auto stage = [&](int mb, int nb) __attribute__((always_inline)) {
        // C block: lane = pixel l % 16, registers = channels 32ct + 16mb + 4 (l / 16) .. +3
        const f32x4 a = P.v[mb][nb];
        const f32x2 lo = {a[0], a[1]}, hi = {a[2], a[3]};
        u32x2 o;
        o[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(lo, bf16x2));
        o[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(hi, bf16x2));
        *reinterpret_cast<u32x2*>(stg + soff[mb] + nb * 2048) = o;
    };
#pragma unroll
    for (int s = 0; s < 14; ++s) {
        const int o = s + 3;
        if (!(P4C_EXP & 16)) {
            if (o < 12) {
                fb[o % 6] = *reinterpret_cast<const bf16x8*>(lring + boff[o >> 2][(o >> 1) & 1] + (o & 1) * 2048);
            } else if (o < 15) {
                if (pf) fb[o % 6] = *reinterpret_cast<const bf16x8*>(lring + boff[(o - 12) >> 2][((o - 12) >> 1) & 1] + ((o - 12) & 1) * 2048);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (HP && s < 12) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    P.v[mb][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[6 + (s >> 2)][(s >> 1) & 1][mb], fb[s % 6], P.v[mb][s & 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (s == 8) {   // every read of this row has been issued: the addresses move on to the next ring row
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) boff[kx][ks] += delta;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (s >= 2) {
            const int q = s - 2;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                if (HQ && !(P4C_EXP & 16)) {
                    Q.v[mb][q & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[3 + (q >> 2)][(q >> 1) & 1][mb], fb[q % 6], Q.v[mb][q & 1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (HP && s >= 12) { stage(s - 12, mb); __builtin_amdgcn_sched_barrier(0); }
                if (HS && !(P4C_EXP & 16)) {
                    if (q < 2) {   // first operand of this pixel block: start from zero
                        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                        S.v[mb][q & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q >> 2][(q >> 1) & 1][mb], fb[q % 6], z, 0, 0, 0);
                    } else {
                        S.v[mb][q & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[q >> 2][(q >> 1) & 1][mb], fb[q % 6], S.v[mb][q & 1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
}

// Every argument of the kernel in ONE by-value struct.  The fields of `fin` (in-kernel BatchNorm finalize) and bst.rstd are needed
// only after the row loop; left to the compiler they are loaded from the kernel-argument segment at entry and stay in ~30 scalar
// registers for the whole kernel -- with four buffer descriptors and the loop state that overflowed the scalar file, the spilled
// scalars took vector-register lanes and the loader spilled to scratch (.sgpr_spill_count 54-78).  late_arg() loads a field from the
// argument segment at the point of use (the pointer is made opaque there, so the load cannot be hoisted).
struct RowsArgs {
    const __bf16* in; const __bf16* wp; const float* in_scale; const float* in_shift; __bf16* out; float* stat_partial;
    int out_cs, H, W, rows_lo, rows_rem, fin_on, in_cs;
    RingBwdStats bst;
    NormBwdCoef nb;
    BatchFin fin;
};
template <typename T>
__device__ __forceinline__ T late_arg(size_t offset) {
#if defined(__HIP_DEVICE_COMPILE__)
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *reinterpret_cast<const T*>(ka + offset);
#else
    (void)offset;
    return T{};
#endif
}

// CPT ("compact channels", plain launches only): the input has in_cs <= 64 channels per pixel and the output out_cs <= 64 (multiples
// of 8) -- channel slots beyond them are loaded as zeros / not stored, so a 24- or 48-channel feature map is convolved in place
// instead of through a zero-padded 64-channel copy and a sliced 64-channel result (SwinUNetR's decoder levels).
// IV: input rows per interval (one workgroup barrier each).  4 everywhere but in the launch that takes BOTH normalisation-backward roles
// (MODE 4 + BST: pass 2 of this block's on the way in, pass 1 of the next one's in the drain): its loader would hold 72 registers of
// prefetched dA / y rows + 64 of output / y rows for the drain -- 137 spilled registers; with two rows per interval the images are
// half that and the kernel has none (240 registers).
template <int MODE, bool BST, int MF, int KS, bool CPT = false, int IV = 4>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
    conv3x3_bf16_rows_kernel(RowsArgs args) {
    static_assert(!CPT || (MODE == 0 && !BST && MF == 32), "compact channels: plain launches only");
    static_assert(IV == 4 || IV == 2, "rows per interval");
    using namespace rows;
    constexpr int NLD = (IV * ROWSLOTS + 255) / 256, NST = IV * 2;
    const __bf16* __restrict__ in = args.in;
    const __bf16* __restrict__ wp = args.wp;
    const float* __restrict__ in_scale = args.in_scale;
    const float* __restrict__ in_shift = args.in_shift;
    __bf16* __restrict__ out = args.out;
    float* __restrict__ stat_partial = args.stat_partial;
    const int out_cs = args.out_cs, H = args.H, W = args.W, rows_lo = args.rows_lo, rows_rem = args.rows_rem;
    const RingBwdStats& bst = args.bst;   // (scale / shift / mean / y: read during the set-up; rstd late)
    const NormBwdCoef& nb = args.nb;
    typedef Lay<MF> LY;
    constexpr int PIXB = LY::PIXB, RROW = LY::RROW, RINGB = LY::RINGB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lring = smem;
    char* lstg = smem + RINGB;

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    // workgroup -> (sample, strip, row segment) = blockIdx (z, y, x); the first H % nseg segments of a strip have one row more
    // (no divisions on the way into a kernel whose whole job takes a few microseconds on the coarse levels)
    const int seg = blockIdx.x, strip = blockIdx.y, b = blockIdx.z;
    const int nseg = gridDim.x, nstrips = gridDim.y;
    const int wg_id = (b * nstrips + strip) * nseg + seg, wg_count = nseg * nstrips * gridDim.z;
    const int y0 = seg * rows_lo + (seg < rows_rem ? seg : rows_rem);
    const int R = rows_lo + (seg < rows_rem ? 1 : 0);          // >= 2 (host)
    const int x0 = strip * SW;
    const int K = (R + 2 + IV - 1) / IV;                       // intervals of IV input rows: ceil((R + 2) / IV)

    if (wv >= 4) {
        // ------------------------------------------------------------ memory side
        const int ltid = threadIdx.x - 256, lwv = wv - 4, c8 = ltid & 7;
        const int ipb = CPT ? args.in_cs * 2 : 128;   // bytes per input pixel
        const bool in_ch = !CPT || 8 * c8 < args.in_cs, out_ch = !CPT || 8 * c8 < out_cs;   // this lane's channel octet exists
        const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(in + (int64_t)b * H * W * (ipb >> 1), (unsigned int)H * W * (unsigned int)ipb);
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(out + (int64_t)b * H * W * out_cs, (unsigned int)H * W * out_cs * 2u);
        // per-lane tables: slot `it` of a staged group of four input rows (row-major, 528 slots per row)
        // Column validity is static per lane (the strip is fixed): slots of columns outside the image get lim = "never active";
        // those LDS columns are zeroed once below and never written, so only ROWS outside the image need the checked path.
        int gofs[NLD], lofs[NLD], lim[NLD];
        int rr = 0, rem = ltid;   // idx = ltid + 256 it = rr * ROWSLOTS + rem, kept incrementally (no divisions: 256 < ROWSLOTS)
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int idx = ltid + it * 256;
            if (it > 0) { rem += 256; if (rem >= ROWSLOTS) { rem -= ROWSLOTS; ++rr; } }
            const int col = rem >> 3;
            gofs[it] = in_ch ? (rr * W + col) * ipb + 16 * c8 : OOB;   // (an absent octet loads as zeros and is staged as such)
            lofs[it] = rr * RROW + LY::slot_off(col, c8);
            lim[it] = ((unsigned)(x0 - 1 + col) < (unsigned)W) ? idx : OOB;
        }
        {
            const int left = x0 == 0 ? 1 : 0;
            const int c_hi = W - x0 + 1 < LW ? W - x0 + 1 : LW;       // first column at or beyond the right image edge
            const int ninv = left + (LW - c_hi);
            for (int rz = 0; rz < NR; ++rz)
                for (int j = ltid; j < ninv * 8; j += 256) {
                    const int s8 = j & 7, ci = j >> 3;
                    const int col = (left && ci == 0) ? 0 : c_hi + ci - left;
                    *reinterpret_cast<u32x4*>(lring + rz * RROW + col * PIXB + 16 * s8) = u32x4{0u, 0u, 0u, 0u};   // (all 8 slots of the pixel: any order)
                }
        }
        int dofs[NST], sofs[NST];
#pragma unroll
        for (int j = 0; j < NST; ++j) {
            const int px = (ltid + j * 256) >> 3, row = px >> 6, col = px & 63;
            dofs[j] = out_ch ? ((row * W + col) * out_cs + 8 * c8) * 2 : OOB;
            sofs[j] = row * SROW + col * 128 + ((c8 ^ ((col >> 1) & 7)) << 4);
        }
        // normalisation of this lane's 8 channels (one sample per workgroup)
        f32x2 sc[4], sh[4], al[4], be[4], dl[4];   // scale, shift of the producing norm (MODE 2 / 3 / 4);  MODE 4: alpha, beta, delta of NormBwdCoef
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            sc[k] = sh[k] = al[k] = be[k] = dl[k] = f32x2{0.f, 0.f};
            if (MODE == 2 || MODE == 3) {
                sc[k] = *reinterpret_cast<const f32x2*>(in_scale + b * 64 + 8 * c8 + 2 * k);
                sh[k] = *reinterpret_cast<const f32x2*>(in_shift + b * 64 + 8 * c8 + 2 * k);
            }
            if (MODE == 4) {
                const int ch = 8 * c8 + 2 * k;
                const f32x2 ga = *reinterpret_cast<const f32x2*>(nb.gamma + ch), rs = *reinterpret_cast<const f32x2*>(nb.rstd + b * 64 + ch);
                const f32x2 mu = *reinterpret_cast<const f32x2*>(nb.mean + b * 64 + ch);
                const f32x2 q1 = *reinterpret_cast<const f32x2*>(nb.k1 + b * 64 + ch), q2 = *reinterpret_cast<const f32x2*>(nb.k2 + b * 64 + ch);
                sc[k] = *reinterpret_cast<const f32x2*>(nb.scale + b * 64 + ch);
                sh[k] = *reinterpret_cast<const f32x2*>(nb.shift + b * 64 + ch);
                al[k] = rs * ga;
                be[k] = -(rs * rs) * q2;
                dl[k] = rs * rs * q2 * mu - rs * q1;
            }
        }
        struct Img { u32x4 s[NLD]; };
        Img ta, ty;   // ty: the y rows of the same slots (MODE 4 only)
        const __amdgpu_buffer_rsrc_t rs_nby = make_rsrc(MODE == 4 ? reinterpret_cast<const __bf16*>(nb.y) + (int64_t)b * H * W * 64 : in,
                                                        (unsigned int)H * W * 128u);
        auto load = [&](Img& im, int k) __attribute__((always_inline)) {
            const int m0 = IV * k;
            int nrows = R + 2 - m0;
            nrows = nrows > IV ? IV : (nrows < 0 ? 0 : nrows);
            if ((P4C_EXP & 2) && k > 1) nrows = 0;
            const int nact = nrows * ROWSLOTS;
            const int gy0 = y0 - 1 + m0, gx0 = x0 - 1;
            if (gy0 >= 0 && gy0 + nrows <= H) {
                const int so = (gy0 * W + gx0) * ipb;   // rows inside the image: one scalar offset + the lane's table
#pragma unroll
                for (int it = 0; it < NLD; ++it)
                    im.s[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (lim[it] < nact) ? gofs[it] : OOB, so, 0);
                if (MODE == 4) {
#pragma unroll
                    for (int it = 0; it < NLD; ++it)
                        ty.s[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_nby, (lim[it] < nact) ? gofs[it] : OOB, so, 0);
                }
            } else {
#pragma unroll
                for (int it = 0; it < NLD; ++it) {
                    const int idx = ltid + it * 256, rr = idx / ROWSLOTS;   // (rows outside the image: first / last interval of a strip only)
                    const int gy = gy0 + rr, gx = gx0 + ((idx - rr * ROWSLOTS) >> 3);
                    const bool ok = (idx < nact) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W) & in_ch;
                    im.s[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? (gy * W + gx) * ipb + 16 * c8 : OOB, 0, 0);
                    if (MODE == 4) ty.s[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_nby, ok ? (gy * W + gx) * 128 + 16 * c8 : OOB, 0, 0);
                }
            }
        };
        auto store = [&](const Img& im, int k) __attribute__((always_inline)) {
            const int m0 = IV * k;
            int nrows = R + 2 - m0;
            nrows = nrows > IV ? IV : (nrows < 0 ? 0 : nrows);
            if ((P4C_EXP & 1) && k > 1) nrows = 0;
            const int nact = nrows * ROWSLOTS;
            char* dst = lring + (m0 & 7) * RROW;
            const int gy0 = y0 - 1 + m0, gx0 = x0 - 1;
            const bool interior = gy0 >= 0 && gy0 + nrows <= H;
            if (MODE == 0 || interior) {
#pragma unroll
                for (int it = 0; it < NLD; ++it) {
                    u32x4 o = im.s[it];
                    if (MODE == 4) {
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) o[k2] = nb2(o[k2], ty.s[it][k2], al[k2], be[k2], dl[k2], sc[k2], sh[k2]);
                    } else if (MODE != 0) {
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) o[k2] = xform2<MODE>(o[k2], sc[k2], sh[k2]);
                    }
                    if (lim[it] < nact) *reinterpret_cast<u32x4*>(dst + lofs[it]) = o;
                }
            } else {
#pragma unroll
                for (int it = 0; it < NLD; ++it) {
                    // zero padding applies to the NORMALISED activation: out-of-image slots are cleared after the transform
                    const int idx = ltid + it * 256, rr = idx / ROWSLOTS;
                    const int gy = gy0 + rr, gx = gx0 + ((idx - rr * ROWSLOTS) >> 3);
                    const unsigned int keep = (((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W)) ? 0xffffffffu : 0u;
                    u32x4 o = im.s[it];
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2)
                        o[k2] = (MODE == 4 ? nb2(o[k2], ty.s[it][k2], al[k2], be[k2], dl[k2], sc[k2], sh[k2]) : xform2<MODE>(o[k2], sc[k2], sh[k2])) & keep;
                    if (lim[it] < nact) *reinterpret_cast<u32x4*>(dst + lofs[it]) = o;
                }
            }
        };
        // channel statistics of this lane's 8 channels over the pixels it drains
        float a1[8], a2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a1[q] = a2[q] = 0.f;
        // BST: the y rows of the next group to drain (prefetched one interval ahead), the lane's normalisation constants
        u32x4 yq[NST];
        f32x2 bsc[4], bsh[4], bmu[4];
#pragma unroll
        for (int j = 0; j < NST; ++j) yq[j] = u32x4{0u, 0u, 0u, 0u};   // (y offsets = dofs: these launches have out_cs == 64)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bsc[k] = bsh[k] = bmu[k] = f32x2{0.f, 0.f};
            if (BST) {
                bsc[k] = *reinterpret_cast<const f32x2*>(bst.scale + b * 64 + 8 * c8 + 2 * k);
                bsh[k] = *reinterpret_cast<const f32x2*>(bst.shift + b * 64 + 8 * c8 + 2 * k);
                bmu[k] = *reinterpret_cast<const f32x2*>(bst.mean + b * 64 + 8 * c8 + 2 * k);
            }
        }
        const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(BST ? reinterpret_cast<const __bf16*>(bst.y) + (int64_t)b * H * W * 64 : in,
                                                      (unsigned int)H * W * 128u);
        auto yload = [&](int k) __attribute__((always_inline)) {   // y rows of the group drain(k) handles
            const int n0 = IV * k - IV - 2;
            if (n0 >= 0 && n0 + IV <= R && x0 + SW <= W) {
                const int so = ((y0 + n0) * W + x0) * 128;
#pragma unroll
                for (int j = 0; j < NST; ++j) yq[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, dofs[j], so, 0);
            } else {
#pragma unroll
                for (int j = 0; j < NST; ++j) {
                    const int px = (ltid + j * 256) >> 3;
                    const int n = n0 + (px >> 6), gx = x0 + (px & 63);
                    const bool valid = ((unsigned)n < (unsigned)R) & (gx < W);
                    yq[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, valid ? (((y0 + n) * W + gx) * 64 + 8 * c8) * 2 : OOB, 0, 0);
                }
            }
        };
        // drain(k): the IV output rows IV (k-1) - 2 .. the compute waves staged during interval k-1 (IV = 4: rows 4k-6 .. 4k-3)
        auto drain = [&](int k) __attribute__((always_inline)) {
            const int n0 = IV * k - IV - 2;
            const char* stg = lstg + ((IV * (k - 1)) & 7) * SROW;
            u32x4 v[NST];
#pragma unroll
            for (int j = 0; j < NST; ++j) v[j] = *reinterpret_cast<const u32x4*>(stg + sofs[j]);
            if (n0 >= 0 && n0 + IV <= R && x0 + SW <= W) {
                const int so = ((y0 + n0) * W + x0) * out_cs * 2;
#pragma unroll
                for (int j = 0; j < NST; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(v[j], rs_out, (P4C_EXP & 4) ? OOB : dofs[j], so, 0);
            } else {
#pragma unroll
                for (int j = 0; j < NST; ++j) {
                    const int px = (ltid + j * 256) >> 3;
                    const int n = n0 + (px >> 6), gx = x0 + (px & 63);
                    const bool valid = ((unsigned)n < (unsigned)R) & (gx < W);
                    if (!valid) v[j] = u32x4{0u, 0u, 0u, 0u};   // keeps the statistics below unmasked
                    __builtin_amdgcn_raw_buffer_store_b128(v[j], rs_out, (valid && out_ch && !(P4C_EXP & 4)) ? (((y0 + n) * W + gx) * out_cs + 8 * c8) * 2 : OOB, 0, 0);
                }
            }
            if (BST) {
#pragma unroll
                for (int j = 0; j < NST; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float ylo = __builtin_bit_cast(float, yq[j][q] << 16);
                        const float yhi = __builtin_bit_cast(float, yq[j][q] & 0xffff0000u);
                        float lo = __builtin_bit_cast(float, v[j][q] << 16);
                        float hi = __builtin_bit_cast(float, v[j][q] & 0xffff0000u);
                        lo = __builtin_fmaf(ylo, bsc[q].x, bsh[q].x) > 0.f ? lo : 0.f;
                        hi = __builtin_fmaf(yhi, bsc[q].y, bsh[q].y) > 0.f ? hi : 0.f;
                        a1[2 * q] += lo; a2[2 * q] = __builtin_fmaf(lo, ylo - bmu[q].x, a2[2 * q]);
                        a1[2 * q + 1] += hi; a2[2 * q + 1] = __builtin_fmaf(hi, yhi - bmu[q].y, a2[2 * q + 1]);
                    }
                yload(k + 1);
            } else if (stat_partial && !(P4C_EXP & 8)) {
#pragma unroll
                for (int j = 0; j < NST; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float lo = __builtin_bit_cast(float, v[j][q] << 16);
                        const float hi = __builtin_bit_cast(float, v[j][q] & 0xffff0000u);
                        a1[2 * q] += lo; a2[2 * q] = __builtin_fmaf(lo, lo, a2[2 * q]);
                        a1[2 * q + 1] += hi; a2[2 * q + 1] = __builtin_fmaf(hi, hi, a2[2 * q + 1]);
                    }
            }
        };

        // One register image: the loads of group k+2 are issued right after group k+1 went to LDS, i.e. a whole interval
        // (~3 us of matrix work) before they are needed.
        if (lwv == 0) P4C_STAMP(0);
        load(ta, 0);
        store(ta, 0);
        load(ta, 1);
        if (lwv == 0) P4C_STAMP(1);
        lds_barrier();
        for (int k = 0; k < K; ++k) {
            // compute reads the rows of interval k: stage interval k+1, prefetch k+2, drain the output rows of interval k-1
            if (lwv == 0) P4C_STAMP_ROW(100 + 4 * k);
            store(ta, k + 1);
            if (lwv == 0) P4C_STAMP_ROW(101 + 4 * k);
            load(ta, k + 2);
            drain(k);
            if (lwv == 0) P4C_STAMP(102 + 4 * k);
            lds_barrier();
            if (lwv == 0) P4C_STAMP(103 + 4 * k);
        }
        drain(K);
        if (lwv == 0) P4C_STAMP(2);

        if (args.fin_on) {
            const BatchFin fin = late_arg<BatchFin>(offsetof(RowsArgs, fin));
            // ---- BatchNorm finished in place (kernels.hpp: BatchFin).  The compute waves have left (their last barrier was the
            // one that released the final staged rows), so barriers from here on are among the four loader waves only.
            float* lred = reinterpret_cast<float*>(lring);                     // [4 waves][128], then doubles [8][128]; the ring is dead
            unsigned int* lflag = reinterpret_cast<unsigned int*>(lred + 4 * 128 + 8 * 256);
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
                // release FENCE (at the end of a kernel that has just written its output map it would write back the XCD's whole L2).
#pragma unroll
                for (int j = lane; j < 128; j += 64)   // fixed order over the waves
                    __hip_atomic_store(fin.slots + (int64_t)wg_id * 128 + j,
                                       (lred[j] + lred[128 + j]) + (lred[256 + j] + lred[384 + j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) *lflag = __hip_atomic_fetch_add(fin.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            lds_barrier();
            if (*lflag != (unsigned)wg_count - 1) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // invalidate only: the other workgroups' slots are read from memory
            // 256 threads: 32 column quads x 8 slot groups; slots summed in increasing order within a group, groups in order
            const int cq = ltid & 31, sg = ltid >> 5;
            double acc4[4] = {0.0, 0.0, 0.0, 0.0};
            for (int s0 = sg; s0 < wg_count; s0 += 8 * 8) {
                p4c_f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sl = s0 + 8 * u;
                    v[u] = sl < wg_count ? *(reinterpret_cast<const p4c_f32x4*>(fin.slots + (int64_t)sl * 128) + cq)
                                               : p4c_f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) { acc4[0] += v[u].x; acc4[1] += v[u].y; acc4[2] += v[u].z; acc4[3] += v[u].w; }
            }
            double* dred = reinterpret_cast<double*>(lred + 4 * 128);   // [8][128]
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
                const float scl = fin.gamma[c] * rstd, shf = fin.beta[c] - (float)mean * scl;
                for (int bb = 0; bb < fin.B; ++bb) {
                    fin.scale[bb * 64 + c] = scl; fin.shift[bb * 64 + c] = shf; fin.mean[bb * 64 + c] = (float)mean; fin.rstd[bb * 64 + c] = rstd;
                }
            }
            if (ltid == 0) __hip_atomic_store(fin.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (stat_partial) {
            const float* brstd = BST ? late_arg<const float*>(offsetof(RowsArgs, bst) + offsetof(RingBwdStats, rstd)) : nullptr;
            // one slot per (workgroup, loader wave) of this sample: [2][64] sums
            const int nslot = nstrips * nseg * 4;
            float* dst = stat_partial + ((int64_t)b * nslot + (strip * nseg + seg) * 4 + lwv) * 128;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float u = a1[q], v = a2[q];
                u += __shfl_xor(u, 8); v += __shfl_xor(v, 8);
                u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
                u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
                if (BST) v *= brstd[b * 64 + 8 * c8 + q];   // sums of g * (y - mean) -> sums of g * xhat
                if (lane < 8) { dst[8 * c8 + q] = u; dst[64 + 8 * c8 + q] = v; }
            }
        }
        return;
    }

    // ---------------------------------------------------------------- compute waves
    // wave wv: output channels 32*ct .. +31 (ct = wv >> 1), pixel columns 32*ph .. +31 of the strip (ph = wv & 1)
    const int ct = wv >> 1, ph = wv & 1;
    if (wv == 0) P4C_STAMP(9);
    const int last = R + 1;
    int m = 0;
    if constexpr (KS == 1) {
        // 1x1 convolution on the same machinery (memory-bound: 4 MFMAs per row): only the centre column of the ring rows is
        // read, output row n is formed from input row n + 1 and -- to keep the staging / drain cadence of the 3x3 kernel --
        // handed to the memory side one row later, exactly where the 3x3 form finishes it.
        bf16x8 A1[4];
        {
            const char* wsrc = reinterpret_cast<const char*>(wp) + (h * 64 + ct * 32 + r) * 16;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) A1[ks] = *reinterpret_cast<const bf16x8*>(wsrc + ks * 2048);
        }
        const int lane_base = (32 * ph + r + 1) * PIXB + 16 * h;   // centre column: pixel + 1 (the ring keeps the halo column)
        int soff[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) soff[g] = (32 * ph + r) * 128 + 8 * h + (((4 * ct + g) ^ (((32 * ph + r) >> 1) & 7)) << 4);
        f32x16 prev, cur;
#pragma unroll
        for (int i = 0; i < 16; ++i) prev[i] = cur[i] = 0.f;
        if (wv == 0) P4C_STAMP(10);
        lds_barrier();
        if (wv == 0) P4C_STAMP(11);
        for (; m <= last; ++m) {
            if (m >= 1 && m <= R && !(P4C_EXP & 16)) {
                const char* rb = lring + (m & 7) * RROW + lane_base;
                bf16x8 fbq[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fbq[ks] = *reinterpret_cast<const bf16x8*>(rb + ks * 32);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks == 0) {
                        f32x16 z;
#pragma unroll
                        for (int i = 0; i < 16; ++i) z[i] = 0.f;
                        cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[ks], fbq[ks], z, 0, 0, 0);
                    } else {
                        cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[ks], fbq[ks], cur, 0, 0, 0);
                    }
                }
            }
            if (m >= 2) {   // output row m - 2 (held since the previous trip) goes to staging row m & 7
                char* stg = lstg + (m & 7) * SROW;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x2 lo = {prev[4 * g], prev[4 * g + 1]}, hi = {prev[4 * g + 2], prev[4 * g + 3]};
                    u32x2 o;
                    o[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(lo, bf16x2));
                    o[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(hi, bf16x2));
                    *reinterpret_cast<u32x2*>(stg + soff[g]) = o;
                }
            }
            prev = cur;
            if (m == last || (m & (IV - 1)) == IV - 1) lds_barrier();
        }
    } else if constexpr (MF == 32) {
        bf16x8 A[9][4];
        {
            const char* wsrc = reinterpret_cast<const char*>(wp) + (h * 64 + ct * 32 + r) * 16;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) A[tap][ks] = *reinterpret_cast<const bf16x8*>(wsrc + (tap * 4 + ks) * 2048);
        }
        const int lane_base = (32 * ph + r) * PIXB + 16 * h;   // B operand (kx, ks) of ring row j: + j*RROW + kx*PIXB + 32*ks
        int soff[4];   // staging offsets of this lane's 4 channel quads (slot XOR (px >> 1) & 7: two passes per 8-byte store, the minimum)
#pragma unroll
        for (int g = 0; g < 4; ++g) soff[g] = (32 * ph + r) * 128 + 8 * h + (((4 * ct + g) ^ (((32 * ph + r) >> 1) & 7)) << 4);
        f32x16 a, bq, c;
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = bq[i] = c[i] = 0.f;
        bf16x8 fb[6];
        auto rbase = [&](int mm) __attribute__((always_inline)) { return lring + (mm & 7) * RROW + lane_base; };
        auto issue3 = [&](const char* rb) __attribute__((always_inline)) {
            if (!(P4C_EXP & 16)) {
#pragma unroll
                for (int o = 0; o < 3; ++o) fb[o] = *reinterpret_cast<const bf16x8*>(rb + (o >> 2) * PIXB + (o & 3) * 32);
            }
        };
        if (wv == 0) P4C_STAMP(10);
        lds_barrier();   // the first interval's rows are staged
        if (wv == 0) P4C_STAMP(11);
        issue3(rbase(0));
#define P4C_ROW(HP, HQ, HS, P, Q, S)                                                                                        \
    {                                                                                                                       \
        const bool pf = ((m & (IV - 1)) != IV - 1) && (m != last);                                                          \
        if (wv == 0) P4C_STAMP_ROW(300 + 2 * m);                                                                            \
        conv_row<HP, HQ, HS>(A, P, Q, S, fb, rbase(m), rbase(m + 1), pf, lstg + (m & 7) * SROW, soff);                     \
        if (wv == 0) P4C_STAMP_ROW(301 + 2 * m);                                                                            \
        if (m == last) {                                                                                                    \
            P4C_STAMP(700 + 8 * (m >> 2) + 2 * wv);                                                                         \
            lds_barrier();                                                                                                  \
            P4C_STAMP(701 + 8 * (m >> 2) + 2 * wv);                                                                         \
        } else if ((m & (IV - 1)) == IV - 1) {                                                                              \
            P4C_STAMP(700 + 8 * (m >> 2) + 2 * wv);                                                                         \
            lds_barrier();                                                                                                  \
            P4C_STAMP(701 + 8 * (m >> 2) + 2 * wv);                                                                         \
            issue3(rbase(m + 1));                                                                                           \
        }                                                                                                                   \
        ++m;                                                                                                                \
    }
        P4C_ROW(false, false, true, a, a, a)      // input row 0: tap row 0 of output row 0
        P4C_ROW(false, true, true, a, a, bq)      // input row 1
        // input rows 2 .. R-1 feed three output rows each; the accumulators rotate by NAME in the main loop (unrolled by three,
        // single exit: anything else makes the compiler copy accumulators and spill weights) ...
        while (m + 3 <= R) {
            P4C_ROW(true, true, true, a, bq, c)
            P4C_ROW(true, true, true, bq, c, a)
            P4C_ROW(true, true, true, c, a, bq)
        }
        // ... and by register moves in the at most two left-over rows and the two closing rows (once per workgroup)
#define P4C_ROTATE() { const f32x16 t_ = a; a = bq; bq = c; c = t_; }
        while (m < R) {
            P4C_ROW(true, true, true, a, bq, c)
            P4C_ROTATE()
        }
        P4C_ROW(true, true, false, a, bq, a)      // input row R: tap rows 2 / 1 of the last two output rows
        P4C_ROTATE()
        P4C_ROW(true, false, false, a, a, a)      // input row R+1
#undef P4C_ROTATE
#undef P4C_ROW
        if (P4C_EXP & 16) { asm volatile("" ::"v"(A[0][0]), "v"(A[8][3])); }
    } else {
        const int l16 = lane & 15, g = lane >> 4;
        bf16x8 A[9][2][2];   // [tap][channel half (K step of 32)][channel block]: lane = output channel l16, input channels 8g .. 8g+7 of the step
        {
            const char* wsrc = reinterpret_cast<const char*>(wp) + g * 1024 + (ct * 32 + l16) * 16;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) A[tap][ks][mb] = *reinterpret_cast<const bf16x8*>(wsrc + (tap * 8 + 4 * ks) * 1024 + mb * 256);
        }
        int boff[3][2];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) boff[kx][ks] = LY::slot_off(32 * ph + l16 + kx, 4 * ks + g);
        int soff[2];   // staging offsets of the lane's 4 channels of channel block mb, pixel block 0 (block 1: + 2048)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int px = 32 * ph + l16;
            soff[mb] = px * 128 + 8 * (g & 1) + (((4 * ct + 2 * mb + (g >> 1)) ^ ((px >> 1) & 7)) << 4);
        }
        Acc16 a, bq, c;
#pragma unroll
        for (int i = 0; i < 4; ++i) a.v[i >> 1][i & 1] = bq.v[i >> 1][i & 1] = c.v[i >> 1][i & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 fb[6];
        auto issue3 = [&]() __attribute__((always_inline)) {
            if (!(P4C_EXP & 16)) {
#pragma unroll
                for (int o = 0; o < 3; ++o) fb[o] = *reinterpret_cast<const bf16x8*>(lring + boff[o >> 2][(o >> 1) & 1] + (o & 1) * 2048);
            }
        };
        if (wv == 0) P4C_STAMP(10);
        lds_barrier();   // the first interval's rows are staged
        if (wv == 0) P4C_STAMP(11);
        issue3();
#define P4C_ROW(HP, HQ, HS, P, Q, S)                                                                                        \
    {                                                                                                                       \
        const bool pf = ((m & (IV - 1)) != IV - 1) && (m != last);                                                          \
        const int delta = (m & 7) == 7 ? -7 * RROW : RROW;                                                                  \
        if (wv == 0) P4C_STAMP_ROW(300 + 2 * m);                                                                            \
        conv_row16<HP, HQ, HS>(A, P, Q, S, fb, boff, delta, pf, lring, lstg + (m & 7) * SROW, soff);                        \
        if (wv == 0) P4C_STAMP_ROW(301 + 2 * m);                                                                            \
        if (m == last) {                                                                                                    \
            P4C_STAMP(700 + 8 * (m >> 2) + 2 * wv);                                                                         \
            lds_barrier();                                                                                                  \
            P4C_STAMP(701 + 8 * (m >> 2) + 2 * wv);                                                                         \
        } else if ((m & (IV - 1)) == IV - 1) {                                                                              \
            P4C_STAMP(700 + 8 * (m >> 2) + 2 * wv);                                                                         \
            lds_barrier();                                                                                                  \
            P4C_STAMP(701 + 8 * (m >> 2) + 2 * wv);                                                                         \
            issue3();                                                                                                       \
        }                                                                                                                   \
        ++m;                                                                                                                \
    }
        P4C_ROW(false, false, true, a, a, a)
        P4C_ROW(false, true, true, a, a, bq)
        while (m + 3 <= R) {
            P4C_ROW(true, true, true, a, bq, c)
            P4C_ROW(true, true, true, bq, c, a)
            P4C_ROW(true, true, true, c, a, bq)
        }
#define P4C_ROTATE() { const Acc16 t_ = a; a = bq; bq = c; c = t_; }
        while (m < R) {
            P4C_ROW(true, true, true, a, bq, c)
            P4C_ROTATE()
        }
        P4C_ROW(true, true, false, a, bq, a)
        P4C_ROTATE()
        P4C_ROW(true, false, false, a, a, a)
#undef P4C_ROTATE
#undef P4C_ROW
        if (P4C_EXP & 16) { asm volatile("" ::"v"(A[0][0][0]), "v"(A[8][1][1])); }
    }
    if (wv == 0) P4C_STAMP(12);
}

static int rows_mfma_shape() {   // 32 (default): v_mfma_f32_32x32x16_bf16; P4C_ROWS_MFMA=16: v_mfma_f32_16x16x32_bf16 (A/B runs: 15 % faster in a bare MFMA loop, tools/diagnostics/mfma_power.hip, but 2-5 % slower here -- its 16-cycle MFMAs leave the memory-side waves half the issue slots)
    const char* e = diag_env("P4C_ROWS_MFMA");
    return (e && atoi(e) == 16) ? 16 : 32;
}

template <int MODE, bool BST, int KS, int IV = 4>
int launch_rows_mode(const __bf16* in, const __bf16* wp, const float* in_scale, const float* in_shift, __bf16* out, int out_cs,
                     float* stat_partial, int B, int H, int W, int nstrips, int nseg, hipStream_t stream, const BatchFin& fin,
                     const RingBwdStats& bst, const NormBwdCoef& nb = NormBwdCoef{}) {
    const RowsArgs args{in, wp, in_scale, in_shift, out, stat_partial, out_cs, H, W, H / nseg, H % nseg, fin.slots ? 1 : 0, 64, bst, nb, fin};
    if (KS == 1 || MODE == 4 || rows_mfma_shape() == 32) {
        P4C_TRY(ensure_dyn_smem((const void*)conv3x3_bf16_rows_kernel<MODE, BST, 32, KS, false, IV>, rows::Lay<32>::SMEM));
        hipLaunchKernelGGL((conv3x3_bf16_rows_kernel<MODE, BST, 32, KS, false, IV>), dim3(nseg, nstrips, B), dim3(512), rows::Lay<32>::SMEM, stream, args);
    } else if (KS == 3 && MODE != 4) {
        constexpr int M16 = MODE == 4 ? 0 : MODE;   // (MODE 4 never takes this branch: keeps the instantiation list short)
        P4C_TRY(ensure_dyn_smem((const void*)conv3x3_bf16_rows_kernel<M16, BST, 16, 3>, rows::Lay<16>::SMEM));
        hipLaunchKernelGGL((conv3x3_bf16_rows_kernel<M16, BST, 16, 3>), dim3(nseg, nstrips, B), dim3(512), rows::Lay<16>::SMEM, stream, args);
    }
    return P4C_OK;
}

template <int KS>
int launch_rows_ks(const __bf16* in, const __bf16* wp, const float* in_scale, const float* in_shift, int in_relu, __bf16* out, int out_cs,
                   float* stat_partial, int B, int H, int W, int nstrips, int nseg, hipStream_t stream, const BatchFin& fin,
                   const RingBwdStats* bst, const NormBwdCoef* nb) {
    if (nb && KS == 3) {   // the operand is g and y of a normalisation backward: dY is formed while the rows are staged
        constexpr int M = KS == 3 ? 4 : 0;
        // both roles in one launch: two rows per interval (half-size loader images: see the kernel's IV)
        if (bst) return launch_rows_mode<M, true, KS, 2>(in, wp, nullptr, nullptr, out, out_cs, stat_partial, B, H, W, nstrips, nseg, stream, fin, *bst, *nb);
        return launch_rows_mode<M, false, KS>(in, wp, nullptr, nullptr, out, out_cs, stat_partial, B, H, W, nstrips, nseg, stream, fin, RingBwdStats{}, *nb);
    }
    if (bst) return launch_rows_mode<0, true, KS>(in, wp, nullptr, nullptr, out, out_cs, stat_partial, B, H, W, nstrips, nseg, stream, fin, *bst);
    if (in_scale)
        return in_relu ? launch_rows_mode<2, false, KS>(in, wp, in_scale, in_shift, out, out_cs, stat_partial, B, H, W, nstrips, nseg, stream, fin, RingBwdStats{})
                       : launch_rows_mode<3, false, KS>(in, wp, in_scale, in_shift, out, out_cs, stat_partial, B, H, W, nstrips, nseg, stream, fin, RingBwdStats{});
    return in_relu ? launch_rows_mode<1, false, KS>(in, wp, in_scale, in_shift, out, out_cs, stat_partial, B, H, W, nstrips, nseg, stream, fin, RingBwdStats{})
                   : launch_rows_mode<0, false, KS>(in, wp, in_scale, in_shift, out, out_cs, stat_partial, B, H, W, nstrips, nseg, stream, fin, RingBwdStats{});
}

}  // namespace

template <int KS>
int launch_rows_compact(const __bf16* in, int in_cs, const __bf16* wp, __bf16* out, int out_cs, int B, int H, int W, int nstrips, int nseg,
                        hipStream_t stream) {
    const RowsArgs args{in, wp, nullptr, nullptr, out, nullptr, out_cs, H, W, H / nseg, H % nseg, 0, in_cs, RingBwdStats{}, NormBwdCoef{}, BatchFin{}};
    P4C_TRY(ensure_dyn_smem((const void*)conv3x3_bf16_rows_kernel<0, false, 32, KS, true>, rows::Lay<32>::SMEM));
    hipLaunchKernelGGL((conv3x3_bf16_rows_kernel<0, false, 32, KS, true>), dim3(nseg, nstrips, B), dim3(512), rows::Lay<32>::SMEM, stream, args);
    return P4C_OK;
}

// Strip / segment geometry of a launch: nstrips 64-pixel strips per sample, each cut into nseg row segments (one workgroup
// each).  nseg minimises (waves of workgroups over the CUs) x (rows per segment + the fixed cost of a workgroup, ~6 rows).
void conv_rows_geometry(int B, int H, int W, int* nstrips_out, int* nseg_out) {
    const int nstrips = (W + rows::SW - 1) / rows::SW;
    int best = 1;
    if (const char* e = diag_env("P4C_ROWS_NSEG")) {
        best = atoi(e);
    } else {
        const int cus = num_cus();
        double best_cost = 1e30;
        const int max_seg = H / 2 > 1 ? H / 2 : 1;   // (small maps: many two-row segments -- the halo re-reads cost less than idle CUs)
        for (int n = 1; n <= max_seg; ++n) {
            const int64_t wgs = (int64_t)B * nstrips * n;
            const int64_t waves = (wgs + cus - 1) / cus;
            const double cost = (double)waves * ((H + n - 1) / n + 6.0);
            if (cost < best_cost - 1e-9) { best_cost = cost; best = n; }
        }
    }
    if (best < 1) best = 1;
    if (best > H / 2) best = H / 2 > 0 ? H / 2 : 1;
    *nstrips_out = nstrips;
    *nseg_out = best;
}

bool conv_bf16_is_rows(int storage, int CI, int ks, int m_blocks, int out_cs, int B, int H, int W) {
    const char* e = diag_env("P4C_NO_ROWS");   // (read per call: the A/B scripts and the parity tests switch it)
    const bool off = e && e[0] == '1';
    const char* e1 = diag_env("P4C_NO_ROWS_1X1");
    if (ks == 1 && e1 && e1[0] == '1') return false;
    return !off && storage == P4C_BF16 && CI == 64 && (ks == 3 || ks == 1) && m_blocks == 1 && out_cs % 8 == 0 && W > 32 && H >= 8 &&
           (int64_t)H * W * out_cs * 2 < (int64_t)1 << 31 && (int64_t)H * W * 128 < (int64_t)1 << 31;
}

bool conv_bf16_norm_bwd_fused_ok(int storage, int CI, int B, int H, int W) {
    const char* e = diag_env("P4C_NO_FUSED_APPLY");   // (A/B switch and parity tests)
    if (e && e[0] == '1') return false;
    // data gradient on the row kernel, weight gradient on the role-split kernel (one 64-channel chunk, its sample limit)
    return CI == 64 && B <= 32 && conv_bf16_is_rows(storage, 64, 3, 1, 64, B, H, W);
}

// plain convolution with compact channel counts (see CPT): in (B,H,W,in_cs) -> out (B,H,W,out_cs), weights prepared for 64 x 64
bool conv_rows_compact_ok(int storage, int in_cs, int out_cs, int ks, int B, int H, int W) {
    return in_cs > 0 && out_cs > 0 && in_cs <= 64 && out_cs <= 64 && in_cs % 8 == 0 && out_cs % 8 == 0 &&
           conv_bf16_is_rows(storage, 64, ks, 1, 64, B, H, W);
}
int launch_conv_bf16_rows_compact(const void* in, int in_cs, const void* wp, int ks, void* out, int out_cs, int B, int H, int W,
                                  hipStream_t stream) {
    if (!conv_rows_compact_ok(P4C_BF16, in_cs, out_cs, ks, B, H, W))
        return fail(P4C_ERR_UNSUPPORTED, "conv_bf16_rows_compact: unsupported shape (%d -> %d channels, ks %d, %dx%dx%d)", in_cs, out_cs, ks, B, H, W);
    int nstrips, nseg;
    conv_rows_geometry(B, H, W, &nstrips, &nseg);
    const int rc = ks == 3 ? launch_rows_compact<3>((const __bf16*)in, in_cs, (const __bf16*)wp, (__bf16*)out, out_cs, B, H, W, nstrips, nseg, stream)
                           : launch_rows_compact<1>((const __bf16*)in, in_cs, (const __bf16*)wp, (__bf16*)out, out_cs, B, H, W, nstrips, nseg, stream);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_LAUNCH("conv_bf16_rows_compact");
    return P4C_OK;
}

// the first 64 channels of pixels that are `pixel_channels` channels apart (the CPT instantiation: its input channel count doubles as
// the pixel stride, every channel octet below 64 exists)
int launch_conv_bf16_rows_wide_pixels(const void* in, int pixel_channels, const void* wp, void* out, int B, int H, int W, hipStream_t stream) {
    if (pixel_channels < 64 || pixel_channels % 8 || !conv_bf16_is_rows(P4C_BF16, 64, 3, 1, 64, B, H, W) ||
        (int64_t)H * W * pixel_channels * 2 >= (int64_t)1 << 31)
        return fail(P4C_ERR_UNSUPPORTED, "conv_bf16_rows_wide_pixels: unsupported shape (%d-channel pixels, %dx%dx%d)", pixel_channels, B, H, W);
    int nstrips, nseg;
    conv_rows_geometry(B, H, W, &nstrips, &nseg);
    const int rc = launch_rows_compact<3>((const __bf16*)in, pixel_channels, (const __bf16*)wp, (__bf16*)out, 64, B, H, W, nstrips, nseg, stream);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_LAUNCH("conv_bf16_rows_wide_pixels");
    return P4C_OK;
}

int conv_rows_stat_slots(int B, int H, int W) {
    int nstrips, nseg;
    conv_rows_geometry(B, H, W, &nstrips, &nseg);
    return nstrips * nseg * 4;
}

int launch_conv3x3_bf16_rows(const void* inv, const void* wpv, int ks, const float* in_scale, const float* in_shift, int in_relu,
                             void* outv, int out_cs, float* stat_partial, int B, int H, int W, hipStream_t stream, const BatchFin* finp,
                             const RingBwdStats* bst, int* nblk_out, const NormBwdCoef* nb) {
    const __bf16* in = (const __bf16*)inv;
    const __bf16* wp = (const __bf16*)wpv;
    __bf16* out = (__bf16*)outv;
    int nstrips, nseg;
    conv_rows_geometry(B, H, W, &nstrips, &nseg);
    BatchFin fin{};
    if (finp && stat_partial) { fin = *finp; fin.slots = stat_partial; }
    if (bst) {
        P4C_CHECK_ARG(!in_scale && !in_relu && stat_partial && !finp && out_cs == 64 && nblk_out,
                      "conv_bf16_rows: backward statistics need a plain 64-channel launch with a partial buffer");
        *nblk_out = nstrips * nseg * 4;
    }
    const int tag = ks == 3 ? P4C_PROF_CONV3X3_C64 : 0;
    if (tag) prof_begin(tag, (int64_t)B * H * W, stream);
    P4C_CHECK_ARG(!nb || (ks == 3 && !in_scale && !in_relu && nb->y), "conv_bf16_rows: NormBwdCoef needs a plain 3x3 launch");
    const int rc = ks == 3 ? launch_rows_ks<3>(in, wp, in_scale, in_shift, in_relu, out, out_cs, stat_partial, B, H, W, nstrips, nseg, stream, fin, bst, nb)
                           : launch_rows_ks<1>(in, wp, in_scale, in_shift, in_relu, out, out_cs, stat_partial, B, H, W, nstrips, nseg, stream, fin, bst, nullptr);
    if (tag) prof_end(tag, stream);
    if (rc != P4C_OK) return rc;
    P4C_CHECK_LAUNCH("conv_bf16_rows");
    return P4C_OK;
}

}  // namespace p4c
#ifdef P4C_STAMPS
extern "C" int p4c_debug_set_rows_stamps(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(p4c::g_rows_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif
