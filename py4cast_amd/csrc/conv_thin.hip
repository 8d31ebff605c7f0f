// The first convolution of the HalfUNet plan at the benchmark's 69 input channels (60 state + 5 forcing + 4 static features:
// py4cast/lightning.py:256-261, 711-767), as TWO launches instead of one generic K = 96 launch (round 6; bf16 flavour only):
//
//   1. the row-streaming kernel of conv_rows.hip on input channels 0..63 (a 64 -> 64 launch like every other convolution of the
//      network, reading the 96-channel pixels of x with a 192-byte stride): y1 = conv3x3(x[..., :64]);
//   2. first_conv_tail_kernel (here): y = bf16(y1 + conv3x3(x[..., 64:72])) and the BatchNorm / GroupNorm sums of y -- an HBM-bound
//      pass over y (read + write) that carries the 5 channels beyond 64 as a K = 80 product per 32 pixels (9 taps x 8 channels + one
//      zero tap): 10 v_mfma_f32_32x32x16_bf16 per 32 pixels x 64 output channels against 108 for a 96-channel launch.
//
// The generic role-split kernel it replaces (conv_fwd_bf16_kernel<96, 3>) took 91-100 us per launch x 3 AR steps for 1.5 x the matrix
// work of a 64-channel launch that takes 39-43.  Numerics: y1 is rounded to bf16 before the tail adds its fp32 product
// -- a second rounding that the one-launch form does not have; inside every bar the bf16 flavour is held to (the fp32 flavour keeps the
// one-launch kernel).  The backward is unchanged (data gradient: a 64 -> 64 row launch on the state channels; weight gradient: the
// row-streaming kernel's full chunk + thin chunk).
#include "kernels.hpp"

namespace p4c {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int OOB = 0x7fffffff;
constexpr int SP = 68;                   // floats per pixel of a wave's staging tile (64 + 4: 16-byte aligned rows off the bank period)
constexpr int WIMG_BYTES = 2 * 5 * 64 * 16;

struct ThinArgs {
    const __bf16* x;        // (B, H, W, x_cs) bf16
    const float* w;         // fp32 master weight [64][cin][3][3]
    __bf16* y;              // (B, H, W, 64): in = y1, out = y (in place: every element is read and written by one lane)
    float* stat_partial;    // [B][nblk * 4][2][64] sums / sums of squares of the stored y, or NULL
    int x_cs, c0, nc, cin;  // pixel stride of x in channels; first tail channel (64); tail channels (1..8); input channels of w
    int H, W, nblk;
};

__device__ __forceinline__ float bf_lo(unsigned int w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned int w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ unsigned int pack2(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
}

__global__ void __launch_bounds__(256) first_conv_tail_kernel(ThinArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[WIMG_BYTES + 4 * 32 * SP * 4];
    __bf16* wimg = reinterpret_cast<__bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 31, h = lane >> 5;
    float* tile = reinterpret_cast<float*>(smem + WIMG_BYTES) + wv * 32 * SP;
    // A operands: image[(ct * 5 + s) * 64 + lane] = 8 bf16 = W[co = 32 ct + (lane & 31)][c0 + j][tap 2 s + (lane >> 5)], zero for the
    // tenth tap and the channels beyond the real ones
    for (int idx = tid; idx < 2 * 5 * 64; idx += 256) {
        const int l = idx & 63, ts = idx >> 6;
        const int s = ts % 5, ct = ts / 5;
        const int co = 32 * ct + (l & 31), tap = 2 * s + (l >> 5);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)((tap < 9 && j < a.nc) ? a.w[((int64_t)co * a.cin + a.c0 + j) * 9 + tap] : 0.f);
        *reinterpret_cast<bf16x8*>(wimg + idx * 8) = o;
    }
    __syncthreads();
    bf16x8 A[2][5];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int s = 0; s < 5; ++s) A[ct][s] = *reinterpret_cast<const bf16x8*>(wimg + ((ct * 5 + s) * 64 + lane) * 8);

    const int b = blockIdx.y, H = a.H, W = a.W;
    const int64_t npix = (int64_t)H * W;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(a.x + (int64_t)b * npix * a.x_cs), 0,
                                                                        (int)(npix * a.x_cs * 2), 0x00020000);
    __bf16* yb = a.y + (int64_t)b * npix * 64;
    const int ngroups = (int)(npix / 32);            // W % 32 == 0 (host): a group of 32 pixels lies in one image row
    const int wid = blockIdx.x * 4 + wv, nwaves = a.nblk * 4;
    const int prow = lane >> 3, c8 = lane & 7;       // row layout: pixel 8 it + prow of the group, channels 8 c8 .. +7
    float a1[8], a2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a1[q] = a2[q] = 0.f;

    // the NEXT group's operands (5 tap octets + 4 row pieces of y per lane) are in flight while the current group is multiplied, transposed
    // and stored: a wave walks ~8 groups, and without the prefetch every one of them exposed its full memory round trip (45-51 us per
    // launch for 134 MB in the first trace of the round)
    auto load_group = [&](int g, bf16x8 (&bop)[5], u32x4 (&yq)[4]) __attribute__((always_inline)) {
        const bool live = g < ngroups;
        const int p0 = live ? g * 32 : 0, yy = p0 / W, x0 = p0 - yy * W;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int tap = 2 * s + h, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
            const int sy = yy + dy, sx = x0 + r + dx;
            const bool ok = live && tap < 9 && (unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? ((sy * W + sx) * a.x_cs + a.c0) * 2 : OOB, 0, 0);
            bop[s] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) yq[it] = *reinterpret_cast<const u32x4*>(yb + ((int64_t)p0 + 8 * it + prow) * 64 + 8 * c8);
    };
    bf16x8 Bop[5], Bnx[5];
    u32x4 yv[4], ynx[4];
    load_group(wid, Bop, yv);
    for (int g = wid; g < ngroups; g += nwaves) {
        const int p0 = g * 32;
        load_group(g + nwaves, Bnx, ynx);
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[ct][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 5; ++s) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ct][s], Bop[s], acc[ct], 0, 0, 0);
        }
        // accumulator layout (lane = pixel r, registers 4 q + e = channels 32 ct + 8 q + 4 h + e) -> the wave's [pixel][channel] tile
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(tile + r * SP + 32 * ct + 8 * q + 4 * h) = f32x4{acc[ct][4 * q], acc[ct][4 * q + 1], acc[ct][4 * q + 2], acc[ct][4 * q + 3]};
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int px = 8 * it + prow;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(tile + px * SP + 8 * c8), t1 = *reinterpret_cast<const f32x4*>(tile + px * SP + 8 * c8 + 4);
            const float t[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = pack2(bf_lo(yv[it][k]) + t[2 * k], bf_hi(yv[it][k]) + t[2 * k + 1]);
                const float lo = bf_lo(o[k]), hi = bf_hi(o[k]);       // statistics of the STORED values
                a1[2 * k] += lo; a2[2 * k] = __builtin_fmaf(lo, lo, a2[2 * k]);
                a1[2 * k + 1] += hi; a2[2 * k + 1] = __builtin_fmaf(hi, hi, a2[2 * k + 1]);
            }
            *reinterpret_cast<u32x4*>(yb + ((int64_t)p0 + px) * 64 + 8 * c8) = o;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 5; ++s) Bop[s] = Bnx[s];
#pragma unroll
        for (int it = 0; it < 4; ++it) yv[it] = ynx[it];
    }
    if (a.stat_partial) {
        // one [2][64] slot per wave (also from a wave without a group: the finalize sums every slot)
        float* dst = a.stat_partial + ((int64_t)b * nwaves + wid) * 128;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float u = a1[q], v = a2[q];
            u += __shfl_xor(u, 8); v += __shfl_xor(v, 8);
            u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
            u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
            if (lane < 8) { dst[8 * c8 + q] = u; dst[64 + 8 * c8 + q] = v; }
        }
    }
}

int tail_blocks(int B, int H, int W) {
    const int64_t groups = (int64_t)H * W / 32;
    int64_t n = (groups + 3) / 4;
    if (n > 256) n = 256;              // <= 1024 statistics slots per sample (the plan's statistics buffer holds >= 4 x CUs)
    const int64_t cap = 2 * (int64_t)num_cus() / (B > 0 ? B : 1);
    if (n > cap && cap >= 1) n = cap;
    return n < 1 ? 1 : (int)n;
}

}  // namespace

bool first_conv_split_ok(int compute, int storage, int cin, int cin_pad, int B, int H, int W) {
    const char* e = diag_env("P4C_FIRST_CONV_SPLIT");       // (A/B switch of the diagnostic build: 0 = the one-launch K = 96 kernel)
    if (e && e[0] == '0') return false;
    return compute == P4C_BF16 && storage == P4C_BF16 && cin_pad == 96 && cin > 64 && cin <= 72 && W % 32 == 0 &&
           conv_bf16_is_rows(storage, 64, 3, 1, 64, B, H, W) && (int64_t)H * W * 192 < ((int64_t)1 << 31);
}

int first_conv_tail_slots(int B, int H, int W) { return tail_blocks(B, H, W) * 4; }

int launch_first_conv_tail(const void* x, int x_cs, int cin, const float* w, void* y, float* stat_partial, int B, int H, int W, hipStream_t stream) {
    ThinArgs a{(const __bf16*)x, w, (__bf16*)y, stat_partial, x_cs, 64, cin - 64, cin, H, W, tail_blocks(B, H, W)};
    hipLaunchKernelGGL(first_conv_tail_kernel, dim3(a.nblk, B), dim3(256), 0, stream, a);
    P4C_CHECK_LAUNCH("first_conv_tail");
    return P4C_OK;
}

}  // namespace p4c

// single-op entry point (tests): y (B,H,W,64) bf16 in place += conv3x3 of the channels 64 .. cin-1 of x (B,H,W,x_cs) bf16 with the fp32
// master weight w [64][cin][3][3]; stat_partial: p4c_first_conv_tail_slots(B,H,W) slots of [2][64] per sample, or NULL
extern "C" int p4c_first_conv_tail_slots(int B, int H, int W) { return p4c::first_conv_tail_slots(B, H, W); }
extern "C" int p4c_first_conv_tail(const void* x, int x_cs, int cin, const float* w, void* y, float* stat_partial, int B, int H, int W,
                                   p4c_stream_t stream) {
    P4C_CHECK_ARG(x && w && y, "p4c_first_conv_tail: NULL pointer");
    P4C_CHECK_ARG(cin > 64 && cin <= 72 && x_cs >= 72 && x_cs % 8 == 0 && W % 32 == 0 && B > 0 && H > 0,
                  "p4c_first_conv_tail: 65..72 input channels on pixels of >= 72 channels (multiple of 8), W a multiple of 32");
    return p4c::launch_first_conv_tail(x, x_cs, cin, w, y, stat_partial, B, H, W, p4c::as_stream(stream));
}
