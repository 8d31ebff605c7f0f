// out_conv_bwd: the whole backward of the network's 1x1 output convolution in ONE pass over dy and the last block's raw output y
// (what autograd does for mfai's `outconv = nn.Conv2d(num_filters, out_channels, 1)` under py4cast/lightning.py:591-596, plus the head
// of the BatchNorm backward before it) -- data gradient AND weight gradient of the same convolution from one read of its operands:
//
//   dA[px][ci]  = sum_co W[co][ci] * dy[px][co]                                  (data gradient: dA of the last block's normalisation)
//   S1[ci], S2[ci] = sums over pixels of g and g * xhat, g = dA * [relu alive]   (pass 1 of that normalisation's backward)
//   dW[co][ci] += sum_px dy[px][co] * relu(y[px][ci] * scale + shift)            (weight gradient, K = pixels)
//
// It replaces three launches of the backward plan: the 1x1 data-gradient launch of the row kernel with its statistics
// (conv3x3_bf16_rows_kernel<0, true, 32, 1>: 86 us in the step, a streaming job on a kernel built for 3x3 row reuse: one workgroup per
// CU), and the 1x1 weight-gradient launch on the weight-gradient stream (59 us on half of the chip), which read dy and y a second time.
// HBM: dy + y read once, dA written once (200 MB at 2 x 512 x 512 against 335 MB) + one 16 KB partial per workgroup.
//
// A workgroup (256 threads, 4 waves) owns a contiguous run of 64-pixel tiles of one sample; per tile:
//   1. the dy rows (prefetched a tile ahead, coalesced 16-byte loads) go to an LDS tile;                                    barrier
//   2. matrix phase 1: wave = 32 ci x 32 px, 4 MFMAs 32x32x16 (A = W^T from the prepared data-gradient stream, stationary in
//      registers; B = 16-byte reads of the dy tile) -> bf16 -> LDS tile of dA;                                             barrier
//   3. element phase (thread = pixel x channel octet, coalesced): dA octet from LDS, y octet from HBM (prefetched) -> statistics,
//      dA to HBM, relu(norm(y)) to an LDS tile;                                                                             barrier
//   4. matrix phase 2: wave = 32 ci x 32 co, K = the tile's 64 pixels: transposed LDS reads (ds_read_b64_tr_b16) of the activation
//      tile and the dy tile, 4 MFMAs into 16 accumulator registers that live for the whole run.
// The dy tile is double-buffered (phase 4 of tile t overlaps phase 1 of tile t + 1 in other waves); LDS 32 KB, three workgroups per CU.
// Output: statistics slot [2][64] per workgroup (the layout norm_bwd_finalize reads), weight-gradient partial [64 ci][64 co] per
// workgroup (the layout wgrad_reduce reads).  dA is bit-identical to the row kernel's (same MFMA chain per element).
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
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace ocb {
constexpr int TP = 64;              // pixels per tile
constexpr int TILEB = TP * 128;     // one tile of 64 bf16 channels
constexpr int SMEM = 4 * TILEB;     // dy tile x 2 | dA tile | activation tile
// byte offset of channel octet c8 of pixel px in a tile read by transposed LDS reads: the two 64-byte channel halves of a pixel are
// swapped when bit 1 of the pixel index is set (conv_wgrad_rows.hip)
__device__ __forceinline__ int slot_off(int px, int c8) { return px * 128 + ((((c8 >> 2) ^ (px >> 1)) & 1) << 6) + ((c8 & 3) << 4); }
// ... and in the dA tile (written 8 bytes at a time from accumulator layout, read 16 bytes at a time): slot XOR (px >> 1) & 7
__device__ __forceinline__ int g_off(int px, int c8) { return px * 128 + ((c8 ^ ((px >> 1) & 7)) << 4); }
}  // namespace ocb

constexpr int OOB = 0x7fffffff;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

struct OutConvBwdArgs {
    const __bf16* dy;       // (B,N,64) bf16: gradient of the convolution's output (channels >= CO are zero)
    const __bf16* wp;       // prepared data-gradient operand stream of the 1x1 weight (M = ci, K = co)
    const __bf16* y;        // (B,N,64) bf16: raw output of the last block's convolution
    const float* scale;     // (B,64) each: that block's normalisation
    const float* shift;
    const float* mean;
    const float* rstd;
    __bf16* dA;             // (B,N,64) bf16 out
    float* stat_partial;    // [B][workgroups per sample][2][64] out
    float* wpartial;        // [B * workgroups per sample][64 ci][64 co] out
    int N, tiles_lo, tiles_rem;
};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
    out_conv_bwd_kernel(OutConvBwdArgs a) {
    using namespace ocb;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* dyT = smem;                  // 2 tiles
    char* gT = smem + 2 * TILEB;
    char* aT = smem + 3 * TILEB;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b = blockIdx.y, wg = blockIdx.x, nwg = gridDim.x;
    const int N = a.N;
    const int t0 = wg * a.tiles_lo + (wg < a.tiles_rem ? wg : a.tiles_rem);
    const int nt = a.tiles_lo + (wg < a.tiles_rem ? 1 : 0);

    const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(a.dy + (int64_t)b * N * 64, (unsigned int)N * 128u);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.y + (int64_t)b * N * 64, (unsigned int)N * 128u);
    const __amdgpu_buffer_rsrc_t rs_dA = make_rsrc(a.dA + (int64_t)b * N * 64, (unsigned int)N * 128u);

    // ---- element role: thread = (pixel pxl + 32 it, channel octet c8)
    const int c8 = tid & 7, pxl = tid >> 3;
    f32x2 sc[4], sh[4], mu[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        sc[k] = *reinterpret_cast<const f32x2*>(a.scale + b * 64 + 8 * c8 + 2 * k);
        sh[k] = *reinterpret_cast<const f32x2*>(a.shift + b * 64 + 8 * c8 + 2 * k);
        mu[k] = *reinterpret_cast<const f32x2*>(a.mean + b * 64 + 8 * c8 + 2 * k);
    }
    int eoff[2], toff[2], goff[2];   // byte offsets: in a tile of HBM rows, in a transposed-read tile, in the dA tile
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int px = pxl + 32 * it;
        eoff[it] = px * 128 + 16 * c8;
        toff[it] = slot_off(px, c8);
        goff[it] = g_off(px, c8);
    }
    float a1[8], a2[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) a1[q] = a2[q] = 0.f;

    // ---- matrix roles
    const int r = lane & 31, h = lane >> 5;
    const int ct = wv >> 1, ph = wv & 1;   // phase 1: ci block, pixel block;  phase 2: ci block (cit), co block (cot)
    bf16x8 A1[4];
    {
        const char* wsrc = reinterpret_cast<const char*>(a.wp) + (h * 64 + ct * 32 + r) * 16;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) A1[ks] = *reinterpret_cast<const bf16x8*>(wsrc + ks * 2048);
    }
    int boff[4], soff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) boff[ks] = slot_off(32 * ph + r, 2 * ks + h);
#pragma unroll
    for (int g = 0; g < 4; ++g) soff[g] = g_off(32 * ph + r, 4 * ct + g) + 8 * h;
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int cb = (tg * 16 + tp * 4) * 2;
    const int xoff = (8 * h + tq) * 128 + (((ct ^ (tq >> 1)) & 1) << 6) + cb;   // activation operand: ci block ct
    const int doff = (8 * h + tq) * 128 + (((ph ^ (tq >> 1)) & 1) << 6) + cb;   // dy operand: co block ph
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    u32x4 dyv[2], yv[2], dyn[2], yn[2];
    auto load = [&](u32x4 (&d)[2], u32x4 (&yy)[2], int t) __attribute__((always_inline)) {
        const int so = (t0 + t) * TILEB;
        const int npx = t < nt ? N - (t0 + t) * TP : 0;   // pixels of the sample from this tile on (the last tile may be partial)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int vo = pxl + 32 * it < npx ? eoff[it] : OOB;   // (an out-of-range offset loads zeros)
            d[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, vo, so, 0);
            yy[it] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, vo, so, 0);
        }
    };
    load(dyv, yv, 0);
    for (int t = 0; t < nt; ++t) {
        char* dyt = dyT + (t & 1) * TILEB;
#pragma unroll
        for (int it = 0; it < 2; ++it) *reinterpret_cast<u32x4*>(dyt + toff[it]) = dyv[it];
        lds_barrier();
        load(dyn, yn, t + 1);
        // ---- matrix phase 1: dA[ci][px] for this wave's 32 x 32 block
        {
            bf16x8 fb[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb[ks] = *reinterpret_cast<const bf16x8*>(dyt + boff[ks]);
            f32x16 c;
#pragma unroll
            for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1[ks], fb[ks], c, 0, 0, 0);
            // C[ci][px]: lane = pixel r (+ half h), register quad g -> channels 32 ct + 8 g + 4 h .. + 3
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x2 lo = {c[4 * g], c[4 * g + 1]}, hi = {c[4 * g + 2], c[4 * g + 3]};
                u32x2 o;
                o[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(lo, bf16x2));
                o[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(hi, bf16x2));
                *reinterpret_cast<u32x2*>(gT + soff[g]) = o;
            }
        }
        lds_barrier();
        // ---- element phase
        {
            const int so = (t0 + t) * TILEB, npx = N - (t0 + t) * TP;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const u32x4 gv = *reinterpret_cast<const u32x4*>(gT + goff[it]);
                u32x4 av;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float ylo = __builtin_bit_cast(float, yv[it][q] << 16);
                    const float yhi = __builtin_bit_cast(float, yv[it][q] & 0xffff0000u);
                    const float nlo = __builtin_fmaf(ylo, sc[q].x, sh[q].x), nhi = __builtin_fmaf(yhi, sc[q].y, sh[q].y);
                    const float glo = nlo > 0.f ? __builtin_bit_cast(float, gv[q] << 16) : 0.f;
                    const float ghi = nhi > 0.f ? __builtin_bit_cast(float, gv[q] & 0xffff0000u) : 0.f;
                    a1[2 * q] += glo; a2[2 * q] = __builtin_fmaf(glo, ylo - mu[q].x, a2[2 * q]);
                    a1[2 * q + 1] += ghi; a2[2 * q + 1] = __builtin_fmaf(ghi, yhi - mu[q].y, a2[2 * q + 1]);
                    const f32x2 nv = {nlo, nhi};
                    const s16x2 z = {0, 0};   // relu on the rounded value: negative floats are negative int16
                    av[q] = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(
                                                                 __builtin_bit_cast(s16x2, __builtin_convertvector(nv, bf16x2)), z));
                }
                __builtin_amdgcn_raw_buffer_store_b128(gv, rs_dA, pxl + 32 * it < npx ? eoff[it] : OOB, so, 0);   // (out of range: dropped)
                *reinterpret_cast<u32x4*>(aT + toff[it]) = av;
            }
        }
        lds_barrier();
        // ---- matrix phase 2: dW[ci][co] += activation^T x dy over the tile's 64 pixels
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            union { s16x4 q[2]; bf16x8 v; } ua, ub;
            ua.q[0] = tr_read(aT + s * 2048 + xoff);
            ua.q[1] = tr_read(aT + s * 2048 + xoff + 512);
            ub.q[0] = tr_read(dyt + s * 2048 + doff);
            ub.q[1] = tr_read(dyt + s * 2048 + doff + 512);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, ub.v, acc, 0, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) { dyv[it] = dyn[it]; yv[it] = yn[it]; }
    }

    // ---- weight-gradient partial: C[ci][co]: lane = co (r), register i -> ci = (i & 3) + 8 (i >> 2) + 4 h
    {
        float* pbase = a.wpartial + ((int64_t)b * nwg + wg) * 4096;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ci = ct * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            pbase[ci * 64 + ph * 32 + r] = acc[i];
        }
    }
    // ---- statistics slot of this workgroup: lanes with the same channel octet, then the four waves in a fixed order
    lds_barrier();   // (every wave is past its last tile: the tiles are free)
    float* lred = reinterpret_cast<float*>(smem);   // [4 waves][128]
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float u = a1[q], v = a2[q];
        u += __shfl_xor(u, 8); v += __shfl_xor(v, 8);
        u += __shfl_xor(u, 16); v += __shfl_xor(v, 16);
        u += __shfl_xor(u, 32); v += __shfl_xor(v, 32);
        if (lane < 8) { lred[wv * 128 + 8 * c8 + q] = u; lred[wv * 128 + 64 + 8 * c8 + q] = v; }
    }
    lds_barrier();
    if (tid < 128) {
        float v = (lred[tid] + lred[128 + tid]) + (lred[256 + tid] + lred[384 + tid]);
        if (tid >= 64) v *= a.rstd[b * 64 + tid - 64];   // sums of g * (y - mean) -> sums of g * xhat
        a.stat_partial[((int64_t)b * nwg + wg) * 128 + tid] = v;
    }
}

}  // namespace

// workgroups per sample: three per CU over the batch, one tile each at least, within the statistics buffer's slots
int out_conv_bwd_slots(int B, int64_t N) {
    const int64_t ntiles = (N + ocb::TP - 1) / ocb::TP;
    // (read ONCE per process: the workspace layout and every launch must agree on the slot count)
    static const int per_cu = [] {
        int v = 3;
        if (const char* e = diag_env("P4C_OCB_PER_CU")) { const int u = atoi(e); if (u > 0 && u <= 8) v = u; }
        return v;
    }();
    int64_t n = ((int64_t)num_cus() * per_cu + B - 1) / B;
    if (n > ntiles) n = ntiles;
    if (n > NORM_BWD_MAX_BLOCKS) n = NORM_BWD_MAX_BLOCKS;
    if (n < 1) n = 1;
    return (int)n;
}

bool out_conv_bwd_ok(int storage, int B, int64_t N) {
    const char* e = diag_env("P4C_FUSED_OUT_BWD");   // (read per call: A/B scripts and the parity tests switch it)
    if (e && e[0] == '0') return false;
    return storage == P4C_BF16 && B > 0 && N > 0 && N * 128 < ((int64_t)1 << 31);
}

// wpartial: B * out_conv_bwd_slots(B, N) * 4096 floats; stat_partial: B * slots * 128 floats; *nblk_out = slots per sample
int launch_out_conv_bwd(const void* dy, const void* wp_dgrad, const void* y, const float* scale, const float* shift, const float* mean,
                        const float* rstd, void* dA, float* stat_partial, float* wpartial, int B, int64_t N, hipStream_t stream,
                        int* nblk_out) {
    const int nwg = out_conv_bwd_slots(B, N);
    const int64_t ntiles = (N + ocb::TP - 1) / ocb::TP;
    const OutConvBwdArgs a{(const __bf16*)dy, (const __bf16*)wp_dgrad, (const __bf16*)y, scale, shift, mean, rstd, (__bf16*)dA,
                           stat_partial, wpartial, (int)N, (int)(ntiles / nwg), (int)(ntiles % nwg)};
    hipLaunchKernelGGL(out_conv_bwd_kernel, dim3(nwg, B), dim3(256), ocb::SMEM, stream, a);
    P4C_CHECK_LAUNCH("out_conv_bwd");
    if (nblk_out) *nblk_out = nwg;
    return P4C_OK;
}

}  // namespace p4c
