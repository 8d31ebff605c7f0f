// up_bwd_x4 on the matrix cores: the x pass of the adjoint of HalfUNet's four bilinear up-samplings (mfai's HalfUNet decoder merge,
// `F.interpolate(..., mode="bilinear")` of levels 1..4 summed at full resolution; backward under py4cast/lightning.py:591-596), all four
// levels from ONE read of dS:
//     Tx_k[b,y,X,:] = sum_x wx_k(x,X) dS[b,y,x,:]          k = 1..4, s = 2^k, Tx_k is (B,H,W/s,64)
// For a strip of 64 pixels of one row this is a (60 outputs x 80 pixels) banded matrix times the (80 pixels x 64 channels) slab of dS:
// the VALU kernel (norm_pool.hip: 32 taps x 16-byte LDS read + 4 FMAs per thread) ran at 2.2-2.4 TB/s (55-59 us for 130 MB at 2 x 512 x 512)
// with the vector ALU, the LDS and the memory pipe adding up; this one takes 44 us (3.0 TB/s; deeper prefetch does not change it):
// the slab is staged ONCE as bf16, read with transposed LDS reads, and the taps are 5 MFMAs per wave.
// The weights are multiples of 1/32 (exact in bf16) and dS is bf16, so every product is exact in fp32 -- the result differs from
// the VALU kernel's only in the order of the fp32 sum.
//
// Workgroup = 256 threads = ROWS consecutive rows (b*H + y is flat; the outputs are indexed by it too) of one 64-pixel strip.  Per row:
//   1. the slab (8 halo + 64 + 8 halo pixels, prefetched a row ahead as 16-byte loads) goes to an LDS tile [pixel][64 ch];   barrier
//   2. wave (ct, ot): C[32 channels of block ct][32 outputs of tile ot] = dS^T (transposed LDS reads) x W^T (stationary registers),
//      K = 80 pixels = 5 MFMAs 32x32x16 -> bf16 -> LDS staging tile [output][64 ch];                                           barrier
//   3. the staging tile leaves as whole 16-byte slots (coalesced) to the four Tx arrays.
// Output tile 0 = the 32 outputs of level 1; tile 1 = 16 (level 2) + 8 (level 3) + 4 (level 4) + 4 idle rows.
// Border columns as the VALU kernel: X = 0 takes all the weight of the pixels left of its centre, X = Wk-1 of those right of it.
#include "kernels.hpp"

namespace p4c {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

namespace ubm {
constexpr int ROWS = 8;
constexpr int AHEAD = 2;                // rows of loads in flight per workgroup (1..4 measured: 43.8 / 44.9 / 46.1 / 45.6 us)
constexpr int NPX = 80;                 // 8 halo + 64 + 8 halo
constexpr int TILEB = NPX * 128;        // one slab of bf16 rows
constexpr int STAGEB = 64 * 128;        // staging tile: 2 output tiles x 32 rows
// slab tile, read by transposed LDS reads: the two 64-byte channel halves of a pixel swap when bit 1 of the pixel index is set
__device__ __forceinline__ int slot_off(int px, int c8) { return px * 128 + ((((c8 >> 2) ^ (px >> 1)) & 1) << 6) + ((c8 & 3) << 4); }
// staging tile, written 8 bytes at a time from accumulator layout, read 16 bytes at a time
__device__ __forceinline__ int g_off(int m, int c8) { return m * 128 + ((c8 ^ ((m >> 1) & 7)) << 4); }
}  // namespace ubm

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

struct UpBwdMfmaArgs {
    const __bf16* dS;
    __bf16* tx[4];
    int64_t rows;
    int W;
};

// level (1..4) and column (within the strip) of staging row m, m < 60
__device__ __forceinline__ void row_level(int m, int& k, int& xl) {
    if (m < 32) { k = 1; xl = m; }
    else if (m < 48) { k = 2; xl = m - 32; }
    else if (m < 56) { k = 3; xl = m - 48; }
    else { k = 4; xl = m - 56; }
}

__global__ void __launch_bounds__(256) up_bwd_x4_mfma_kernel(UpBwdMfmaArgs a) {
    using namespace ubm;
    __shared__ __attribute__((aligned(16))) char smem[2 * TILEB + STAGEB];
    char* stage = smem + 2 * TILEB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int W = a.W, strips = W / 64;
    const int64_t row0 = (int64_t)(blockIdx.x / strips) * ROWS;
    const int x0 = (int)(blockIdx.x % strips) * 64;
    const int nrows = (int)(a.rows - row0 < ROWS ? a.rows - row0 : ROWS);

    // ---- loader role: 640 slots of 16 bytes per slab, thread t -> slots t, t + 256, t + 512
    int lsrc[3], ldst[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int sl = tid + 256 * i, px = sl >> 3, c8 = sl & 7, x = x0 - 8 + px;
        const bool ok = sl < NPX * 8 && x >= 0 && x < W;
        lsrc[i] = ok ? x * 128 + c8 * 16 : -1;
        ldst[i] = sl < NPX * 8 ? slot_off(px, c8) : -1;
    }
    const char* src = reinterpret_cast<const char*>(a.dS);
    // AHEAD rows in flight per workgroup (register sets): the loop is otherwise one memory round trip per row
    u32x4 pre[AHEAD][3];
    auto load = [&](u32x4 (&d)[3], int64_t row) __attribute__((always_inline)) {
        const char* rp = src + row * W * 128;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            d[i] = u32x4{0u, 0u, 0u, 0u};
            if (lsrc[i] >= 0) d[i] = *reinterpret_cast<const u32x4*>(rp + lsrc[i]);
        }
    };
#pragma unroll
    for (int q = 0; q < AHEAD; ++q)
        if (q < nrows) load(pre[q], row0 + q);

    // ---- matrix role: wave = channel block ct x output tile ot; B operand (the weights of my output, 8 pixels per k-step) stationary
    const int r = lane & 31, h = lane >> 5;
    const int ct = wv >> 1, ot = wv & 1;
    bf16x8 Bw[5];
    {
        const int m = ot * 32 + r;
        int k = 1, xl = 0;
        if (m < 60) row_level(m, k, xl);
        const int s = 1 << k, Wk = W >> k, X = x0 / s + xl;
        const float inv = 1.0f / (float)s;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int p = 16 * ks + 8 * h + j;
                const int jj = p - 8 - s * xl + s / 2;   // tap index: pixel x = s X - s/2 + jj
                float w = 0.f;
                if (m < 60 && jj >= 0 && jj < 2 * s) {
                    w = 1.f - fabsf(((float)jj + 0.5f) * inv - 1.f);
                    if ((X == 0 && jj < s) || (X == Wk - 1 && jj >= s)) w = 1.f;
                }
                Bw[ks][j] = (__bf16)w;
            }
    }
    const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
    const int xoff = (8 * h + tq) * 128 + (((ct ^ (tq >> 1)) & 1) << 6) + (tg * 16 + tp * 4) * 2;   // A operand: channel block ct
    int soff[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) soff[g] = g_off(ot * 32 + r, 4 * ct + g) + 8 * h;

    // ---- store role: 480 staged slots, thread t -> slots t, t + 256
    int64_t obase[2], ostride[2];
    int ooff[2];
    __bf16* optr[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sl = tid + 256 * i, m = sl >> 3, c8 = sl & 7;
        optr[i] = nullptr; obase[i] = 0; ostride[i] = 0; ooff[i] = 0;
        if (m < 60) {
            int k, xl;
            row_level(m, k, xl);
            const int Wk = W >> k;
            optr[i] = a.tx[k - 1];
            obase[i] = (int64_t)((x0 >> k) + xl) * 64 + c8 * 8;
            ostride[i] = (int64_t)Wk * 64;
            ooff[i] = g_off(m, c8);
        }
    }

#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
        if (rr >= nrows) break;
        char* tile = smem + (rr & 1) * TILEB;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (ldst[i] >= 0) *reinterpret_cast<u32x4*>(tile + ldst[i]) = pre[rr % AHEAD][i];
        lds_barrier();
        if (rr + AHEAD < nrows) load(pre[rr % AHEAD], row0 + rr + AHEAD);
        f32x16 c;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            union { s16x4 q[2]; bf16x8 v; } ua;
            ua.q[0] = tr_read(tile + ks * 2048 + xoff);
            ua.q[1] = tr_read(tile + ks * 2048 + xoff + 512);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ua.v, Bw[ks], c, 0, 0, 0);
        }
        // C[channel][output]: lane = output r (+ half h), register quad g -> channels 32 ct + 8 g + 4 h .. + 3
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x2 lo = {c[4 * g], c[4 * g + 1]}, hi = {c[4 * g + 2], c[4 * g + 3]};
            u32x2 o;
            o[0] = __builtin_bit_cast(unsigned int, __builtin_convertvector(lo, bf16x2));
            o[1] = __builtin_bit_cast(unsigned int, __builtin_convertvector(hi, bf16x2));
            *reinterpret_cast<u32x2*>(stage + soff[g]) = o;
        }
        lds_barrier();
        const int64_t row = row0 + rr;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (optr[i]) *reinterpret_cast<u32x4*>(optr[i] + row * ostride[i] + obase[i]) = *reinterpret_cast<const u32x4*>(stage + ooff[i]);
    }
}

}  // namespace

bool up_bwd_x4_mfma_ok(int B, int H, int W) {
    return B > 0 && H > 0 && W >= 64 && W % 64 == 0 && (int64_t)W * 128 < ((int64_t)1 << 31);
}

int launch_up_bwd_x4_mfma(const void* dS, int64_t rows, int W, void* const* tx, hipStream_t stream) {
    UpBwdMfmaArgs a;
    a.dS = (const __bf16*)dS;
    for (int k = 0; k < 4; ++k) a.tx[k] = (__bf16*)tx[k];
    a.rows = rows;
    a.W = W;
    const int64_t groups = (rows + ubm::ROWS - 1) / ubm::ROWS;
    hipLaunchKernelGGL(up_bwd_x4_mfma_kernel, dim3((unsigned)(groups * (W / 64))), dim3(256), 0, stream, a);
    P4C_CHECK_LAUNCH("up_bwd_x4_mfma");
    return P4C_OK;
}

}  // namespace p4c
