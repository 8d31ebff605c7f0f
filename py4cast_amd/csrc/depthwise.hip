// Ghost module's "cheap operation" (mfai's HalfUNet with use_ghost: config/CLI/model/halfunet.yaml:22; Han et al. 2020): a depthwise
// 3x3 convolution over the 32 channels the primary convolution produced, concatenated behind them.  Features-last tensors of 64
// channels: channels 0..31 = primary features, 32..63 = their depthwise images.  HBM-bound streaming kernels (the 3x3 halo comes
// from L2): forward, data gradient, weight gradient (per-workgroup partials, summed by the caller in a fixed order).
#include "common.hpp"

namespace p4c {
namespace dw {

constexpr int CH = 32;      // channels of each half

__device__ __forceinline__ p4c_f32x4 ld4(const float* p) { return *reinterpret_cast<const p4c_f32x4*>(p); }
__device__ __forceinline__ p4c_f32x4 ld4(const bf16* p) { return load4f(p); }

// out[p, :32] = in[p, :32];  out[p, 32 + c] = sum_tap w[c][tap] * in[p + tap, c]       (zero padding)
template <typename T>
__global__ void __launch_bounds__(256) fwd_kernel(const T* __restrict__ in, const float* __restrict__ w, T* __restrict__ out, int B, int H, int W) {
    __shared__ float lw[CH * 9];
    for (int i = threadIdx.x; i < CH * 9; i += 256) lw[i] = w[i];
    __syncthreads();
    const int64_t total = (int64_t)B * H * W * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 7), c = 4 * q;
        int64_t pix = i >> 3;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const T* base = in + pix * 64 + c;
        p4c_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
            for (int kx = -1; kx <= 1; ++kx) {
                if ((unsigned)(y + ky) >= (unsigned)H || (unsigned)(x + kx) >= (unsigned)W) continue;
                const p4c_f32x4 v = ld4(base + ((int64_t)ky * W + kx) * 64);
                const int tap = (ky + 1) * 3 + kx + 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(lw[(c + k) * 9 + tap], v[k], acc[k]);
            }
        store4f(out + pix * 64 + c, ld4(base));
        store4f(out + pix * 64 + CH + c, acc);
    }
}

// din[p, c] = dout[p, c] + sum_tap w[c][tap] * dout[p - tap, 32 + c];   din[p, 32 + c] = 0
template <typename T>
__global__ void __launch_bounds__(256) bwd_data_kernel(const T* __restrict__ dout, const float* __restrict__ w, T* __restrict__ din, int B, int H,
                                                       int W) {
    __shared__ float lw[CH * 9];
    for (int i = threadIdx.x; i < CH * 9; i += 256) lw[i] = w[i];
    __syncthreads();
    const int64_t total = (int64_t)B * H * W * 8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i & 7), c = 4 * q;
        int64_t pix = i >> 3;
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const T* base = dout + pix * 64 + c;
        p4c_f32x4 acc = ld4(base);
#pragma unroll
        for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
            for (int kx = -1; kx <= 1; ++kx) {
                // output pixel p' = p - tap read in[p' + tap] = in[p]: its gradient sits at p - (ky, kx)
                if ((unsigned)(y - ky) >= (unsigned)H || (unsigned)(x - kx) >= (unsigned)W) continue;
                const p4c_f32x4 v = ld4(base + CH - ((int64_t)ky * W + kx) * 64);
                const int tap = (ky + 1) * 3 + kx + 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(lw[(c + k) * 9 + tap], v[k], acc[k]);
            }
        store4f(din + pix * 64 + c, acc);
        store4f(din + pix * 64 + CH + c, p4c_f32x4{0.f, 0.f, 0.f, 0.f});
    }
}

// partial[blk][c][tap] = sum over the block's pixels of in[p + tap, c] * dout[p, 32 + c]
template <typename T>
__global__ void __launch_bounds__(256) wgrad_kernel(const T* __restrict__ in, const T* __restrict__ dout, float* __restrict__ partial, int B, int H,
                                                    int W) {
    __shared__ float red[32][CH * 9 + 1];
    const int q = threadIdx.x & 7, c = 4 * q, pl = threadIdx.x >> 3;   // 32 pixels per iteration, 8 channel quads
    float acc[4][9];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[k][t] = 0.f;
    const int64_t npix = (int64_t)B * H * W;
    for (int64_t pix = (int64_t)blockIdx.x * 32 + pl; pix < npix; pix += (int64_t)gridDim.x * 32) {
        const int x = (int)(pix % W);
        const int y = (int)((pix / W) % H);
        const p4c_f32x4 g = ld4(dout + pix * 64 + CH + c);
#pragma unroll
        for (int ky = -1; ky <= 1; ++ky)
#pragma unroll
            for (int kx = -1; kx <= 1; ++kx) {
                if ((unsigned)(y + ky) >= (unsigned)H || (unsigned)(x + kx) >= (unsigned)W) continue;
                const p4c_f32x4 v = ld4(in + (pix + (int64_t)ky * W + kx) * 64 + c);
                const int tap = (ky + 1) * 3 + kx + 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k][tap] = __builtin_fmaf(v[k], g[k], acc[k][tap]);
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int t = 0; t < 9; ++t) red[pl][(c + k) * 9 + t] = acc[k][t];
    __syncthreads();
    for (int i = threadIdx.x; i < CH * 9; i += 256) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += red[r][i];
        partial[(int64_t)blockIdx.x * CH * 9 + i] = s;
    }
}

static int grid_for(int64_t threads, int per_cu) {
    int64_t blocks = (threads + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * per_cu;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace dw
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_ghost_dw_wgrad_blocks(int B, int H, int W) {
    int64_t b = ((int64_t)B * H * W + 31) / 32;
    const int64_t cap = (int64_t)num_cus() * 4;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

extern "C" int p4c_ghost_dw_fwd(const void* in, const float* w, void* out, int dtype, int B, int H, int W, p4c_stream_t stream) {
    P4C_CHECK_ARG(in && w && out && B > 0 && H > 0 && W > 0, "p4c_ghost_dw_fwd: bad arguments");
    const int grid = dw::grid_for((int64_t)B * H * W * 8, 16);
    if (dtype == P4C_F32)
        hipLaunchKernelGGL(dw::fwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)in, w, (float*)out, B, H, W);
    else if (dtype == P4C_BF16)
        hipLaunchKernelGGL(dw::fwd_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16*)in, w, (bf16*)out, B, H, W);
    else
        return fail(P4C_ERR_INVALID, "p4c_ghost_dw_fwd: bad dtype");
    P4C_CHECK_LAUNCH("p4c_ghost_dw_fwd");
    return P4C_OK;
}

extern "C" int p4c_ghost_dw_bwd_data(const void* dout, const float* w, void* din, int dtype, int B, int H, int W, p4c_stream_t stream) {
    P4C_CHECK_ARG(dout && w && din && B > 0 && H > 0 && W > 0, "p4c_ghost_dw_bwd_data: bad arguments");
    const int grid = dw::grid_for((int64_t)B * H * W * 8, 16);
    if (dtype == P4C_F32)
        hipLaunchKernelGGL(dw::bwd_data_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)dout, w, (float*)din, B, H, W);
    else if (dtype == P4C_BF16)
        hipLaunchKernelGGL(dw::bwd_data_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16*)dout, w, (bf16*)din, B, H, W);
    else
        return fail(P4C_ERR_INVALID, "p4c_ghost_dw_bwd_data: bad dtype");
    P4C_CHECK_LAUNCH("p4c_ghost_dw_bwd_data");
    return P4C_OK;
}

extern "C" int p4c_ghost_dw_wgrad(const void* in, const void* dout, float* partial, int dtype, int B, int H, int W, p4c_stream_t stream) {
    P4C_CHECK_ARG(in && dout && partial && B > 0 && H > 0 && W > 0, "p4c_ghost_dw_wgrad: bad arguments");
    const int grid = p4c_ghost_dw_wgrad_blocks(B, H, W);
    if (dtype == P4C_F32)
        hipLaunchKernelGGL(dw::wgrad_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)in, (const float*)dout, partial, B, H, W);
    else if (dtype == P4C_BF16)
        hipLaunchKernelGGL(dw::wgrad_kernel<bf16>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16*)in, (const bf16*)dout, partial, B, H, W);
    else
        return fail(P4C_ERR_INVALID, "p4c_ghost_dw_wgrad: bad dtype");
    P4C_CHECK_LAUNCH("p4c_ghost_dw_wgrad");
    return P4C_OK;
}
