// Bilinear up-sampling of features-last maps by an integer factor, torch.nn.functional.interpolate(mode="bilinear",
// align_corners=False) semantics, with the decoder's skip connection added on the way out -- UNETR++'s `linear_upsampling: true`
// UpBlocks (config/CLI/model/unetrpp.yaml:29; mfai's UnetrUpBlock: up-sample, then `out + skip`).  The library route was an NCHW
// kernel behind two layout copies, and its backward on features-last gradients an atomics kernel (not reproducible).  Here:
//   forward : out[b, oy, ox, :] = sum of the 2 x 2 source pixels' rows (+ skip[b, oy, ox, :]), 16 bytes per lane along the channels;
//   backward: GATHER form -- dx[b, iy, ix, :] = sum over the <= (3 s)^2 output pixels that read input pixel (iy, ix) of weight * dout,
//             in a fixed order (bit-identical reruns, no atomics); the skip's gradient is dout itself.
#include "common.hpp"

namespace p4c {
namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void unpack8(u32x4 w, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __builtin_bit_cast(float, w[j] << 16);
        v[2 * j + 1] = __builtin_bit_cast(float, w[j] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4 pack8(const float (&v)[8]) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 p = {v[2 * j], v[2 * j + 1]};
        o[j] = __builtin_bit_cast(unsigned int, __builtin_convertvector(p, bf16x2));
    }
    return o;
}

// source coordinate of output index o: i0, i1 and the weight of i1 (torch's area_pixel_compute_source_index, align_corners = False)
__device__ __forceinline__ void src_index(int o, float inv_scale, int in_size, int& i0, int& i1, float& lam) {
    float s = ((float)o + 0.5f) * inv_scale - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    lam = s - (float)i0;
}

__global__ void __launch_bounds__(256) upsample_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ skip, bf16* __restrict__ out,
                                                           int B, int H, int W, int C, int scale) {
    const int OH = H * scale, OW = W * scale, cq = C >> 3;
    const float inv = 1.f / (float)scale;
    const int64_t total = (int64_t)B * OH * OW * cq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % cq);
        int64_t p = i / cq;
        const int ox = (int)(p % OW);
        p /= OW;
        const int oy = (int)(p % OH), b = (int)(p / OH);
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(oy, inv, H, y0, y1, ly);
        src_index(ox, inv, W, x0, x1, lx);
        const bf16* xb = x + (int64_t)b * H * W * C + 8 * q;
        float a[8], bq[8], c[8], d[8], o[8];
        unpack8(*reinterpret_cast<const u32x4*>(xb + ((int64_t)y0 * W + x0) * C), a);
        unpack8(*reinterpret_cast<const u32x4*>(xb + ((int64_t)y0 * W + x1) * C), bq);
        unpack8(*reinterpret_cast<const u32x4*>(xb + ((int64_t)y1 * W + x0) * C), c);
        unpack8(*reinterpret_cast<const u32x4*>(xb + ((int64_t)y1 * W + x1) * C), d);
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = w00 * a[j] + w01 * bq[j] + w10 * c[j] + w11 * d[j];
        const int64_t off = (((int64_t)b * OH + oy) * OW + ox) * C + 8 * q;
        if (skip) {
            float sv[8];
            unpack8(*reinterpret_cast<const u32x4*>(skip + off), sv);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += sv[j];
        }
        *reinterpret_cast<u32x4*>(out + off) = pack8(o);
    }
}

__global__ void __launch_bounds__(256) upsample_bwd_kernel(const bf16* __restrict__ dout, bf16* __restrict__ dx, int B, int H, int W, int C,
                                                           int scale) {
    const int OH = H * scale, OW = W * scale, cq = C >> 3;
    const float inv = 1.f / (float)scale;
    const int64_t total = (int64_t)B * H * W * cq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q = (int)(i % cq);
        int64_t p = i / cq;
        const int ix = (int)(p % W);
        p /= W;
        const int iy = (int)(p % H), b = (int)(p / H);
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        const int oy_lo = max(0, scale * (iy - 1)), oy_hi = min(OH - 1, scale * (iy + 2) - 1);
        const int ox_lo = max(0, scale * (ix - 1)), ox_hi = min(OW - 1, scale * (ix + 2) - 1);
        const bf16* db = dout + (int64_t)b * OH * OW * C + 8 * q;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1;
            float ly;
            src_index(oy, inv, H, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1;
                float lx;
                src_index(ox, inv, W, x0, x1, lx);
                const float w = wy * ((x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f));
                if (w == 0.f) continue;
                float g[8];
                unpack8(*reinterpret_cast<const u32x4*>(db + ((int64_t)oy * OW + ox) * C), g);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(w, g[j], acc[j]);
            }
        }
        *reinterpret_cast<u32x4*>(dx + (((int64_t)b * H + iy) * W + ix) * C + 8 * q) = pack8(acc);
    }
}

int grid_for(int64_t total) {
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    return (int)(blocks < 1 ? 1 : blocks);
}

}  // namespace
}  // namespace p4c

using namespace p4c;

extern "C" int p4c_upsample_bilinear_fwd(const void* x, const void* skip, void* out, int B, int H, int W, int C, int scale, p4c_stream_t stream) {
    P4C_CHECK_ARG(x && out, "p4c_upsample_bilinear_fwd: NULL pointer");
    P4C_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && scale >= 1 && scale <= 8, "p4c_upsample_bilinear_fwd: B=%d H=%d W=%d C=%d scale=%d "
                  "(C a multiple of 8, scale 1..8)", B, H, W, C, scale);
    const int64_t total = (int64_t)B * H * scale * W * scale * (C / 8);
    hipLaunchKernelGGL(upsample_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const bf16*)x, (const bf16*)skip, (bf16*)out, B, H,
                       W, C, scale);
    P4C_CHECK_LAUNCH("upsample_fwd");
    return P4C_OK;
}

extern "C" int p4c_upsample_bilinear_bwd(const void* dout, void* dx, int B, int H, int W, int C, int scale, p4c_stream_t stream) {
    P4C_CHECK_ARG(dout && dx, "p4c_upsample_bilinear_bwd: NULL pointer");
    P4C_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && scale >= 1 && scale <= 8, "p4c_upsample_bilinear_bwd: bad sizes");
    const int64_t total = (int64_t)B * H * W * (C / 8);
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const bf16*)dout, (bf16*)dx, B, H, W, C, scale);
    P4C_CHECK_LAUNCH("upsample_bwd");
    return P4C_OK;
}
