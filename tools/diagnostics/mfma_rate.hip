// micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 in several instruction-stream shapes (one wave per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(const bf16x8* in, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 16; i += 256) reinterpret_cast<bf16x8*>(lds)[i] = in[i & 255];
    __syncthreads();
    bf16x8 a = in[lane], b = in[64 + lane];
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const char* p = lds + lane * 16;
    unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {          // pure MFMA, 8 accumulators
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
        }
    } else if (MODE == 1) {   // 1 ds_read_b128 per MFMA, read two steps ahead, interleaved one by one
        bf16x8 q[10];
#pragma unroll
        for (int t = 0; t < 2; ++t) q[t] = *reinterpret_cast<const bf16x8*>(p + t * 1024);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                q[t + 2] = *reinterpret_cast<const bf16x8*>(p + ((it * 8 + t + 2) & 63) * 1024);
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            q[0] = q[8]; q[1] = q[9];
        }
    } else if (MODE == 2) {   // bursts: 4 reads then 4 MFMAs (the ring kernel's shape)
        bf16x8 q[3][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) q[s][t] = *reinterpret_cast<const bf16x8*>(p + (s * 4 + t) * 1024);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 6; ++s) {
#pragma unroll
                for (int t = 0; t < 4; ++t) q[(s + 2) % 3][t] = *reinterpret_cast<const bf16x8*>(p + ((it * 24 + s * 4 + t + 8) & 63) * 1024);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q[s % 3][t], acc[t & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (MODE == 4) {   // fp32 matrix cores: v_mfma_f32_32x32x2_f32, 8 accumulators
        const float af = __builtin_bit_cast(float, ((const unsigned*)&a)[0]), bf_ = __builtin_bit_cast(float, ((const unsigned*)&b)[1]);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf_, acc[t], 0, 0, 0);
        }
    } else if (MODE == 3) {   // pure MFMA, 2 accumulators alternating
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t & 1], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int mfma_per_iter, bool zeros) {
    const int G = 256, iters = 2000;
    bf16x8* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 4096 * 16); hipMalloc(&out, G * 256 * 4); hipMalloc(&cyc, G * 8);
    std::vector<unsigned short> h(4096 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = zeros ? 0 : (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(G); hipMemcpy(c.data(), cyc, G * 8, hipMemcpyDeviceToHost);
    double n = (double)iters * mfma_per_iter;
    // s_memtime-based counter ticks at a constant 100 MHz on this part; wall time gives ns per MFMA
    printf("%-44s %s: %.1f ns per MFMA (wall), counter ticks/MFMA %.2f\n", name, zeros ? "zeros " : "random", ms / 10 * 1e6 / n, c[0] / n);
    hipFree(in); hipFree(out); hipFree(cyc);
}
int main() {
    for (int z = 0; z < 2; ++z) {
        run<0>("pure MFMA, 8 accumulators", 8, z);
        run<3>("pure MFMA, 2 accumulators", 8, z);
        run<1>("1 ds_read_b128 per MFMA, interleaved", 8, z);
        run<2>("4 reads then 4 MFMAs (bursts)", 24, z);
        run<4>("fp32 v_mfma_f32_32x32x2_f32, 8 accumulators", 8, z);
    }
    return 0;
}
