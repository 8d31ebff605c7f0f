// micro-benchmark: wall time per bf16 MFMA on RANDOM operands (the chip is power-limited there) as a function of how the
// instruction stream uses accumulators and operands -- the shapes the 3x3 convolution kernels can choose between.
//   hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// MODE: 0 one accumulator chain | 1 two alternating | 2 three round-robin (P Q S per operand, the row kernel) |
//       3 three in runs of 4 | 4 three in runs of 12 | 5 eight round-robin |
//       6 three round-robin + one ds_read_b128 per 3 MFMAs (B from LDS) | 7 runs of 4 + the same reads |
//       10 16x16x32: 6 accumulators (3 rows x 2 pixel blocks) round-robin, 72 per "row" | 11 16x16x32 in runs of 4 per accumulator
template <int MODE>
__global__ void __launch_bounds__(256, 1) k(const bf16x8* in, float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 16; i += 256) reinterpret_cast<bf16x8*>(lds)[i] = in[i & 1023];
    __syncthreads();
    bf16x8 A[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) A[i] = in[lane + 64 * (i % 16)];
    bf16x8 Bq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) Bq[i] = in[lane + 64 * (3 + i)];
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    f32x4 c4[6];
    for (int t = 0; t < 6; ++t) for (int i = 0; i < 4; ++i) c4[t][i] = 0.f;
    const char* p = lds + lane * 16;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 36; ++i) { acc[0] = MFMA32(A[i], Bq[i & 3], acc[0]); FENCE(); }
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 36; ++i) { acc[i & 1] = MFMA32(A[i], Bq[i & 3], acc[i & 1]); FENCE(); }
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 36; ++i) { acc[i % 3] = MFMA32(A[i], Bq[(i / 3) & 3], acc[i % 3]); FENCE(); }
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 36; ++i) { acc[(i / 4) % 3] = MFMA32(A[i], Bq[i & 3], acc[(i / 4) % 3]); FENCE(); }
        } else if (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 36; ++i) { acc[i / 12] = MFMA32(A[i], Bq[i & 3], acc[i / 12]); FENCE(); }
        } else if (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 36; ++i) { acc[i & 7] = MFMA32(A[i], Bq[i & 3], acc[i & 7]); FENCE(); }
        } else if (MODE == 6) {
            bf16x8 q[4];
#pragma unroll
            for (int t = 0; t < 3; ++t) q[t] = *reinterpret_cast<const bf16x8*>(p + ((it + t) & 63) * 1024);
#pragma unroll
            for (int o = 0; o < 12; ++o) {
                q[(o + 3) & 3] = *reinterpret_cast<const bf16x8*>(p + ((it + o + 3) & 63) * 1024);
                FENCE();
#pragma unroll
                for (int c = 0; c < 3; ++c) { acc[c] = MFMA32(A[3 * o + c], q[o & 3], acc[c]); FENCE(); }
            }
        } else if (MODE == 7) {
            bf16x8 q[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] = *reinterpret_cast<const bf16x8*>(p + ((it + t) & 63) * 1024);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int o = 0; o < 4; ++o) {
                        if (c == 0 || (c == 1 && o < 0)) {}
                        if (c < 2 && o < 2) { q[((g + 1) & 1) * 4 + c * 2 + o] = *reinterpret_cast<const bf16x8*>(p + ((it + g * 4 + c * 2 + o + 4) & 63) * 1024); FENCE(); }
                        acc[c] = MFMA32(A[g * 12 + c * 4 + o], q[(g & 1) * 4 + o], acc[c]);
                        FENCE();
                    }
            }
        } else if (MODE == 10) {
#pragma unroll
            for (int i = 0; i < 72; ++i) { c4[i % 6] = MFMA16(A[i % 36], Bq[(i / 6) & 3], c4[i % 6]); FENCE(); }
        } else if (MODE == 11) {
#pragma unroll
            for (int i = 0; i < 72; ++i) { c4[(i / 4) % 6] = MFMA16(A[i % 36], Bq[i & 3], c4[(i / 4) % 6]); FENCE(); }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    for (int t = 0; t < 6; ++t) for (int i = 0; i < 4; ++i) s += c4[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int mfma_per_iter, double flop_scale) {
    const int G = 256, iters = 1500;
    bf16x8* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 4096 * 16); hipMalloc(&out, G * 256 * 4); hipMalloc(&cyc, G * 8);
    std::vector<unsigned short> h(4096 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e0);
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k<MODE>, dim3(G), dim3(256), 0, 0, in, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> c(G); hipMemcpy(c.data(), cyc, G * 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * mfma_per_iter;
    const double ns = ms / 20 * 1e6 / n;
    printf("%-64s %.2f ns per MFMA (= %.2f ns per 32x32x16 of work), 100 MHz ticks per MFMA %.3f\n", name, ns, ns * flop_scale, c[0] / n);
    hipFree(in); hipFree(out); hipFree(cyc);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("32x32x16, one accumulator chain", 36, 1);
        run<1>("32x32x16, two accumulators alternating", 36, 1);
        run<2>("32x32x16, three round-robin (row kernel)", 36, 1);
        run<3>("32x32x16, three in runs of 4", 36, 1);
        run<4>("32x32x16, three in runs of 12", 36, 1);
        run<5>("32x32x16, eight round-robin", 36, 1);
        run<6>("32x32x16, three round-robin + ds_read_b128 per 3 MFMAs", 36, 1);
        run<7>("32x32x16, runs of 4 + the same reads", 36, 1);
        run<10>("16x16x32, six accumulators round-robin", 72, 2);
        run<11>("16x16x32, six accumulators in runs of 4", 72, 2);
    }
    return 0;
}
