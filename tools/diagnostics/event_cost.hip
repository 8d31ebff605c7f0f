// What does ordering a second stream after a point of the main stream cost the MAIN stream?
// Main stream: N kernels that each stream `bytes` through HBM (read + write); after every `every`-th one a dependency is handed
// to a side stream that runs one small kernel per dependency.  Variants:
//   0  no side stream, no events                      (floor)
//   1  hipEventRecord (hipEventDisableTiming)         (what SideStream::order does)
//   2  hipEventRecord (DisableTiming | DisableSystemFence)
//   3  hipExtLaunchKernelGGL(stopEvent) on the producing kernel, no separate record
//   4  as 1, plus a hipStreamWaitEvent on the MAIN stream for the side kernel two dependencies back (buffer-reuse waits)
// build: hipcc -O3 --offload-arch=gfx950 event_cost.hip -o event_cost ; run: ./event_cost [N] [every] [MiB]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) stream_kernel(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = a[i];
        v.x += 1.f;
        b[i] = v;
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 240, every = argc > 2 ? atoi(argv[2]) : 4;
    const size_t mib = argc > 3 ? atoi(argv[3]) : 64;
    const size_t n = mib * 1024 * 1024 / 16;
    float4 *a, *b, *c, *d;
    CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMalloc(&c, n * 16 / 8)); CK(hipMalloc(&d, n * 16 / 8));
    CK(hipMemset(a, 0, n * 16)); CK(hipMemset(c, 0, n * 16 / 8));
    hipStream_t m, s;
    CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int ndep = N / every + 2;
    std::vector<hipEvent_t> plain(ndep), nofence(ndep), back(ndep);
    for (int i = 0; i < ndep; ++i) {
        CK(hipEventCreateWithFlags(&plain[i], hipEventDisableTiming));
        CK(hipEventCreateWithFlags(&nofence[i], hipEventDisableTiming | hipEventDisableSystemFence));
        CK(hipEventCreateWithFlags(&back[i], hipEventDisableTiming | hipEventDisableSystemFence));
    }
    for (int variant = 0; variant <= 4; ++variant)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            int dep = 0;
            for (int i = 0; i < N; ++i) {
                const bool hand = variant != 0 && (i + 1) % every == 0;
                if (hand && variant == 3) {
                    hipExtLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, m, nullptr, plain[dep], 0, a, b, n);
                } else {
                    hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, m, a, b, n);
                }
                if (hand) {
                    hipEvent_t ev = variant == 2 ? nofence[dep] : plain[dep];
                    if (variant != 3) CK(hipEventRecord(ev, m));
                    CK(hipStreamWaitEvent(s, ev, 0));
                    hipLaunchKernelGGL(stream_kernel, dim3(256), dim3(256), 0, s, c, d, n / 8);
                    if (variant == 4) {
                        CK(hipEventRecord(back[dep], s));
                        if (dep >= 2) CK(hipStreamWaitEvent(m, back[dep - 2], 0));
                    }
                    ++dep;
                }
            }
            CK(hipStreamSynchronize(m));
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            if (rep == 2)
                printf("variant %d: %8.1f us for %d kernels of %zu MiB (%.2f us each), %d dependencies\n", variant,
                       std::chrono::duration<double, std::micro>(t1 - t0).count(), N, mib,
                       std::chrono::duration<double, std::micro>(t1 - t0).count() / N, dep);
        }
    return 0;
}
