// Shared host/device helpers for the py4cast MI355X (gfx950) kernels.
// Every exported entry point is declared in include/py4cast_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/py4cast_hip.h"

namespace p4c {

// ---------------------------------------------------------------- errors
// Thread-local message; entry points return 0 or a negative code and never throw/abort.
char* err_buf();
int fail(int code, const char* fmt, ...);

#define P4C_CHECK_ARG(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return ::p4c::fail(P4C_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define P4C_CHECK_LAUNCH(name)                                                               \
    do {                                                                                     \
        hipError_t e__ = hipGetLastError();                                                  \
        if (e__ != hipSuccess)                                                               \
            return ::p4c::fail(P4C_ERR_LAUNCH, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

#define P4C_CHECK_HIP(expr)                                                                       \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess)                                                                    \
            return ::p4c::fail(P4C_ERR_RUNTIME, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

#define P4C_TRY(expr)                    \
    do {                                 \
        int rc__ = (expr);               \
        if (rc__ != P4C_OK) return rc__; \
    } while (0)

// Zero-fill of small device buffers as a KERNEL, never hipMemsetAsync: on this stack (ROCm 7.2, gfx950) a memset node captured into
// a HIP graph writes zeros in the first replay only -- from the second replay on it writes another byte value (0x01 observed;
// tools/diagnostics/memset_node_probe.py).  That is also what breaks torch's multi-block reductions in replays (they zero their
// semaphores with a memset), so nothing of ours that may run inside a captured step uses one.
#if defined(__HIPCC__)
static __global__ void p4c_zero_words_kernel(unsigned int* p, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = 0u;
}
inline hipError_t zero_words_async(void* p, size_t bytes, hipStream_t st) {   // bytes: a multiple of 4
    const long long n = (long long)(bytes / 4);
    if (n <= 0) return hipSuccess;
    const int blocks = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(p4c_zero_words_kernel, dim3(blocks), dim3(256), 0, st, (unsigned int*)p, n);
    return hipGetLastError();
}
#endif

static inline hipStream_t as_stream(p4c_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// A/B switches (P4C_* environment variables that select an older kernel, a geometry, a timing experiment) exist in DIAGNOSTIC builds
// only: `make diag` (-DP4C_DIAG_BUILD) -> libpy4cast_hip_diag.so, loaded by tools/diagnostics and by the A/B parity tests.  In the
// product library every switch compiles to its default -- one path per shape, and a deployment cannot differ silently from the
// measured configuration through its environment.
#include <stdlib.h>
#ifdef P4C_DIAG_BUILD
static inline const char* diag_env(const char* name) { return getenv(name); }
#else
static inline const char* diag_env(const char*) { return nullptr; }
#endif

// Timing diagnostics only (results become wrong): P4C_DIAG bit mask, honoured after the first 400 calls of each site so that
// buffers hold plausible values.  1: skip forward norm_finalize, 2: skip norm_bwd_finalize, 4: skip norm_bwd_reduce too,
// 8: skip wgrad_reduce, 16: skip norm_bwd_apply.
int diag_skip(int bit);

// Number of CUs of the current device (cached).  Used to size persistent grids.
int num_cus();

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size): lock-guarded table, so concurrent first
// launches from several host threads (Lightning's fit thread + autograd's backward thread) and several devices are safe.
int ensure_dyn_smem(const void* kernel, int bytes);

// Optional per-launch timing of selected kernels with HIP events recorded on the launch stream
// (bench.py roofline leg; see p4c_prof_enable in include/py4cast_hip.h).  No-ops unless enabled.
void prof_begin(int tag, int64_t units, hipStream_t stream);
void prof_end(int tag, hipStream_t stream);
// launches issued by the backward plan are recorded under the *_BWD tag (they overlap the side-stream weight gradients)
void prof_set_backward(bool on);

// ---------------------------------------------------------------- device helpers
typedef __hip_bfloat16 bf16;

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16>(bf16 v) { return __bfloat162float(v); }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16 from_f32<bf16>(float v) { return __float2bfloat16(v); }

// 64-lane wave all-reduce (sum) through DPP/shuffles.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over groups of `width` consecutive lanes (width = power of two <= 64).
__device__ __forceinline__ float seg_sum(float v, int width) {
    for (int off = width >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over lanes that share (lane % width): i.e. across the 64/width segments.
__device__ __forceinline__ float cross_seg_sum(float v, int width) {
    for (int off = 32; off >= width; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float nan_to_zero(float v) { return (v != v) ? 0.0f : v; }

// 4 consecutive elements as fp32, from/to fp32 (16 B) or bf16 (8 B) storage
typedef float p4c_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 p4c_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ p4c_f32x4 load4f(const float* p) { return *reinterpret_cast<const p4c_f32x4*>(p); }
__device__ __forceinline__ p4c_f32x4 load4f(const bf16* p) {
    const p4c_bf16x4 v = *reinterpret_cast<const p4c_bf16x4*>(p);
    return p4c_f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void store4f(float* p, p4c_f32x4 v) { *reinterpret_cast<p4c_f32x4*>(p) = v; }
__device__ __forceinline__ void store4f(bf16* p, p4c_f32x4 v) {
    p4c_bf16x4 o;
    o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<p4c_bf16x4*>(p) = o;
}

}  // namespace p4c
