// Shared helpers of the HIP translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/visp_hip_kernels.h"

void vx_set_error(const char* fmt, ...);

#define VX_CHECK(expr)                                                                     \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            vx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 0;                                                                      \
        }                                                                                  \
    } while (0)

#define VX_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            vx_set_error(__VA_ARGS__);        \
            return 0;                         \
        }                                     \
    } while (0)

#define VX_LAUNCH_CHECK() VX_CHECK(hipGetLastError())

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute: set it once per (kernel, device), not once per process
hipError_t vx_ensure_dynamic_lds(const void* kernel, int bytes);

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
