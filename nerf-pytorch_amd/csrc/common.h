// Shared host/device helpers for libnerf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include <mutex>

#include "../../include/nerf_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// thread-local text of the last error, read through nerf_last_error()
extern thread_local char g_nerf_err[256];

static inline int nerf_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_nerf_err, sizeof(g_nerf_err), fmt, ap);
    va_end(ap);
    return code;
}

#define NERF_REQUIRE(cond, ...)                              \
    do {                                                     \
        if (!(cond)) return nerf_fail(NERF_E_ARG, __VA_ARGS__); \
    } while (0)

#define NERF_REQUIRE_PTR(p)                                                   \
    do {                                                                      \
        if ((p) == nullptr) return nerf_fail(NERF_E_NULL, "%s is NULL", #p);  \
    } while (0)

static inline int nerf_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nerf_fail(NERF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return NERF_OK;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: raise it when `kernel` first
// runs on a device, or when a launch needs more than every earlier one asked for (a kernel whose LDS size
// depends on its arguments), and skip the runtime call otherwise.  The table of (kernel, device) -> largest size
// set so far is shared by all host threads and guarded by a mutex.
template <typename K>
static inline int nerf_set_lds(K kernel, int bytes) {
    constexpr int kSlots = 64, kDevs = 16;
    static std::mutex mu;
    static const void* known[kSlots] = {};
    static int set_bytes[kSlots][kDevs] = {};
    const void* fn = reinterpret_cast<const void*>(kernel);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kDevs) dev = -1;
    std::lock_guard<std::mutex> lock(mu);
    int slot = -1;
    for (int i = 0; i < kSlots; ++i) {
        if (known[i] == fn) { slot = i; break; }
        if (known[i] == nullptr) { known[i] = fn; slot = i; break; }
    }
    if (slot >= 0 && dev >= 0 && set_bytes[slot][dev] >= bytes) return NERF_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return nerf_fail(NERF_E_LAUNCH, "hipFuncSetAttribute(%d B LDS): %s", bytes, hipGetErrorString(e));
    if (slot >= 0 && dev >= 0) set_bytes[slot][dev] = bytes;
    return NERF_OK;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Individually rounded fp32 operations.  HIP's __fadd_rn/__fmul_rn/... are plain operators
// defined in a header compiled with -ffp-contract=fast, so a mul feeding an add still fuses
// into an FMA; these helpers carry no contraction permission and survive inlining.
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float div_rn(float a, float b) {
#pragma clang fp contract(off)
    return a / b;
}
