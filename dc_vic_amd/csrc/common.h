// Shared host/device helpers for libdcvic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "dcvic.h"

void dcvic_set_error(const char* fmt, ...);

#define DCVIC_CHECK_ARG(cond, ...)            \
    do {                                      \
        if (!(cond)) {                        \
            dcvic_set_error(__VA_ARGS__);     \
            return DCVIC_EINVAL;              \
        }                                     \
    } while (0)

#define DCVIC_CHECK_LAUNCH(what)                                                   \
    do {                                                                           \
        hipError_t e__ = hipGetLastError();                                        \
        if (e__ != hipSuccess) {                                                   \
            dcvic_set_error("%s: %s", what, hipGetErrorString(e__));               \
            return DCVIC_ELAUNCH;                                                  \
        }                                                                          \
    } while (0)

__device__ __forceinline__ float dcvic_act(float v, int act) {
    switch (act) {
        case DCVIC_ACT_RELU: return fmaxf(v, 0.f);
        case DCVIC_ACT_LRELU02: return v > 0.f ? v : 0.2f * v;
        case DCVIC_ACT_SWISH: return v / (1.f + expf(-v));
        case DCVIC_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        case DCVIC_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case DCVIC_ACT_HALF_TANH: return 0.5f * tanhf(v);
        default: return v;
    }
}

// Function attributes (dynamic-LDS limit) are per device: returns true the first time the calling site runs
// on the current HIP device (bit per ordinal in a site-local mask; devices >= 32 simply re-apply every call).
static inline bool dcvic_first_use_on_device(std::atomic<unsigned>& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return true;
    const unsigned bit = 1u << dev;
    return (mask.fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
}
// Compute units of the current HIP device (cached per ordinal).
int dcvic_num_cu();

static inline int dcvic_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
