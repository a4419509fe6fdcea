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

// Function attributes (dynamic-LDS limit) are per device.  `if (DcvicAttrOnce once{mask}) { hipFuncSetAttribute(...); }` runs the
// body until it has COMPLETED once on the current HIP device: the device's bit in the site-local mask is set by the destructor,
// i.e. after the attribute calls -- a second thread arriving meanwhile applies the (idempotent) attribute itself instead of
// launching a > 64 KiB-LDS kernel before it is in effect.  Devices >= 32 re-apply on every call.
struct DcvicAttrOnce {
    std::atomic<unsigned>& mask;
    unsigned bit = 0;
    bool need = true;
    explicit DcvicAttrOnce(std::atomic<unsigned>& m) : mask(m) {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 32) {
            bit = 1u << dev;
            need = (mask.load(std::memory_order_acquire) & bit) == 0;
        }
    }
    ~DcvicAttrOnce() { if (need && bit) mask.fetch_or(bit, std::memory_order_release); }
    explicit operator bool() const { return need; }
    DcvicAttrOnce(const DcvicAttrOnce&) = delete;
    DcvicAttrOnce& operator=(const DcvicAttrOnce&) = delete;
};
// Compute units of the current HIP device (cached per ordinal).
int dcvic_num_cu();

static inline int dcvic_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
