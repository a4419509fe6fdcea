// Thread-local error string + library-wide queries.
#include "common.h"

static thread_local char g_err[512] = "";

void dcvic_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* dcvic_last_error(void) { return g_err; }
extern "C" int dcvic_version(void) { return 100; }

extern "C" int dcvic_device_info(int* n_cu, int* lds_bytes) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        dcvic_set_error("device_info: no HIP device");
        return DCVIC_ELAUNCH;
    }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    return DCVIC_OK;
}

int dcvic_num_cu() {
    static std::atomic<int> cache[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < 32) {
        const int c = cache[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
    if (dev >= 0 && dev < 32) cache[dev].store(cu, std::memory_order_relaxed);
    return cu;
}
