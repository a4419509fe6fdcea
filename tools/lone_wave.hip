// Diagnostic: what limits ONE wave per SIMD running a dependent fp32-MFMA chain fed from LDS?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int OFF>
__device__ __forceinline__ float ldsr(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// MODE 0: MFMA chain only.  1: + 2 asm ds_reads per MFMA, waits lgkmcnt(0) per 4 MFMAs (prefetch one group ahead).
// 2: same with plain C++ LDS loads.  3: mode 1 + 20 dependent SALU ops per 4 MFMAs.  4: mode 1 with NACC=2 (two tiles)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int step) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (float)((i * 2654435761u) >> 8) * 1e-9f;
    __syncthreads();
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    const unsigned base = (unsigned)(uintptr_t)(lds_ptr_t)lds + 4u * (threadIdx.x & 63);
    unsigned addr = base;
    float a[4], b[4], an[4], bn[4];
    for (int j = 0; j < 4; ++j) { a[j] = threadIdx.x * 1e-3f + j; b[j] = a[j] * 0.5f; an[j] = a[j]; bn[j] = b[j]; }
    int sal = step;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
        } else if (MODE == 2) {
            const float* p = lds + (threadIdx.x & 63) + ((it * step) & 4095);
#pragma unroll
            for (int j = 0; j < 4; ++j) { an[j] = p[j * 256]; bn[j] = p[j * 256 + 128]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = an[j]; b[j] = bn[j]; }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); FENCE();
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
            if (MODE == 4) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[1], acc2, 0, 0, 0);
            FENCE();
            an[0] = ldsr<0>(addr); an[1] = ldsr<1024>(addr); bn[0] = ldsr<512>(addr); bn[1] = ldsr<1536>(addr);
            FENCE();
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
            if (MODE == 4) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[0], acc2, 0, 0, 0);
            FENCE();
            an[2] = ldsr<2048>(addr); an[3] = ldsr<3072>(addr); bn[2] = ldsr<2560>(addr); bn[3] = ldsr<3584>(addr);
            FENCE();
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
            if (MODE == 4) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[3], acc2, 0, 0, 0);
            FENCE();
            if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 10; ++q) { sal = (sal == 7) ? step : sal + 1; sal += (sal & 1) ? step : 3; }
            }
            addr = base + 4u * (unsigned)((it * step + (MODE == 3 ? (sal & 1) : 0)) & 4095);
            FENCE();
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
            if (MODE == 4) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[2], acc2, 0, 0, 0);
            FENCE();
            // swap (register renaming by unroll would be better; these are 8 v_mov)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); FENCE();
#pragma unroll
            for (int j = 0; j < 4; ++j) { a[j] = an[j]; b[j] = bn[j]; }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// LDS round trip seen by a lone wave: ds_read_b32 -> s_waitcnt lgkmcnt(0), dependent chain (address from the value read)
__global__ __launch_bounds__(256) void klat(float* out, unsigned long long* cyc, int iters, int nreads) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
    __syncthreads();
    unsigned addr = (unsigned)(uintptr_t)(lds_ptr_t)lds + 4u * (threadIdx.x & 63);
    float acc = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float v0, v1, v2, v3;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(addr));
        if (nreads > 1) asm volatile("ds_read_b32 %0, %1 offset:512" : "=v"(v1) : "v"(addr)); else v1 = 0;
        if (nreads > 2) { asm volatile("ds_read_b32 %0, %1 offset:1024" : "=v"(v2) : "v"(addr)); asm volatile("ds_read_b32 %0, %1 offset:1536" : "=v"(v3) : "v"(addr)); } else { v2 = v3 = 0; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        acc += v0 + v1 + v2 + v3;
        addr += (unsigned)(int)(v0 * 4.f);       // dependent (always +0)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    float* out; hipMalloc(&out, sizeof(float) * 256 * 256);
    int iters = 20000;
    k<MODE><<<256, 256>>>(out, 10, 1);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, 256>>>(out, iters, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double n_mfma = (double)iters * 4 * (MODE == 4 ? 2 : 1);
    double cyc = ms * 1e-3 * 2.4e9 / (iters * 4.0);
    printf("%-58s %.3f ms  %.1f TFLOP/s  ~%.0f cycles per chain MFMA\n", name, ms, 256.0 * 4 * n_mfma * 4096.0 / ms * 1e-9, cyc);
}
void runlat(int nreads, int threads) {
    float* out; unsigned long long* cyc; hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8);
    klat<<<256, threads>>>(out, cyc, 2000, nreads); hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("LDS round trip, %d ds_read_b32 then lgkmcnt(0), %d waves/CU: %.0f shader cycles per iteration\n", nreads, threads / 64, (double)h / 2000.0);
}
int main() {
    runlat(1, 256); runlat(2, 256); runlat(4, 256); runlat(4, 1024);
    run<0>("chain only");
    run<1>("chain + asm ds_read prefetch, lgkmcnt(0) per 4");
    run<2>("chain + C++ LDS loads");
    run<3>("mode 1 + 20 dependent SALU per 4 MFMA");
    run<4>("mode 1, two accumulators (2x MFMA per step)");
    return 0;
}
