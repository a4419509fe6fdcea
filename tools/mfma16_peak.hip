// Diagnostic: sustained v_mfma_f32_16x16x4_f32 rate with the Winograd kernel's operand pattern (per position pair: three
// ds_read_b128, eight MFMAs on 32 independent accumulators), 8 waves per workgroup, one workgroup per CU -- and the same with
// v_mfma_f32_32x32x2_f32.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma16_peak.hip -o tools/mfma16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: registers only, 1: + the three ds_read_b128 per pair (random mantissas)
__global__ __launch_bounds__(512, 2) void k16(float* out, int iters) {
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) {
        unsigned h = (i + 16384u * blockIdx.x) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        lds[i] = __uint_as_float(0x3f800000u | (h >> 9)) * ((h & 1) ? 1.f : -1.f) - ((h & 1) ? 1.5f : -1.5f);
    }
    __syncthreads();
    f32x4 acc[16][2];
    for (int p = 0; p < 16; ++p) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) acc[p][b][r] = 0.f;
    const f32x4* base = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
    f32x4 A = base[0], B0 = base[64], B1 = base[128];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 1) { A = base[j * 256 + (it & 1) * 2048]; B0 = base[j * 256 + 64 + (it & 1) * 2048]; B1 = base[j * 256 + 128 + (it & 1) * 2048]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int pq = i >> 2, ks = (i >> 1) & 1, blk = i & 1;
                acc[2 * j + pq][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[pq * 2 + ks], (pq ? B1 : B0)[blk * 2 + ks], acc[2 * j + pq][blk], 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int p = 0; p < 16; ++p) for (int b = 0; b < 2; ++b) for (int r = 0; r < 4; ++r) s += acc[p][b][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(512, 2) void k32(float* out, int iters) {
    f32x16 acc[8];
    for (int p = 0; p < 8; ++p) for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = a * 0.5f + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < 8; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
    }
    float s = 0;
    for (int p = 0; p < 8; ++p) for (int r = 0; r < 16; ++r) s += acc[p][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F>
void timeit(const char* name, F launch, double flops) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(10); hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); launch(4000); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms  %.1f TFLOP/s\n", name, ms, flops * 4000 / ms * 1e-9);
    }
}
int main() {
    float* out; hipMalloc(&out, sizeof(float) * 256 * 512);
    const double f16 = 256.0 * 8 * 64 * 2048.0, f32 = 256.0 * 8 * 32 * 4096.0;   // per iteration
    timeit("16x16x4 registers only, 8 waves/CU", [&](int it) { k16<0><<<256, 512>>>(out, it); }, f16);
    timeit("16x16x4 + 3 ds_read_b128 per 8 MFMAs", [&](int it) { k16<1><<<256, 512>>>(out, it); }, f16);
    timeit("32x32x2 registers only, 8 waves/CU", [&](int it) { k32<<<256, 512>>>(out, it); }, f32);
    return 0;
}
