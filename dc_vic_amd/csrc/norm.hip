// norm.hip -- GroupNorm(+act), channel LayerNorm, channel softmax.  HBM-bound kernels: float4 /
// coalesced plane reads, fixed-order block reductions (deterministic, batch-invariant).
#include "common.h"

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum, result broadcast to every thread.  red: >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// One workgroup per (n, group).  The group's cg*HW floats are contiguous in NCHW.
// Two-pass statistics (mean, then centred variance) like torch's CPU kernel -- no E[x^2]-m^2.
__global__ __launch_bounds__(512) void groupnorm_kernel(const float* __restrict__ x, long long x_bs, float* __restrict__ y,
                                                        long long y_bs, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int C, int HW, int groups, float eps,
                                                        int act) {
    __shared__ float red[16];
    const int n = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cg = C / groups;
    const long long len = (long long)cg * HW;
    const float* xp = x + (long long)n * x_bs + (long long)g * cg * HW;
    float* yp = y + (long long)n * y_bs + (long long)g * cg * HW;
    const bool vec = ((len & 3) == 0) && ((reinterpret_cast<uintptr_t>(xp) & 15) == 0) && ((reinterpret_cast<uintptr_t>(yp) & 15) == 0) && ((HW & 3) == 0);
    float s = 0.f;
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        for (long long i = threadIdx.x; i < len / 4; i += blockDim.x) { float4 v = x4[i]; s += (v.x + v.y) + (v.z + v.w); }
    } else {
        for (long long i = threadIdx.x; i < len; i += blockDim.x) s += xp[i];
    }
    const float mean = block_sum(s, red) / (float)len;
    float q = 0.f;
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        for (long long i = threadIdx.x; i < len / 4; i += blockDim.x) {
            float4 v = x4[i];
            const float a = v.x - mean, b = v.y - mean, c = v.z - mean, d = v.w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (long long i = threadIdx.x; i < len; i += blockDim.x) { const float a = xp[i] - mean; q += a * a; }
    }
    const float var = block_sum(q, red) / (float)len;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        float4* y4 = reinterpret_cast<float4*>(yp);
        const int hw4 = HW / 4;
        for (long long i = threadIdx.x; i < len / 4; i += blockDim.x) {
            const int c = g * cg + (int)(i / hw4);
            const float ga = gamma[c], be = beta[c];
            float4 v = x4[i];
            v.x = dcvic_act((v.x - mean) * rstd * ga + be, act);
            v.y = dcvic_act((v.y - mean) * rstd * ga + be, act);
            v.z = dcvic_act((v.z - mean) * rstd * ga + be, act);
            v.w = dcvic_act((v.w - mean) * rstd * ga + be, act);
            y4[i] = v;
        }
    } else {
        for (long long i = threadIdx.x; i < len; i += blockDim.x) {
            const int c = g * cg + (int)(i / HW);
            yp[i] = dcvic_act((xp[i] - mean) * rstd * gamma[c] + beta[c], act);
        }
    }
}

extern "C" int dcvic_groupnorm_f32(const float* x, long long x_bs, float* y, long long y_bs, const float* gamma,
                                   const float* beta, int N, int C, int HW, int groups, float eps, int act, void* stream) {
    DCVIC_CHECK_ARG(x && y && gamma && beta, "groupnorm: null pointer");
    DCVIC_CHECK_ARG(N > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "groupnorm: C=%d groups=%d", C, groups);
    DCVIC_CHECK_ARG(x_bs >= (long long)C * HW && y_bs >= (long long)C * HW, "groupnorm: batch stride too small");
    groupnorm_kernel<<<N * groups, 512, 0, (hipStream_t)stream>>>(x, x_bs, y, y_bs, gamma, beta, C, HW, groups, eps, act);
    DCVIC_CHECK_LAUNCH("groupnorm");
    return DCVIC_OK;
}

// LayerNorm over C for every pixel of an NCHW map: thread per pixel, coalesced across the wave.
__global__ __launch_bounds__(256) void layernorm_c_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          int C, int HW, float eps) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= HW) return;
    const float* xp = x + (long long)n * C * HW + p;
    float* yp = y + (long long)n * C * HW + p;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += xp[(long long)c * HW];
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = 0; c < C; ++c) { const float a = xp[(long long)c * HW] - mean; q += a * a; }
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    for (int c = 0; c < C; ++c) yp[(long long)c * HW] = (xp[(long long)c * HW] - mean) * rstd * gamma[c] + beta[c];
}

extern "C" int dcvic_layernorm_c_f32(const float* x, float* y, const float* gamma, const float* beta, int N, int C, int HW,
                                     float eps, void* stream) {
    DCVIC_CHECK_ARG(x && y && gamma && beta && N > 0 && C > 0 && HW > 0, "layernorm_c: bad argument");
    dim3 grid(dcvic_cdiv(HW, 256), N);
    layernorm_c_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, gamma, beta, C, HW, eps);
    DCVIC_CHECK_LAUNCH("layernorm_c");
    return DCVIC_OK;
}

// softmax over C of [N][C][P], in place; thread per column p.
__global__ __launch_bounds__(256) void softmax_c_kernel(float* __restrict__ x, int C, int P) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= P) return;
    float* xp = x + (long long)n * C * P + p;
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, xp[(long long)c * P]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) { const float e = expf(xp[(long long)c * P] - m); xp[(long long)c * P] = e; s += e; }
    for (int c = 0; c < C; ++c) xp[(long long)c * P] = xp[(long long)c * P] / s;
}

extern "C" int dcvic_softmax_c_f32(float* x, int N, int C, int P, void* stream) {
    DCVIC_CHECK_ARG(x && N > 0 && C > 0 && P > 0, "softmax_c: bad argument");
    dim3 grid(dcvic_cdiv(P, 256), N);
    softmax_c_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, C, P);
    DCVIC_CHECK_LAUNCH("softmax_c");
    return DCVIC_OK;
}
