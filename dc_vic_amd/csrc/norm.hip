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

__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// One workgroup per (n, group).  The group's cg*HW floats are contiguous in NCHW: 2 reads + 1 write of the group.
__global__ __launch_bounds__(512) void groupnorm_kernel(const float* __restrict__ x, long long x_bs, float* __restrict__ y,
                                                        long long y_bs, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int C, int HW, int groups, float eps,
                                                        int act, const float* __restrict__ part, int n_pt) {
    __shared__ double redd[16];
    const int n = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cg = C / groups;
    const long long len = (long long)cg * HW;
    const float* xp = x + (long long)n * x_bs + (long long)g * cg * HW;
    float* yp = y + (long long)n * y_bs + (long long)g * cg * HW;
    const bool vec = ((len & 3) == 0) && ((reinterpret_cast<uintptr_t>(xp) & 15) == 0) && ((reinterpret_cast<uintptr_t>(yp) & 15) == 0) && ((HW & 3) == 0);
    // one statistics pass: sum and sum of squares accumulated in fp64 (E[x^2] - mean^2 is then exact to fp32
    // accuracy, no second read of the group), reduced in a fixed order
    double s = 0.0, q = 0.0;
    if (part) {
        // statistics handed over by the producing convolution (csrc/wino44.hip epilogue): per (channel, pixel tile) partial sums of the
        // very values stored in x; the cg * n_pt pairs of this group are contiguous.  Added in fp64, fixed order: no read of x here.
        const float2* pp = reinterpret_cast<const float2*>(part) + ((long long)n * C + (long long)g * cg) * n_pt;
        for (int i = threadIdx.x; i < cg * n_pt; i += blockDim.x) { const float2 v = pp[i]; s += (double)v.x; q += (double)v.y; }
    } else if (vec) {
        // four 16-byte loads in flight per thread (one per iteration left half of the HBM bandwidth unused: ~32 KiB in flight per CU);
        // the additions keep the one-accumulator order i = tid, tid + B, tid + 2B, ... of the plain loop, so the sums are the same bits
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        const long long n4 = len / 4, B = blockDim.x;
        auto acc = [&](const float4 v) {
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        };
        long long i = threadIdx.x;
        for (; i + 3 * B < n4; i += 4 * B) {
            const float4 v0 = x4[i], v1 = x4[i + B], v2 = x4[i + 2 * B], v3 = x4[i + 3 * B];
            acc(v0); acc(v1); acc(v2); acc(v3);
        }
        for (; i < n4; i += B) acc(x4[i]);
    } else {
        for (long long i = threadIdx.x; i < len; i += blockDim.x) { const double v = xp[i]; s += v; q += v * v; }
    }
    const double S = block_sum_d(s, redd), Q = block_sum_d(q, redd);
    const double mean_d = S / (double)len;
    const float mean = (float)mean_d;
    const float var = (float)fmax(Q / (double)len - mean_d * mean_d, 0.0);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        float4* y4 = reinterpret_cast<float4*>(yp);
        const int hw4 = HW / 4;
        const long long n4 = len / 4;
        const int B = blockDim.x;
        // the channel of element i advances incrementally (no 64-bit division per element); four loads in flight per thread
        int c = (int)threadIdx.x / hw4, j = (int)threadIdx.x - c * hw4;
        const int sc = B / hw4, sj = B - sc * hw4;
        auto put = [&](long long i, float4 v, int cc) {
            const float ga = gamma[g * cg + cc], be = beta[g * cg + cc];
            v.x = dcvic_act((v.x - mean) * rstd * ga + be, act);
            v.y = dcvic_act((v.y - mean) * rstd * ga + be, act);
            v.z = dcvic_act((v.z - mean) * rstd * ga + be, act);
            v.w = dcvic_act((v.w - mean) * rstd * ga + be, act);
            y4[i] = v;
        };
        auto step = [&]() { c += sc; j += sj; if (j >= hw4) { j -= hw4; ++c; } };
        long long i = threadIdx.x;
        for (; i + 3LL * B < n4; i += 4LL * B) {
            const float4 v0 = x4[i], v1 = x4[i + B], v2 = x4[i + 2LL * B], v3 = x4[i + 3LL * B];
            const int c0 = c; step(); const int c1 = c; step(); const int c2 = c; step(); const int c3 = c; step();
            put(i, v0, c0); put(i + B, v1, c1); put(i + 2LL * B, v2, c2); put(i + 3LL * B, v3, c3);
        }
        for (; i < n4; i += B) { put(i, x4[i], c); step(); }
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
    groupnorm_kernel<<<N * groups, 512, 0, (hipStream_t)stream>>>(x, x_bs, y, y_bs, gamma, beta, C, HW, groups, eps, act, nullptr, 0);
    DCVIC_CHECK_LAUNCH("groupnorm");
    return DCVIC_OK;
}

extern "C" int dcvic_groupnorm_part_f32(const float* x, long long x_bs, float* y, long long y_bs, const float* gamma,
                                        const float* beta, int N, int C, int HW, int groups, float eps, int act,
                                        const float* part, int n_pt, void* stream) {
    DCVIC_CHECK_ARG(x && y && gamma && beta && part && n_pt > 0, "groupnorm_part: null pointer / no tiles");
    DCVIC_CHECK_ARG(N > 0 && C > 0 && HW > 0 && groups > 0 && C % groups == 0, "groupnorm_part: C=%d groups=%d", C, groups);
    DCVIC_CHECK_ARG(x_bs >= (long long)C * HW && y_bs >= (long long)C * HW, "groupnorm_part: batch stride too small");
    groupnorm_kernel<<<N * groups, 512, 0, (hipStream_t)stream>>>(x, x_bs, y, y_bs, gamma, beta, C, HW, groups, eps, act, part, n_pt);
    DCVIC_CHECK_LAUNCH("groupnorm_part");
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

// The same arithmetic (identical order: results are the same bits) with the pixel's C <= CMAX values held in registers: one read and
// one write of the map instead of three reads and one write, and CMAX independent loads in flight per thread (the loop form ran at
// 0.9 TB/s on the estimator's 128-channel maps).
template <int CMAX>
__global__ __launch_bounds__(64) void layernorm_c_reg_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             int C, int HW, float eps) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = blockIdx.y;
    if (p >= HW) return;
    const float* xp = x + (long long)n * C * HW + p;
    float* yp = y + (long long)n * C * HW + p;
    float v[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) v[c] = c < C ? xp[(long long)c * HW] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) s += v[c];
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) { const float a = v[c] - mean; q += a * a; }
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) yp[(long long)c * HW] = (v[c] - mean) * rstd * gamma[c] + beta[c];
}

extern "C" int dcvic_layernorm_c_f32(const float* x, float* y, const float* gamma, const float* beta, int N, int C, int HW,
                                     float eps, void* stream) {
    DCVIC_CHECK_ARG(x && y && gamma && beta && N > 0 && C > 0 && HW > 0, "layernorm_c: bad argument");
    if (C <= 128 && N <= 65535) {
        dim3 grid(dcvic_cdiv(HW, 64), N);
        layernorm_c_reg_kernel<128><<<grid, 64, 0, (hipStream_t)stream>>>(x, y, gamma, beta, C, HW, eps);
    } else {
        dim3 grid(dcvic_cdiv(HW, 256), N);
        layernorm_c_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, y, gamma, beta, C, HW, eps);
    }
    DCVIC_CHECK_LAUNCH("layernorm_c");
    return DCVIC_OK;
}

// softmax over C of [N][C][P], in place.  A workgroup owns 64 columns; its 4 waves split the rows (c = wave mod 4),
// keep an online (max, sum) pair per lane, merge the four pairs through LDS, then write exp(x - M) / S:
// 2 reads + 1 write of the map (coalesced 256-B rows) and N * P/64 workgroups.
__global__ __launch_bounds__(256) void softmax_c_kernel(float* __restrict__ x, int C, int P) {
    __shared__ float sm[4][64], ss[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + tx;
    const int n = blockIdx.y;
    const bool live = p < P;
    float* xp = x + (long long)n * C * P + (live ? p : 0);
    float m = -INFINITY, sum = 0.f;
    if (live) {
        for (int c = ty; c < C; c += 4) {
            const float v = xp[(long long)c * P];
            if (v > m) { sum = sum * expf(m - v) + 1.f; m = v; }
            else sum += expf(v - m);
        }
    }
    sm[ty][tx] = m; ss[ty][tx] = sum;
    __syncthreads();
    float M = fmaxf(fmaxf(sm[0][tx], sm[1][tx]), fmaxf(sm[2][tx], sm[3][tx]));
    float S = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) S += (sm[g][tx] == -INFINITY) ? 0.f : ss[g][tx] * expf(sm[g][tx] - M);
    if (live) {
        const float inv = 1.0f / S;
        for (int c = ty; c < C; c += 4) xp[(long long)c * P] = expf(xp[(long long)c * P] - M) * inv;
    }
}

extern "C" int dcvic_softmax_c_f32(float* x, int N, int C, int P, void* stream) {
    DCVIC_CHECK_ARG(x && N > 0 && C > 0 && P > 0, "softmax_c: bad argument");
    DCVIC_CHECK_ARG(N <= 65535, "softmax_c: batch too large");
    dim3 grid(dcvic_cdiv(P, 64), N);
    softmax_c_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, C, P);
    DCVIC_CHECK_LAUNCH("softmax_c");
    return DCVIC_OK;
}
