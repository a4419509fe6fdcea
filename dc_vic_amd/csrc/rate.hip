// rate.hip -- rate estimation + symbolisation of the learned entropy model.
//
// dcvic_gaussian_rate_f32: GaussianConditional (CompressAI 1.2.4) as driven by
//   ste_gaussian_conditional.py:16-23 and minnen20_charm_context_model.py:96,148,164,199:
//     sym = rint(y - mu); y_hat = sym + mu; v = |y_hat - mu|; s = max(sigma, 0.11)
//     p = max(0.5 erfc(-(0.5 - v)/(s sqrt2)) - 0.5 erfc(-(-0.5 - v)/(s sqrt2)), 1e-9)
//     index = (n_scales - 1) - #{t in table[:-1] : s <= t}
//   decode mode (y == NULL): y_hat = float(sym_in) + mu.
// dcvic_eb_rate_f32: EntropyBottleneck eval forward (entropy_bottleneck.py:19-28 -> App-B).
// Both are 12-20 B/element streaming kernels.  One workgroup owns one image, walks it in a fixed
// order and reduces -log2 p with a fixed tree in fp64, so bits[n] is deterministic and independent
// of the batch size; bits_out[n] is ACCUMULATED (+=) so CHARM slices can add up.
#include "common.h"

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

__device__ __forceinline__ float std_cumulative(float x) { return 0.5f * erfcf(-0.70710678118654752440f * x); }

// one element: symbolise / de-quantise, cdf index, likelihood; returns log(p) (0 when not wanted)
__device__ __forceinline__ float gr_element(float yv, int sv, bool have_y, float m, float sg, const float* tab, int n_scales,
                                            bool want_idx, bool want_p, float& yh, int& sym_o, int& ix_o, float& p_o) {
    const float s = fmaxf(sg, 0.11f);
    const float sym = have_y ? rintf(__fsub_rn(yv, m)) : (float)sv;
    yh = __fadd_rn(sym, m);
    sym_o = (int)sym;
    if (want_idx) {
        // index = (n - 1) - #{t < n - 1 : s <= tab[t]} = first t with tab[t] >= s (the table ascends), clipped to n - 1:
        // a 6-step branchless binary search instead of 63 compares
        int lo = 0, len = n_scales - 1;
        while (len > 0) {
            const int half = len >> 1;
            const bool right = tab[lo + half] < s;
            lo = right ? lo + half + 1 : lo;
            len = right ? len - half - 1 : half;
        }
        ix_o = lo;
    }
    float lp = 0.f;
    if (want_p) {
        const float v = fabsf(__fsub_rn(yh, m));
        const float up = std_cumulative((0.5f - v) / s);
        const float lw = std_cumulative((-0.5f - v) / s);
        p_o = fmaxf(up - lw, 1e-9f);
        lp = logf(p_o);
    }
    return lp;
}

// VEC: every stream is 16-B aligned and the block spans are multiples of 4 -> float4 / int4 accesses, four elements per
// lane and iteration (the kernel is a pure streaming op: 20-28 B per element)
template <bool VEC>
__global__ __launch_bounds__(256) void gaussian_rate_kernel(const float* __restrict__ y, long long y_bs,
                                                            const int32_t* __restrict__ sym_in, const float* __restrict__ mu,
                                                            const float* __restrict__ sigma, long long ms_bs,
                                                            const float* __restrict__ table, int n_scales,
                                                            float* __restrict__ y_hat, long long yh_bs,
                                                            int32_t* __restrict__ sym_out, int32_t* __restrict__ index_out,
                                                            long long si_bs, float* __restrict__ lik_out,
                                                            double* __restrict__ partial, long long CHW) {
    __shared__ double red[4];
    __shared__ float tab[64];
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < n_scales && i < 64; i += blockDim.x) tab[i] = table[i];
    __syncthreads();
    double acc = 0.0;
    // block b of image n owns the contiguous span [b*span, (b+1)*span): the decomposition depends on CHW only,
    // so the per-image sum has one fixed order whatever the batch size
    long long span = (CHW + gridDim.x - 1) / gridDim.x;
    if (VEC) span = (span + 3) & ~3ll;
    const long long i_end = min(CHW, (long long)(blockIdx.x + 1) * span);
    const bool want_p = lik_out || partial;
    if (VEC) {
        for (long long i = (long long)blockIdx.x * span + 4 * threadIdx.x; i < i_end; i += 4 * blockDim.x) {
            const float4 m4 = *reinterpret_cast<const float4*>(mu + n * ms_bs + i);
            const float4 s4 = *reinterpret_cast<const float4*>(sigma + n * ms_bs + i);
            float4 y4 = make_float4(0.f, 0.f, 0.f, 0.f);
            int4 q4 = make_int4(0, 0, 0, 0);
            if (y) y4 = *reinterpret_cast<const float4*>(y + n * y_bs + i);
            else q4 = *reinterpret_cast<const int4*>(sym_in + n * si_bs + i);
            float4 yh4, p4 = make_float4(0.f, 0.f, 0.f, 0.f);
            int4 so4, ix4 = make_int4(0, 0, 0, 0);
            float lp = gr_element(y4.x, q4.x, y != nullptr, m4.x, s4.x, tab, n_scales, index_out != nullptr, want_p, yh4.x, so4.x, ix4.x, p4.x);
            acc += (double)lp;
            lp = gr_element(y4.y, q4.y, y != nullptr, m4.y, s4.y, tab, n_scales, index_out != nullptr, want_p, yh4.y, so4.y, ix4.y, p4.y);
            acc += (double)lp;
            lp = gr_element(y4.z, q4.z, y != nullptr, m4.z, s4.z, tab, n_scales, index_out != nullptr, want_p, yh4.z, so4.z, ix4.z, p4.z);
            acc += (double)lp;
            lp = gr_element(y4.w, q4.w, y != nullptr, m4.w, s4.w, tab, n_scales, index_out != nullptr, want_p, yh4.w, so4.w, ix4.w, p4.w);
            acc += (double)lp;
            if (y_hat) *reinterpret_cast<float4*>(y_hat + n * yh_bs + i) = yh4;
            if (sym_out) *reinterpret_cast<int4*>(sym_out + n * si_bs + i) = so4;
            if (index_out) *reinterpret_cast<int4*>(index_out + n * si_bs + i) = ix4;
            if (lik_out) *reinterpret_cast<float4*>(lik_out + n * si_bs + i) = p4;
        }
    } else {
        for (long long i = (long long)blockIdx.x * span + threadIdx.x; i < i_end; i += blockDim.x) {
            float yh, p = 0.f;
            int so, ix = 0;
            const float lp = gr_element(y ? y[n * y_bs + i] : 0.f, sym_in ? sym_in[n * si_bs + i] : 0, y != nullptr, mu[n * ms_bs + i],
                                        sigma[n * ms_bs + i], tab, n_scales, index_out != nullptr, want_p, yh, so, ix, p);
            if (y_hat) y_hat[n * yh_bs + i] = yh;
            if (sym_out) sym_out[n * si_bs + i] = so;
            if (index_out) index_out[n * si_bs + i] = ix;
            if (lik_out) lik_out[n * si_bs + i] = p;
            acc += (double)lp;
        }
    }
    if (partial) {
        const double t = block_sum_d(acc, red);
        if (threadIdx.x == 0) partial[(long long)n * gridDim.x + blockIdx.x] = t;
    }
}

// bits[n] += -(sum of the image's block partials, ascending) / ln 2
__global__ void rate_finish_kernel(const double* __restrict__ partial, int nb, float* __restrict__ bits_out) {
    const int n = threadIdx.x;          // one thread per image (N <= 1024, checked by the caller)
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += partial[(long long)n * nb + b];
    bits_out[n] += (float)(-t / 0.693147180559945309417);
}

extern "C" int dcvic_rate_blocks(long long CHW) {
    long long b = (CHW + 2047) / 2048;
    return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

extern "C" int dcvic_gaussian_rate_f32(const float* y, long long y_bs, const int32_t* sym_in, const float* mu,
                                       const float* sigma, long long ms_bs, const float* scale_table, int n_scales,
                                       float* y_hat, long long yh_bs, int32_t* sym_out, int32_t* index_out, long long si_bs,
                                       float* lik_out, float* bits_out, double* partial_ws, int N, int C, int HW, void* stream) {
    DCVIC_CHECK_ARG(mu && sigma && scale_table && N > 0 && C > 0 && HW > 0, "gaussian_rate: bad argument");
    DCVIC_CHECK_ARG((y != nullptr) != (sym_in != nullptr), "gaussian_rate: exactly one of y / sym_in must be given");
    DCVIC_CHECK_ARG(n_scales >= 2 && n_scales <= 64, "gaussian_rate: n_scales %d", n_scales);
    const long long CHW = (long long)C * HW;
    DCVIC_CHECK_ARG(ms_bs >= CHW && si_bs >= CHW, "gaussian_rate: batch stride too small");
    DCVIC_CHECK_ARG(!bits_out || partial_ws, "gaussian_rate: bits_out needs a partial-sum workspace of N * dcvic_rate_blocks(C*HW) doubles");
    DCVIC_CHECK_ARG(N <= 65535 && (!bits_out || N <= 1024), "gaussian_rate: batch too large (bits_out: <= 1024 images per call)");
    const int nb = dcvic_rate_blocks(CHW);
    dim3 grid(nb, N);
    auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // NOTE the vector form is chosen from CHW, strides and alignment only (never N): the reduction order of bits[n] is the
    // same for every batch size of a given layout
    const bool vec = (CHW & 3) == 0 && (ms_bs & 3) == 0 && (si_bs & 3) == 0 && (!y || (y_bs & 3) == 0) && (!y_hat || (yh_bs & 3) == 0) &&
                     al16(y) && al16(sym_in) && al16(mu) && al16(sigma) && al16(y_hat) && al16(sym_out) && al16(index_out) && al16(lik_out);
    if (vec)
        gaussian_rate_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(y, y_bs, sym_in, mu, sigma, ms_bs, scale_table, n_scales, y_hat,
                                                                        yh_bs, sym_out, index_out, si_bs, lik_out, bits_out ? partial_ws : nullptr, CHW);
    else
        gaussian_rate_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(y, y_bs, sym_in, mu, sigma, ms_bs, scale_table, n_scales, y_hat,
                                                                         yh_bs, sym_out, index_out, si_bs, lik_out, bits_out ? partial_ws : nullptr, CHW);
    DCVIC_CHECK_LAUNCH("gaussian_rate");
    if (bits_out) {
        rate_finish_kernel<<<1, N, 0, (hipStream_t)stream>>>(partial_ws, nb, bits_out);
        DCVIC_CHECK_LAUNCH("rate_finish");
    }
    return DCVIC_OK;
}

// Per-channel parameter pack (host side precomputes softplus(matrix) and tanh(factor)):
//   matrices [C][33]: m0 (3x1), m1..m3 (3x3 row-major), m4 (1x3);  biases [C][13]: b0..b3 (3), b4 (1);
//   factors  [C][12]: f0..f3 (3).
__device__ __forceinline__ float eb_logits(float x, const float* M, const float* Bv, const float* Fv) {
    float v[3], u[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) { v[r] = M[r] * x + Bv[r]; v[r] += Fv[r] * tanhf(v[r]); }
#pragma unroll
    for (int l = 1; l < 4; ++l) {
        const float* m = M + 3 + (l - 1) * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float a = m[r * 3 + 0] * v[0];
            a += m[r * 3 + 1] * v[1];
            a += m[r * 3 + 2] * v[2];
            a += Bv[l * 3 + r];
            u[r] = a + Fv[l * 3 + r] * tanhf(a);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) v[r] = u[r];
    }
    const float* m4 = M + 30;
    float a = m4[0] * v[0];
    a += m4[1] * v[1];
    a += m4[2] * v[2];
    return a + Bv[12];
}

__global__ __launch_bounds__(256) void eb_rate_kernel(const float* __restrict__ z, const int32_t* __restrict__ sym_in,
                                                      const float* __restrict__ matrices,
                                                      const float* __restrict__ biases, const float* __restrict__ factors,
                                                      const float* __restrict__ medians, float* __restrict__ z_hat,
                                                      int32_t* __restrict__ sym_out, float* __restrict__ lik_out,
                                                      float* __restrict__ bits_out, int C, int HW) {
    __shared__ double red[4];
    const int n = blockIdx.x;
    const long long CHW = (long long)C * HW;
    double acc = 0.0;
    for (long long i = threadIdx.x; i < CHW; i += blockDim.x) {
        const int c = (int)(i / HW);
        const float med = medians[c];
        const float sym = z ? rintf(__fsub_rn(z[n * CHW + i], med)) : (float)sym_in[n * CHW + i];
        const float zh = __fadd_rn(sym, med);
        if (z_hat) z_hat[n * CHW + i] = zh;
        if (sym_out) sym_out[n * CHW + i] = (int32_t)sym;
        if (lik_out || bits_out) {
            const float* M = matrices + c * 33;
            const float* Bv = biases + c * 13;
            const float* Fv = factors + c * 12;
            const float lower = eb_logits(zh - 0.5f, M, Bv, Fv);
            const float upper = eb_logits(zh + 0.5f, M, Bv, Fv);
            const float t = lower + upper;
            const float sg = t > 0.f ? -1.f : (t < 0.f ? 1.f : 0.f);
            const float a = 1.f / (1.f + expf(-sg * upper)), b = 1.f / (1.f + expf(-sg * lower));
            const float p = fmaxf(fabsf(a - b), 1e-9f);
            if (lik_out) lik_out[n * CHW + i] = p;
            acc += (double)logf(p);
        }
    }
    if (bits_out) {
        const double t = block_sum_d(acc, red);
        if (threadIdx.x == 0) bits_out[n] += (float)(-t / 0.693147180559945309417);
    }
}

extern "C" int dcvic_eb_rate_f32(const float* z, const int32_t* sym_in, const float* matrices, const float* biases,
                                 const float* factors, const float* medians, float* z_hat, int32_t* sym_out, float* lik_out,
                                 float* bits_out, int N, int C, int HW, void* stream) {
    DCVIC_CHECK_ARG(matrices && biases && factors && medians && N > 0 && C > 0 && HW > 0, "eb_rate: bad argument");
    DCVIC_CHECK_ARG((z != nullptr) != (sym_in != nullptr), "eb_rate: exactly one of z / sym_in must be given");
    eb_rate_kernel<<<N, 256, 0, (hipStream_t)stream>>>(z, sym_in, matrices, biases, factors, medians, z_hat, sym_out, lik_out, bits_out, C, HW);
    DCVIC_CHECK_LAUNCH("eb_rate");
    return DCVIC_OK;
}

// bits[n] = -(sum over one image of ln x) / ln 2        (likelihood_to_bit, hyperprior_vic_model.py:80-82)
__global__ __launch_bounds__(256) void neglog2_partial_kernel(const float* __restrict__ x, long long x_bs, double* __restrict__ partial,
                                                              long long CHW) {
    __shared__ double red[4];
    const int n = blockIdx.y;
    const long long span = (CHW + gridDim.x - 1) / gridDim.x;
    const long long i_end = min(CHW, (long long)(blockIdx.x + 1) * span);
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * span + threadIdx.x; i < i_end; i += blockDim.x) acc += (double)logf(x[n * x_bs + i]);
    const double t = block_sum_d(acc, red);
    if (threadIdx.x == 0) partial[(long long)n * gridDim.x + blockIdx.x] = t;
}

__global__ void neglog2_finish_kernel(const double* __restrict__ partial, int nb, float* __restrict__ bits_out) {
    const int n = threadIdx.x;
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += partial[(long long)n * nb + b];
    bits_out[n] = (float)(-t / 0.693147180559945309417);
}

extern "C" int dcvic_neglog2_sum_f32(const float* x, long long x_bs, float* bits_out, double* partial_ws, int N, long long CHW,
                                     void* stream) {
    DCVIC_CHECK_ARG(x && bits_out && partial_ws && N > 0 && N <= 1024 && CHW > 0 && x_bs >= CHW, "neglog2_sum: bad argument");
    const int nb = dcvic_rate_blocks(CHW);
    dim3 grid(nb, N);
    neglog2_partial_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, x_bs, partial_ws, CHW);
    DCVIC_CHECK_LAUNCH("neglog2_partial");
    neglog2_finish_kernel<<<1, N, 0, (hipStream_t)stream>>>(partial_ws, nb, bits_out);
    DCVIC_CHECK_LAUNCH("neglog2_finish");
    return DCVIC_OK;
}
