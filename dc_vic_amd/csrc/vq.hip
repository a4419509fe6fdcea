// vq.hip -- VQ nearest-codeword search and the decoder-side argmax -> codebook LUT.
//
// dcvic_vq_argmin_f32 replaces VectorQuantizer2.forward (taming/modules/vqvae/quantize.py:271-312):
//   d[j] = (sum_d z_d^2 + sum_d e_jd^2) - 2 * (z . e_j)        -- the reference's EXPANDED form
//   idx  = first argmin_j d[j];  z_q = z + (e_idx - z)          -- straight-through arithmetic (line 298)
// The association order of the reference expression is kept: squares are rounded products summed
// left to right (torch.sum over 4 contiguous elements), the dot is an fmaf chain over d (what a
// sgemm micro-kernel does for K = 4), then one add and one subtract.  __f*_rn intrinsics stop the
// compiler from contracting these into different FMAs.
// Layout: z is NCHW, so for a fixed channel consecutive lanes read consecutive pixels (coalesced
// 256-B wave reads); the codebook (n_e x D) and its squared norms are staged once per workgroup in
// LDS and read as wave-uniform broadcasts; each lane owns one latent vector and scans the codes.
// Roofline: 16 B read + 8 B index (+16 B z_q, + (D+n_e)*4 B one-hot feature when requested) per vector.
#include "common.h"

#define VQ_MAXD 8

template <int D>
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                        int64_t* __restrict__ idx, float* __restrict__ zq,
                                                        float* __restrict__ feat, int HW, int n_e) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* E = sm;                 // [n_e][D]
    float* E2 = sm + n_e * D;      // [n_e]
    for (int i = threadIdx.x; i < n_e * D; i += blockDim.x) E[i] = cb[i];
    __syncthreads();
    for (int j = threadIdx.x; j < n_e; j += blockDim.x) {
        float s = __fmul_rn(E[j * D], E[j * D]);
#pragma unroll
        for (int d = 1; d < D; ++d) s = __fadd_rn(s, __fmul_rn(E[j * D + d], E[j * D + d]));
        E2[j] = s;
    }
    __syncthreads();
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* zp = z + (long long)n * D * HW + p;
    float zv[D];
#pragma unroll
    for (int d = 0; d < D; ++d) zv[d] = zp[(long long)d * HW];
    float z2 = __fmul_rn(zv[0], zv[0]);
#pragma unroll
    for (int d = 1; d < D; ++d) z2 = __fadd_rn(z2, __fmul_rn(zv[d], zv[d]));
    float best = INFINITY;
    int bi = 0;
    for (int j = 0; j < n_e; ++j) {
        float dot = __fmul_rn(zv[0], E[j * D]);
#pragma unroll
        for (int d = 1; d < D; ++d) dot = fmaf(zv[d], E[j * D + d], dot);
        const float dist = __fsub_rn(__fadd_rn(z2, E2[j]), __fmul_rn(2.0f, dot));
        if (dist < best) { best = dist; bi = j; }
    }
    idx[(long long)n * HW + p] = (int64_t)bi;
    float q[D];
#pragma unroll
    for (int d = 0; d < D; ++d) q[d] = __fadd_rn(zv[d], __fsub_rn(E[bi * D + d], zv[d]));
    if (zq) {
#pragma unroll
        for (int d = 0; d < D; ++d) zq[(long long)n * D * HW + (long long)d * HW + p] = q[d];
    }
    if (feat) {
        float* fp = feat + (long long)n * (D + n_e) * HW + p;
#pragma unroll
        for (int d = 0; d < D; ++d) fp[(long long)d * HW] = q[d];
        for (int j = 0; j < n_e; ++j) fp[(long long)(D + j) * HW] = (j == bi) ? 1.0f : 0.0f;
    }
}

extern "C" int dcvic_vq_argmin_f32(const float* z, const float* codebook, int64_t* idx, float* zq, float* feat, int N, int D,
                                   int HW, int n_e, void* stream) {
    DCVIC_CHECK_ARG(z && codebook && idx && N > 0 && HW > 0 && n_e > 0, "vq_argmin: bad argument");
    DCVIC_CHECK_ARG(D == 4 || D == 8, "vq_argmin: embed_dim %d unsupported (4 or 8)", D);
    DCVIC_CHECK_ARG((size_t)n_e * (D + 1) * 4 <= 160 * 1024, "vq_argmin: codebook %d x %d does not fit LDS", n_e, D);
    DCVIC_CHECK_ARG(N <= 65535, "vq_argmin: batch too large");
    dim3 grid(dcvic_cdiv(HW, 256), N);
    const size_t lds = (size_t)n_e * (D + 1) * sizeof(float);
    if (D == 4) {
        static bool set4 = false;
        if (!set4) { hipFuncSetAttribute(reinterpret_cast<const void*>(vq_argmin_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set4 = true; }
        vq_argmin_kernel<4><<<grid, 256, lds, (hipStream_t)stream>>>(z, codebook, idx, zq, feat, HW, n_e);
    } else {
        static bool set8 = false;
        if (!set8) { hipFuncSetAttribute(reinterpret_cast<const void*>(vq_argmin_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set8 = true; }
        vq_argmin_kernel<8><<<grid, 256, lds, (hipStream_t)stream>>>(z, codebook, idx, zq, feat, HW, n_e);
    }
    DCVIC_CHECK_LAUNCH("vq_argmin");
    return DCVIC_OK;
}

// logits [N][n_e][HW] -> idx = first argmax over channels; latent[c] = pq_b[c] + sum_d pq_w[c][d] * codebook[idx][d]
__global__ __launch_bounds__(256) void argmax_lut_kernel(const float* __restrict__ logits, int64_t* __restrict__ idx,
                                                         float* __restrict__ latent, const float* __restrict__ cb,
                                                         const float* __restrict__ pq_w, const float* __restrict__ pq_b,
                                                         int n_e, int D, int HW) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* lp = logits + (long long)n * n_e * HW + p;
    float best = lp[0];
    int bi = 0;
    for (int j = 1; j < n_e; ++j) {
        const float v = lp[(long long)j * HW];
        if (v > best) { best = v; bi = j; }
    }
    if (idx) idx[(long long)n * HW + p] = (int64_t)bi;
    if (latent) {
        for (int c = 0; c < D; ++c) {
            float a = 0.f;
            for (int d = 0; d < D; ++d) a = fmaf(pq_w[c * D + d], cb[bi * D + d], a);
            if (pq_b) a += pq_b[c];
            latent[(long long)n * D * HW + (long long)c * HW + p] = a;
        }
    }
}

extern "C" int dcvic_argmax_lut_f32(const float* logits, int64_t* idx, float* latent, const float* codebook, const float* pq_w,
                                    const float* pq_b, int N, int n_e, int D, int HW, void* stream) {
    DCVIC_CHECK_ARG(logits && (idx || latent) && N > 0 && n_e > 0 && HW > 0, "argmax_lut: bad argument");
    DCVIC_CHECK_ARG(!latent || (codebook && pq_w && D > 0), "argmax_lut: latent output needs codebook and post_quant weights");
    DCVIC_CHECK_ARG(N <= 65535, "argmax_lut: batch too large");
    dim3 grid(dcvic_cdiv(HW, 256), N);
    argmax_lut_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(logits, idx, latent, codebook, pq_w, pq_b, n_e, D, HW);
    DCVIC_CHECK_LAUNCH("argmax_lut");
    return DCVIC_OK;
}
