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

// D == 4 fast path (n_e a multiple of 8).  VALU-bound, so the design minimises instructions per (vector, code):
//  * each lane owns 2*NP latent vectors; a packed-fp32 instruction (v_pk_mul/fma/add_f32) evaluates one code for a PAIR
//    of vectors, the code's components coming from SGPRs (the codebook is read with wave-uniform indices, i.e. s_load
//    through the scalar cache -- no LDS traffic, no VGPR copies);
//  * codes are scanned in groups of 8: a v_min3 tree gives the group minimum, one compare + two selects keep
//    (best, winning group); the index inside the group is resolved once at the end by recomputing that group's 8
//    distances with the same instructions' roundings and taking the first one equal to the minimum.
// Same operation order / roundings as the generic kernel: dot = fma(z3,e3,fma(z2,e2,fma(z1,e1,z0*e0))),
// t = |z|^2 + |e|^2, d = fma(-2, dot, t) (== t - 2*dot with one rounding); a strictly smaller group minimum replaces the
// running one, so the first minimum in ascending code order wins exactly as torch.argmin's.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 splat2(float v) { return (f32x2){v, v}; }

template <int NP>
__global__ __launch_bounds__(256) void vq_argmin4_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                         int64_t* __restrict__ idx, float* __restrict__ zq,
                                                         float* __restrict__ feat, int HW, int n_e) {
    extern __shared__ __attribute__((aligned(16))) float E2[];    // [n_e] squared norms
    for (int j = threadIdx.x; j < n_e; j += blockDim.x) {
        float sq = __fmul_rn(cb[j * 4], cb[j * 4]);
#pragma unroll
        for (int d = 1; d < 4; ++d) sq = __fadd_rn(sq, __fmul_rn(cb[j * 4 + d], cb[j * 4 + d]));
        E2[j] = sq;
    }
    __syncthreads();
    constexpr int V = 2 * NP;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * (256 * V) + threadIdx.x;
    float zv[V][4], z2[V], best[V];
    int grp[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int p = min(p0 + 256 * v, HW - 1);
#pragma unroll
        for (int d = 0; d < 4; ++d) zv[v][d] = z[(long long)n * 4 * HW + (long long)d * HW + p];
        float s2 = __fmul_rn(zv[v][0], zv[v][0]);
#pragma unroll
        for (int d = 1; d < 4; ++d) s2 = __fadd_rn(s2, __fmul_rn(zv[v][d], zv[v][d]));
        z2[v] = s2; best[v] = INFINITY; grp[v] = 0;
    }
    f32x2 zp[NP][4], z2p[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int d = 0; d < 4; ++d) zp[q][d] = (f32x2){zv[2 * q][d], zv[2 * q + 1][d]};
        z2p[q] = (f32x2){z2[2 * q], z2[2 * q + 1]};
    }
    for (int g = 0; g < n_e; g += 8) {
        f32x2 dd[NP][8];
        // component-major over the 8 codes: 8*NP independent chains, so no dependent packed op issues back to back
        float en[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) en[c] = E2[g + c];
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int q = 0; q < NP; ++q) dd[q][c] = zp[q][0] * splat2(cb[(g + c) * 4 + 0]);
#pragma unroll
        for (int d = 1; d < 4; ++d)
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int q = 0; q < NP; ++q) dd[q][c] = __builtin_elementwise_fma(zp[q][d], splat2(cb[(g + c) * 4 + d]), dd[q][c]);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int q = 0; q < NP; ++q)
                dd[q][c] = __builtin_elementwise_fma(splat2(-2.f), dd[q][c], z2p[q] + splat2(en[c]));
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            float m0 = __builtin_fminf(__builtin_fminf(dd[q][0].x, dd[q][1].x), dd[q][2].x);
            float m1 = __builtin_fminf(__builtin_fminf(dd[q][3].x, dd[q][4].x), dd[q][5].x);
            m0 = __builtin_fminf(__builtin_fminf(dd[q][6].x, dd[q][7].x), m0);
            const float mx = __builtin_fminf(m0, m1);
            float n0 = __builtin_fminf(__builtin_fminf(dd[q][0].y, dd[q][1].y), dd[q][2].y);
            float n1 = __builtin_fminf(__builtin_fminf(dd[q][3].y, dd[q][4].y), dd[q][5].y);
            n0 = __builtin_fminf(__builtin_fminf(dd[q][6].y, dd[q][7].y), n0);
            const float my = __builtin_fminf(n0, n1);
            const bool bx = mx < best[2 * q], by = my < best[2 * q + 1];
            best[2 * q] = bx ? mx : best[2 * q];         grp[2 * q] = bx ? g : grp[2 * q];
            best[2 * q + 1] = by ? my : best[2 * q + 1]; grp[2 * q + 1] = by ? g : grp[2 * q + 1];
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int p = p0 + 256 * v;
        if (p >= HW) continue;
        int b = grp[v];
        {   // resolve the index inside the winning group (first distance equal to the minimum)
            int found = -1;
#pragma unroll
            for (int c = 7; c >= 0; --c) {
                const float* e = cb + (grp[v] + c) * 4;
                float dot = __fmul_rn(zv[v][0], e[0]);
                dot = fmaf(zv[v][1], e[1], dot);
                dot = fmaf(zv[v][2], e[2], dot);
                dot = fmaf(zv[v][3], e[3], dot);
                const float dist = fmaf(-2.f, dot, __fadd_rn(z2[v], E2[grp[v] + c]));
                found = (dist == best[v]) ? c : found;
            }
            b += max(found, 0);
        }
        idx[(long long)n * HW + p] = (int64_t)b;
        float q[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) q[d] = __fadd_rn(zv[v][d], __fsub_rn(cb[b * 4 + d], zv[v][d]));
        if (zq) {
#pragma unroll
            for (int d = 0; d < 4; ++d) zq[(long long)n * 4 * HW + (long long)d * HW + p] = q[d];
        }
        if (feat) {
            float* fp = feat + (long long)n * (4 + n_e) * HW + p;
#pragma unroll
            for (int d = 0; d < 4; ++d) fp[(long long)d * HW] = q[d];
            for (int j = 0; j < n_e; ++j) fp[(long long)(4 + j) * HW] = (j == b) ? 1.0f : 0.0f;
        }
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
    if (D == 4 && n_e % 8 == 0) {
        // big launches: 4 vectors per lane (1024-vector tiles); small ones keep 512-vector tiles so the chip still fills
        const bool big = (long long)N * dcvic_cdiv(HW, 1024) >= 2048;
        if (big) {
            dim3 grid4(dcvic_cdiv(HW, 1024), N);
            vq_argmin4_kernel<2><<<grid4, 256, (size_t)n_e * sizeof(float), (hipStream_t)stream>>>(z, codebook, idx, zq, feat, HW, n_e);
        } else {
            dim3 grid4(dcvic_cdiv(HW, 512), N);
            vq_argmin4_kernel<1><<<grid4, 256, (size_t)n_e * sizeof(float), (hipStream_t)stream>>>(z, codebook, idx, zq, feat, HW, n_e);
        }
    } else if (D == 4) {
        static bool set4g = false;
        if (!set4g) { hipFuncSetAttribute(reinterpret_cast<const void*>(vq_argmin_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set4g = true; }
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
