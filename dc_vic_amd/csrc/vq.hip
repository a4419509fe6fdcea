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

// ---------------------------------------------------------------------------------------------------------------------
// Sharded-codebook search on the matrix cores (D == 4, n_e a multiple of 256, ANY codebook size -- 16 384 x 4 and beyond,
// which do not fit one LDS image; reference: hyperprior_vic_model.py:149-150,170-188 `_vq_quantize_split`).
//   * the codebook streams through LDS in SHARDS of 256 codes ([dim][code] image + squared norms, 5.3 KiB);
//   * the dot products are v_mfma_f32_16x16x4_f32 with K = 4 = the code dimension: A = 16 codes x 4 dims, B = 4 dims x 16
//     latent vectors, C = 0.  The instruction is an exact k-ordered fmaf chain (tools/mfma_order.hip), i.e. bit for bit
//     the generic kernel's  fma(z3,e3, fma(z2,e2, fma(z1,e1, z0*e0)));
//   * lane (vector v = lane % 16, group g = lane / 16) then owns the distances of ITS vector to codes 16 t + 4 g + r of every
//     tile t: d = fma(-2, dot, |z|^2 + |e|^2) (the reference's expanded form, same roundings as the kernels above), a running
//     (best, index) per lane with a strict `<` in ascending code order;
//   * the four lane groups of a vector are merged by a WAVEFRONT SHUFFLE (d, idx) reduction (xor 16, xor 32): smaller
//     distance wins, equal distances -> smaller index, so the result is torch.argmin's first minimum.
// A wave owns VT = 4 vector tiles (64 latent vectors), a workgroup 256 vectors; z is read once, coalesced along pixels.
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define VQS 256                     // codes per shard
#define VQS_ROW (VQS + 16)          // [dim][code] row stride: lane groups g, g+1 land 16 banks apart (conflict-free)

__global__ __launch_bounds__(256) void vq_shard_mfma_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                            int64_t* __restrict__ idx, float* __restrict__ zq,
                                                            float* __restrict__ feat, int HW, int n_e) {
    constexpr int VT = 4;
    __shared__ __attribute__((aligned(16))) float Es[4 * VQS_ROW];
    __shared__ __attribute__((aligned(16))) float E2s[VQS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vl = lane & 15, g = lane >> 4;
    const int n = blockIdx.y;
    const int pbase = blockIdx.x * 256 + wave * 64;
    const float* zn = z + (long long)n * 4 * HW;
    float zg[VT], z2[VT], zall[VT][4], best[VT];
    int bi[VT];
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
        const int p = min(pbase + vt * 16 + vl, HW - 1);
        zg[vt] = zn[(long long)g * HW + p];                          // B operand: z[dim g][vector vl]
#pragma unroll
        for (int d = 0; d < 4; ++d) zall[vt][d] = __shfl(zg[vt], vl + 16 * d, 64);
        float s2 = __fmul_rn(zall[vt][0], zall[vt][0]);
#pragma unroll
        for (int d = 1; d < 4; ++d) s2 = __fadd_rn(s2, __fmul_rn(zall[vt][d], zall[vt][d]));
        z2[vt] = s2; best[vt] = INFINITY; bi[vt] = 0;
    }
    for (int s0 = 0; s0 < n_e; s0 += VQS) {
        __syncthreads();                                             // the previous shard has been consumed
        {   // stage shard [s0, s0 + 256): thread tid owns code s0 + tid (one float4), transposed into the [dim][code] image
            const f32x4v e = *reinterpret_cast<const f32x4v*>(cb + (long long)(s0 + tid) * 4);
            Es[0 * VQS_ROW + tid] = e[0]; Es[1 * VQS_ROW + tid] = e[1]; Es[2 * VQS_ROW + tid] = e[2]; Es[3 * VQS_ROW + tid] = e[3];
            float sq = __fmul_rn(e[0], e[0]);
            sq = __fadd_rn(sq, __fmul_rn(e[1], e[1])); sq = __fadd_rn(sq, __fmul_rn(e[2], e[2])); sq = __fadd_rn(sq, __fmul_rn(e[3], e[3]));
            E2s[tid] = sq;
        }
        __syncthreads();
        float ea[16];                                                // A operands of the 16 code tiles: E[16 t + vl][dim g]
#pragma unroll
        for (int t = 0; t < 16; ++t) ea[t] = Es[g * VQS_ROW + 16 * t + vl];
#pragma unroll
        for (int vt = 0; vt < VT; ++vt) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const f32x4v dot = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[t], zg[vt], (f32x4v){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                const f32x4v e2 = *reinterpret_cast<const f32x4v*>(&E2s[16 * t + 4 * g]);
#pragma unroll
                for (int r = 0; r < 4; ++r) {                        // code s0 + 16 t + 4 g + r, ascending inside the lane
                    const float dist = fmaf(-2.f, dot[r], __fadd_rn(z2[vt], e2[r]));
                    const bool lt = dist < best[vt];
                    best[vt] = lt ? dist : best[vt];
                    bi[vt] = lt ? (s0 + 16 * t + 4 * g + r) : bi[vt];
                }
            }
        }
    }
    // wavefront (d, idx) reduction over the four lane groups of each vector
#pragma unroll
    for (int vt = 0; vt < VT; ++vt) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float od = __shfl_xor(best[vt], o, 64);
            const int oi = __shfl_xor(bi[vt], o, 64);
            const bool take = (od < best[vt]) || (od == best[vt] && oi < bi[vt]);
            best[vt] = take ? od : best[vt];
            bi[vt] = take ? oi : bi[vt];
        }
        const int p = pbase + vt * 16 + vl;
        if (g != 0 || p >= HW) continue;
        const int b = bi[vt];
        idx[(long long)n * HW + p] = (int64_t)b;
        float q[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) q[d] = __fadd_rn(zall[vt][d], __fsub_rn(cb[(long long)b * 4 + d], zall[vt][d]));
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
    DCVIC_CHECK_ARG(N <= 65535, "vq_argmin: batch too large");
    // DCVIC_VQ_KERNEL = shard | scalar | lds forces a variant (measurements, tests); every variant returns the same indices
    const char* force = getenv("DCVIC_VQ_KERNEL");
    const bool fits_lds = (size_t)n_e * (D + 1) * 4 <= 160 * 1024;
    const bool shard_ok = D == 4 && n_e % VQS == 0 && ((uintptr_t)codebook % 16) == 0;
    if (shard_ok && (n_e > 1024 || (force && force[0] == 's' && force[1] == 'h'))) {
        dim3 grids(dcvic_cdiv(HW, 256), N);
        vq_shard_mfma_kernel<<<grids, 256, 0, (hipStream_t)stream>>>(z, codebook, idx, zq, feat, HW, n_e);
        DCVIC_CHECK_LAUNCH("vq_argmin(shard)");
        return DCVIC_OK;
    }
    DCVIC_CHECK_ARG(fits_lds, "vq_argmin: codebook %d x %d fits neither one LDS image nor the sharded kernel (D = 4, n_e %% 256 == 0)", n_e, D);
    dim3 grid(dcvic_cdiv(HW, 256), N);
    const size_t lds = (size_t)n_e * (D + 1) * sizeof(float);
    if (D == 4 && n_e % 8 == 0 && n_e <= 1024 && !(force && force[0] == 'l')) {
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
        static std::atomic<unsigned> m4{0};
        if (DcvicAttrOnce once_{m4}) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_argmin_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        vq_argmin_kernel<4><<<grid, 256, lds, (hipStream_t)stream>>>(z, codebook, idx, zq, feat, HW, n_e);
    } else {
        static std::atomic<unsigned> m8{0};
        if (DcvicAttrOnce once_{m8}) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(vq_argmin_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
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
