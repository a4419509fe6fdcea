// attn.hip -- fused single-head attention of the ldm AttnBlock (ldm/modules/diffusionmodules/model.py:178-202):
//     w = softmax_j( c^-0.5 * sum_c q[c][i] k[c][j] ),   out[c][i] = sum_j v[c][j] w[i][j]
// on NCHW planes (q, k, v, out: [C][HW] per image, tokens contiguous), head dimension = C (512 in the VQGAN).
// Flash-style: the HW x HW score matrix never exists in HBM.  fp32 MFMA (v_mfma_f32_16x16x4_f32), exact-f32 fmaf chains.
//
// Work split.  One WAVE owns 16 queries for the whole kernel and is a self-contained online-softmax unit:
//   * its 16 x C slice of q lives in registers as the B operand of the score product (C/4 VGPRs),
//   * scores of a 64-key tile are four 16x16 accumulators S^T[key][query] (keys on rows -> the softmax over keys is a
//     reduction over registers + two lane-group shuffles, and the accumulator IS the B operand of the value product:
//     lane group g of register r holds key 16*jt + 4*g + r, which is the k index g of a 16x16x4 step),
//   * the output slice O[c][query] is C/16 accumulators (C/4 VGPRs), rescaled when the running maximum moves.
// A workgroup is NW such waves (NW = 4: 64 queries; NW = 2 for small grids) that share the K / V stream: 64-row x 64-key
// chunks staged global -> registers -> LDS, double buffered, one barrier per chunk (64 MFMAs per wave per chunk).
//   K image: [channel][key], row stride 80 words (lane groups g, g+1 land 16 banks apart: conflict-free ds_read_b32).
//   V image: [channel][perm(key)], row stride 65 words, key bits permuted so that the A-operand read
//            V[c0 + lane%16][16 jt + 4 g + r] is conflict-free as well (the contraction runs over V's contiguous index).
// Determinism: every output element is a function of (q, k, v) and of the fixed 64-key tile order only -- not of the grid,
// the batch size or NW (a wave's arithmetic does not depend on its neighbours), so NW = 2 and NW = 4 are bit-identical.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AttnArgs {
    const float* q;
    const float* k;
    const float* v;
    float* out;
    long long in_bs;    // batch stride of q / k / v (elements)
    long long out_bs;
    int HW;
    int nblocks;        // workgroups
    float scale;
};

template <int C, int NW>
__global__ __launch_bounds__(NW * 64, 1) void attn_fused_kernel(const AttnArgs a) {
    constexpr int NT = NW * 64;
    constexpr int PP = 1024 / NT;          // 16-byte pieces per lane and chunk (chunk = 64 rows x 64 keys = 1024 pieces)
    constexpr int KROW = 80, VROW = 65;
    constexpr int BUF = 64 * KROW;         // words per LDS buffer (the K image is the larger one)
    constexpr int NCC = C / 64;            // chunks per tensor and key tile
    constexpr int NCH = 2 * NCC;           // chunks per key tile (K chunks, then V chunks)
    constexpr int KD = 3;                  // LDS-read prefetch distance in MFMA steps (4 MFMAs = 128 cycles each)
    extern __shared__ float lds[];         // 2 * BUF

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int il = lane & 15, g = lane >> 4;
    const int HW = a.HW;
    // XCD-aware placement: blocks b, b + 8, ... share an XCD (and its L2); give each XCD a contiguous range of logical
    // blocks so that the query blocks of one image stream its K / V through one L2
    int lb = blockIdx.x;
    if ((a.nblocks & 7) == 0) lb = (blockIdx.x & 7) * (a.nblocks >> 3) + (blockIdx.x >> 3);
    const int qblocks = HW / (16 * NW);
    const int n = lb / qblocks;
    const int i0 = (lb % qblocks) * (16 * NW) + wave * 16;
    const float* Q = a.q + (long long)n * a.in_bs;
    const float* K = a.k + (long long)n * a.in_bs;
    const float* V = a.v + (long long)n * a.in_bs;

    // q slice as the B operand: lane (query il, k-group g) of step s holds q[4 s + g][i0 + il]
    float qreg[C / 4];
#pragma unroll
    for (int s = 0; s < C / 4; ++s) qreg[s] = Q[(long long)(4 * s + g) * HW + i0 + il];

    f32x4 o[C / 16];
#pragma unroll
    for (int t = 0; t < C / 16; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 S[4];
    float m_run = -INFINITY, l_run = 0.f;

    // per-lane piece coordinates of the cooperative chunk copy (piece = 4 consecutive keys of one row)
    int prow[PP], pq[PP];
#pragma unroll
    for (int u = 0; u < PP; ++u) {
        const int id = u * NT + tid;
        prow[u] = id >> 4;
        pq[u] = id & 15;
    }
    int poff[PP];
#pragma unroll
    for (int u = 0; u < PP; ++u) poff[u] = prow[u] * HW + 4 * pq[u];
    f32x4 st[PP];
    const int nkt = HW / 64;

    // wave-uniform chunk base (SGPRs) + a 32-bit per-lane piece offset: the saddr form of global_load, 1 VGPR per piece
#define ATTN_ISSUE(T, row0, key0)                                                                             \
    {                                                                                                         \
        const float* cb_ = (T) + (long long)(row0) * HW + (key0);                                             \
        _Pragma("unroll") for (int u = 0; u < PP; ++u) st[u] = *reinterpret_cast<const f32x4*>(cb_ + poff[u]); \
    }
#define ATTN_STAGE_K(buf)                                                                                     \
    _Pragma("unroll") for (int u = 0; u < PP; ++u) *reinterpret_cast<f32x4*>((buf) + prow[u] * KROW + 4 * pq[u]) = st[u];
    // piece index = key >> 2 = (jt1 jt0 g1 g0); position bits (jt0 g0 jt1 g1 r1 r0)
#define ATTN_STAGE_V(buf)                                                                                     \
    _Pragma("unroll") for (int u = 0; u < PP; ++u) {                                                          \
        const int q4 = pq[u];                                                                                 \
        const int pos = 32 * ((q4 >> 2) & 1) + 16 * (q4 & 1) + 8 * (q4 >> 3) + 4 * ((q4 >> 1) & 1);           \
        float* d = (buf) + prow[u] * VROW + pos;                                                              \
        d[0] = st[u][0]; d[1] = st[u][1]; d[2] = st[u][2]; d[3] = st[u][3];                                   \
    }

    const int kbase = g * KROW + il;
    const int vbase = il * VROW + 16 * (g & 1) + 4 * (g >> 1);
    // A-operand reads of MFMA step `st_` of a chunk image: K image -> 4 key tiles, V image -> 4 channel tiles
#define ATTN_READ_K(buf, st_, dst)                                                                            \
    _Pragma("unroll") for (int x_ = 0; x_ < 4; ++x_) (dst)[x_] = (buf)[kbase + (st_) * 4 * KROW + 16 * x_];
#define ATTN_READ_V(buf, st_, dst)                                                                            \
    _Pragma("unroll") for (int x_ = 0; x_ < 4; ++x_)                                                          \
        (dst)[x_] = (buf)[vbase + x_ * 16 * VROW + 32 * (((st_) >> 2) & 1) + 8 * ((st_) >> 3) + ((st_) & 3)];

    // Two LDS buffers, one barrier per chunk:
    //   barrier(n) : chunk n is complete in buf[n&1]; every wave has finished reading buf[(n+1)&1] (chunk n-1)
    //   then       : registers (chunk n+1, requested during chunk n-1) -> buf[(n+1)&1]; request chunk n+2; 64 MFMAs
    // (a three-buffer ring with the barrier mid-chunk and operand reads running ahead across the chunk seam was built and
    //  measured slower: it needs ~30 more live registers and hipcc spills inside the loop at the 512-register cap)
    ATTN_ISSUE(K, 0, 0)
    ATTN_STAGE_K(lds)
    if (NCC > 1) { ATTN_ISSUE(K, 64, 0) } else { ATTN_ISSUE(V, 0, 0) }

    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int rem = 0; rem < NCH; ++rem) {
            float* b0 = lds + (rem & 1) * BUF;           // NCH is even: the buffer parity of a chunk is compile-time
            float* b1 = lds + ((rem + 1) & 1) * BUF;
            const int r1 = (rem + 1) % NCH;
            __syncthreads();
            {   // registers hold chunk rem + 1 -> LDS; then request chunk rem + 2
                const bool have1 = !(kt == nkt - 1 && rem == NCH - 1);
                if (have1) { if (r1 < NCC) { ATTN_STAGE_K(b1) } else { ATTN_STAGE_V(b1) } }
                const int r2 = (rem + 2) % NCH;
                const int kt2 = kt + ((rem + 2) >= NCH ? 1 : 0);
                if (kt2 < nkt) {
                    if (r2 < NCC) { ATTN_ISSUE(K, 64 * r2, kt2 * 64) } else { ATTN_ISSUE(V, 64 * (r2 - NCC), kt2 * 64) }
                }
                __builtin_amdgcn_sched_barrier(0);    // keep the global loads HERE: a whole chunk of MFMAs covers their latency
            }
            if (rem == 0) {
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) S[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            // 16 MFMA steps (4 MFMAs each).  Operand reads run KD steps ahead of their MFMAs and sched_barrier pins that
            // order (left alone, hipcc waits lgkmcnt(0) after each read pair and exposes the LDS latency to every 2nd MFMA)
            float av[16][4];
#pragma unroll
            for (int s = 0; s < KD; ++s) {
                if (rem < NCC) { ATTN_READ_K(b0, s, av[s]) } else { ATTN_READ_V(b0, s, av[s]) }
            }
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int f = s + KD;
                if (f < 16) {
                    if (rem < NCC) { ATTN_READ_K(b0, f, av[f]) } else { ATTN_READ_V(b0, f, av[f]) }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (rem < NCC) {      // scores: S^T[key][query] += K[c][key] * q[c][query], 4 channels per step
#pragma unroll
                    for (int jt = 0; jt < 4; ++jt)
                        S[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][jt], qreg[16 * rem + s], S[jt], 0, 0, 0);
                } else {              // values: O[c][query] += V[c][key] * P[key][query]; step s = 4 jt + r is key 16 jt + 4 g + r
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
                        o[4 * (rem - NCC) + tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][tt], S[s >> 2][s & 3], o[4 * (rem - NCC) + tt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (rem == NCC - 1) {
                // online softmax over this tile's 64 keys (16 values per lane, 4 lane groups per query)
                float mt = -INFINITY;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { S[jt][r] *= a.scale; mt = fmaxf(mt, S[jt][r]); }
                mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
                mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
                const float m_new = fmaxf(m_run, mt);
                const float alpha = expf(m_run - m_new);
                float rs = 0.f;
#pragma unroll
                for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const float p = expf(S[jt][r] - m_new); S[jt][r] = p; rs += p; }
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                l_run = l_run * alpha + rs;
                m_run = m_new;
                // rescale the output slice only when some query of this wave moved its maximum (x 1.0f is exact, so
                // skipping it changes no value); after the first tiles that is rare
                if (__any(alpha != 1.f)) {
#pragma unroll
                    for (int t = 0; t < C / 16; ++t) o[t] *= alpha;
                }
            }
        }
    }
    // out[c][i] = O / l ; tile t, register r of lane (il, g) is channel 16 t + 4 g + r
    float* O = a.out + (long long)n * a.out_bs;
#pragma unroll
    for (int t = 0; t < C / 16; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) O[(long long)(16 * t + 4 * g + r) * HW + i0 + il] = o[t][r] / l_run;
}

template <int C, int NW>
static int launch_attn(const AttnArgs& A, hipStream_t st) {
    static std::atomic<unsigned> attr_mask{0};
    auto k = attn_fused_kernel<C, NW>;
    const size_t lds = (size_t)2 * 64 * 80 * sizeof(float);
    if (DcvicAttrOnce once_{attr_mask})
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    k<<<A.nblocks, NW * 64, lds, st>>>(A);
    DCVIC_CHECK_LAUNCH("attn_fused");
    return DCVIC_OK;
}

extern "C" int dcvic_attn_fused_f32(const float* q, const float* k, const float* v, long long in_bs, float* out, long long out_bs,
                                    int N, int C, int HW, float scale, int force_nw, void* stream) {
    DCVIC_CHECK_ARG(q && k && v && out, "attn_fused: null pointer");
    DCVIC_CHECK_ARG(N > 0 && HW > 0 && (HW % 64) == 0, "attn_fused: HW=%d must be a positive multiple of 64 (images are padded to x64)", HW);
    DCVIC_CHECK_ARG(C == 512 || C == 256 || C == 128, "attn_fused: C=%d not instantiated (128 / 256 / 512)", C);
    DCVIC_CHECK_ARG(((uintptr_t)k % 16) == 0 && ((uintptr_t)v % 16) == 0 && (in_bs % 4) == 0, "attn_fused: k / v planes must be 16-byte aligned");
    AttnArgs A{q, k, v, out, in_bs, out_bs, HW, 0, scale};
    const long long wg4 = (long long)N * (HW / 64);
    // small grids: two waves per workgroup, 2x the workgroups (same values); not at C = 512, where the 8-piece staging
    // registers of a 2-wave workgroup do not fit beside the 272 q / O / S registers (hipcc spills)
    int nw = (wg4 >= dcvic_num_cu() || C == 512) ? 4 : 2;
    if ((force_nw == 2 && C != 512) || force_nw == 4) nw = force_nw;
    A.nblocks = (int)(nw == 4 ? wg4 : wg4 * 2);
    hipStream_t st = (hipStream_t)stream;
    if (C == 512) return launch_attn<512, 4>(A, st);
    if (C == 256) return nw == 4 ? launch_attn<256, 4>(A, st) : launch_attn<256, 2>(A, st);
    return nw == 4 ? launch_attn<128, 4>(A, st) : launch_attn<128, 2>(A, st);
}
