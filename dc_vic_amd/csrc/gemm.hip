// gemm.hip -- batched strided GEMM on the fp32 MFMA (attention score / value products of the ldm
// AttnBlock, model.py:186-196).  C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][k][n], k ascending
// (one fmaf per term), so results do not depend on the grid or the batch size.
// 128x128 (or, when that would leave CUs idle, 64x64) output tile per workgroup, 4 waves as 2x2, each wave
// 2x2 (1x1) MFMA 32x32x2 tiles; operands are staged through LDS as [k][m] / [k][n] images (+1 word of row padding)
// from arbitrary element strides.  The tile size never changes a value (one k-ordered chain per output).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GK 16

__device__ float dcvic_gemm_zero[4];    // zero-initialised: source of out-of-range elements

// Staging: element (k, m) of a stage is chunk-invariant per thread, so its global offset, LDS slot and row validity are derived once;
// the next stage's 16 loads are issued BEFORE this stage's MFMAs and stored to the other LDS buffer after them (one barrier per stage).
// Out-of-range elements read a zero word: a select on the loaded VALUE makes hipcc guard every load with its own exec-masked block and
// `s_waitcnt vmcnt(0)` (sixteen serial round trips per stage -- the first build ran at 0.18 of the matrix pipe).
template <int GT>
__global__ __launch_bounds__(256, 2) void bgemm_kernel(const dcvic_gemm_args g) {
    constexpr int GTP = GT + 1;
    constexpr int WT = GT / 2;            // per-wave tile edge
    constexpr int MI = WT / 32;           // 32x32 accumulators per wave and dimension
    constexpr int NE = GK * GT / 256;     // elements per thread, operand and stage
    __shared__ float As[2][GK * GTP];
    __shared__ float Bs[2][GK * GTP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lane_k = lane >> 5, lane_j = lane & 31;
    const int tiles_n = (g.N + GT - 1) / GT;
    const int bt = blockIdx.x;
    const int tn = bt % tiles_n, tm = bt / tiles_n;
    const int b = blockIdx.y;
    const int m0 = tm * GT, n0 = tn * GT;
    const float* A = g.A + (long long)b * g.a_bs;
    const float* B = g.B + (long long)b * g.b_bs;

    f32x16 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool a_kfast = (g.a_ks == 1);   // k contiguous in memory -> walk k fastest for coalescing
    const bool b_kfast = (g.b_ks == 1);
    long long a_off[NE], b_off[NE];       // -1: row / column outside the matrix
    int a_pk[NE], b_pk[NE];               // LDS slot << 8 | k
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        const int e = tid + u * 256;
        int kk, mm;
        if (a_kfast) { kk = e % GK; mm = e / GK; } else { mm = e % GT; kk = e / GT; }
        a_off[u] = (m0 + mm < g.M) ? (long long)(m0 + mm) * g.a_ms + (long long)kk * g.a_ks : -1;
        a_pk[u] = ((kk * GTP + mm) << 8) | kk;
        int kb, nn;
        if (b_kfast) { kb = e % GK; nn = e / GK; } else { nn = e % GT; kb = e / GT; }
        b_off[u] = (n0 + nn < g.N) ? (long long)kb * g.b_ks + (long long)(n0 + nn) * g.b_ns : -1;
        b_pk[u] = ((kb * GTP + nn) << 8) | kb;
    }
    float ra[NE], rb[NE];
    auto fetch = [&](int k0) {
        const long long ak = (long long)k0 * g.a_ks, bk = (long long)k0 * g.b_ks;
#pragma unroll
        for (int u = 0; u < NE; ++u) {
            ra[u] = *((a_off[u] >= 0 && k0 + (a_pk[u] & 255) < g.K) ? A + (a_off[u] + ak) : dcvic_gemm_zero);
            rb[u] = *((b_off[u] >= 0 && k0 + (b_pk[u] & 255) < g.K) ? B + (b_off[u] + bk) : dcvic_gemm_zero);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < NE; ++u) { As[buf][a_pk[u] >> 8] = ra[u]; Bs[buf][b_pk[u] >> 8] = rb[u]; }
    };
    fetch(0); stage(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < g.K; k0 += GK, buf ^= 1) {
        const bool more = k0 + GK < g.K;
        if (more) fetch(k0 + GK);
#pragma unroll
        for (int ks = 0; ks < GK / 2; ++ks) {
            float a[MI], bb[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = As[buf][(2 * ks + lane_k) * GTP + wm * WT + i * 32 + lane_j];
#pragma unroll
            for (int j = 0; j < MI; ++j) bb[j] = Bs[buf][(2 * ks + lane_k) * GTP + wn * WT + j * 32 + lane_j];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bb[j], acc[i][j], 0, 0, 0);
        }
        if (more) stage(buf ^ 1);           // everyone finished reading buf ^ 1 before the previous barrier
        __syncthreads();
    }
    float* C = g.C + (long long)b * g.c_bs;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int nn = n0 + wn * WT + j * 32 + lane_j;
            if (nn >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + wm * WT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lane_k;
                if (mm < g.M) C[(long long)mm * g.c_ms + nn] = g.alpha * acc[i][j][r];
            }
        }
}

extern "C" int dcvic_bgemm_f32(const dcvic_gemm_args* a, void* stream) {
    DCVIC_CHECK_ARG(a && a->A && a->B && a->C, "bgemm: null pointer");
    DCVIC_CHECK_ARG(a->batch > 0 && a->M > 0 && a->N > 0 && a->K > 0, "bgemm: bad sizes");
    DCVIC_CHECK_ARG(a->batch <= 65535, "bgemm: batch too large");
    static int num_cu = 0;
    if (num_cu == 0) {
        int dev = 0, cu = 0;
        num_cu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) ? cu : 256;
    }
    const long long tiles128 = (long long)dcvic_cdiv(a->M, 128) * dcvic_cdiv(a->N, 128);
    if (tiles128 * a->batch >= num_cu) {
        dim3 grid((unsigned)tiles128, a->batch);
        bgemm_kernel<128><<<grid, 256, 0, (hipStream_t)stream>>>(*a);
    } else {                                              // small batch: 4x the workgroups
        dim3 grid((unsigned)(dcvic_cdiv(a->M, 64) * dcvic_cdiv(a->N, 64)), a->batch);
        bgemm_kernel<64><<<grid, 256, 0, (hipStream_t)stream>>>(*a);
    }
    DCVIC_CHECK_LAUNCH("bgemm");
    return DCVIC_OK;
}
