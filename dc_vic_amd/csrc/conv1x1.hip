// conv1x1.hip -- 1x1 / stride-1 convolutions (attention q/k/v/proj, nin_shortcut, ELIC bottleneck ends, Swin
// MLP-side projections: ~10 % of the path's time) as a DMA-pipelined GEMM  Out[co][p] = sum_ci W[co][ci] X[ci][p].
//
// A 1x1 convolution needs no spatial tiling: a workgroup takes 256 CONSECUTIVE pixels of the flattened H*W plane, so
// every input-channel row of its tile is one contiguous 1 KiB segment -- exactly one `global_load_lds_dwordx4` per
// wave (64 lanes x 16 B) straight into LDS, no VGPR staging, no per-element address arithmetic.  Stages of 16 input
// channels (16 KiB of pixels + 16 x TC weights) are double buffered; 512 threads = 8 waves; one barrier per stage.
// Same packed-weight layout ([cotile][chunk][tap = 1][8][TC]), reduction order (channel ascending, in MFMA k-pairs)
// and epilogue as conv.hip: results are bit-identical to the generic kernel.
#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

static __device__ __attribute__((aligned(16))) float dcvic_zero_row[4];   // source of out-of-range 16-B groups

#define Q_P 256          /* pixels per workgroup */
#define Q_CH 16          /* input channels per stage */
#define Q_THREADS 512

// TCv = 128: waves 2 x 4, each 64 ch x 64 px (MT = NT = 2);  96 / 64: waves 1 x 8, each TCv ch x 32 px (MT = 3 / 2, NT = 1)
template <int TCv>
__global__ __launch_bounds__(Q_THREADS, 4) void conv1x1_dma_kernel(const ConvKArgs K) {
    constexpr int WM = (TCv == 128) ? 2 : 1;
    constexpr int WN = 8 / WM;
    constexpr int MT = TCv / (32 * WM);
    constexpr int NT = Q_P / (32 * WN);
    constexpr int XS = Q_CH * Q_P;                 // floats
    constexpr int WS = Q_CH * TCv;
    constexpr int BUF = XS + WS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR: LDS-DMA destinations stay scalar
    const int wm = wave / WN, wn = wave % WN;
    const int lane_k = lane >> 5, lane_j = lane & 31;

    int b;
    {
        const int orig = blockIdx.x, nb = K.nblocks;
        const int q = nb / NXCD, r = nb % NXCD, x = orig % NXCD;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + orig / NXCD;
    }
    const int cotile = b % K.n_cotiles; b /= K.n_cotiles;
    const int ptile = b % K.tiles_x; b /= K.tiles_x;
    const int n = b;
    const int HW = K.H * K.W;
    const int p0 = ptile * Q_P;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* wbase = K.wp + (long long)cotile * K.n_chunks * (long long)(KC * TCv);
    const int n_stages = (K.n_chunks * KC + Q_CH - 1) / Q_CH;
    // this lane's 16-B group of a channel row; groups past the end of the plane read zeros (HW % 4 == 0)
    const int pl = p0 + lane * 4;
    const bool pin = pl < HW;

    // Running pointers (same idea as conv3x3.hip): this wave's two pixel rows of a stage and this thread's weight pieces
    // advance by fixed strides; only a change of source in the virtual concat re-derives them.
    const float* xrow[Q_CH / 8];
    int xleft[Q_CH / 8];                                          // channels left in the current source, per row slot
    long long xstep[Q_CH / 8];                                    // elements per stage (0 while the slot reads the zero row)
    auto x_rebase = [&](int j, int c) {                          // row slot j -> absolute channel c
        xstep[j] = (long long)Q_CH * HW;
        if (c >= K.Cin || !pin) { xrow[j] = dcvic_zero_row; xleft[j] = 1 << 30; xstep[j] = 0; return; }
        int si = 0, cl = c;
        if (cl >= K.srcC[0]) { cl -= K.srcC[0]; si = 1; if (cl >= K.srcC[1]) { cl -= K.srcC[1]; si = 2; } }
        xrow[j] = K.src[si] + (long long)n * K.src_bs[si] + (long long)cl * HW + pl;
        xleft[j] = min(K.srcC[si] - cl, K.Cin - c);
    };
#pragma unroll
    for (int j = 0; j < Q_CH / 8; ++j) x_rebase(j, wave + 8 * j);
    constexpr int NV = WS / 4;                                    // float4 of weights per stage
    constexpr int NWJ = (NV + Q_THREADS - 1) / Q_THREADS;
    const float* wrow[NWJ];                                       // (float4 rows kept as float*: see conv3x3.hip)
#pragma unroll
    for (int j = 0; j < NWJ; ++j) wrow[j] = wbase + 4 * (j * Q_THREADS + tid);

    auto issue = [&](int stage, int buf) {
        float* xb = smem + buf * BUF;
        float* wb = xb + XS;
#pragma unroll
        for (int j = 0; j < Q_CH / 8; ++j) {
            __builtin_amdgcn_global_load_lds(xrow[j], (lds_ptr_t)(xb + (wave + 8 * j) * Q_P), 16, 0, 0);
            xleft[j] -= Q_CH;
            if (xleft[j] > 0) xrow[j] += xstep[j];
            else x_rebase(j, (stage + 1) * Q_CH + wave + 8 * j);
        }
        // weight rows: Q_CH x TCv floats, contiguous in the pack (two chunks); a missing last chunk reads zeros
        const int nfl = min(Q_CH / KC, K.n_chunks - stage * (Q_CH / KC)) * (KC * TCv);        // floats available
#pragma unroll
        for (int j = 0; j < NWJ; ++j) {
            const int v0 = j * Q_THREADS + wave * 64;                                // wave-uniform
            if (v0 < NV) {
                const float* gp = (v0 * 4 < nfl) ? wrow[j] : dcvic_zero_row;
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float4*>(gp), (lds_ptr_t)(wb + v0 * 4), 16, 0, 0);
            }
            wrow[j] += WS;
        }
    };

    // per-channel epilogue vectors of this workgroup go to LDS (bias, and the beta-FT scale / shift of image n): they are
    // read in the epilogue without touching vmcnt
    float* const sbias = smem + 2 * BUF;
    if (tid < TCv) {
        const int c = min(cotile * TCv + tid, K.Cout - 1);
        sbias[tid] = K.bias ? K.bias[c] : 0.f;
        sbias[TCv + tid] = K.affs ? K.affs[(long long)n * K.aff_bs + c] : 0.f;
        sbias[2 * TCv + tid] = K.afft ? K.afft[(long long)n * K.aff_bs + c] : 0.f;
    }
    issue(0, 0);
    __syncthreads();

    const int alane = lane_k * TCv + wm * (MT * 32) + lane_j;
    const int xlane = lane_k * Q_P + wn * (NT * 32) + lane_j;
    for (int stage = 0; stage < n_stages; ++stage) {
        const int buf = stage & 1;
        if (stage + 1 < n_stages) issue(stage + 1, buf ^ 1);
        const float* xb = smem + buf * BUF + xlane;
        const float* wb = smem + buf * BUF + XS + alane;
        float a_cur[MT], b_cur[NT], a_nxt[MT], b_nxt[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a_cur[mt] = wb[mt * 32];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_cur[nt] = xb[nt * 32];
#pragma unroll
        for (int ks = 0; ks < Q_CH / 2; ++ks) {
            if (ks + 1 < Q_CH / 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = wb[(2 * (ks + 1)) * TCv + mt * 32];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = xb[(2 * (ks + 1)) * Q_P + nt * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt], b_cur[nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
            for (int i = 0; i < NT; ++i) b_cur[i] = b_nxt[i];
        }
        __syncthreads();
    }

    // ---- epilogue: bias -> act -> (+res) -> (affine) -> store  (same order as conv.hip); output plane is flat
    // Fast form for full tiles (see conv3x3.hip): per-channel vectors from LDS, residuals requested one 8-value sub-group
    // ahead of the stores, one 32-bit byte offset shared by residual and output.
    if (p0 + Q_P <= HW && (cotile + 1) * TCv <= K.Cout && (long long)K.Cout * HW * 4 < (1ll << 32)) {
        const char* const rb = reinterpret_cast<const char*>(K.res ? K.res + (long long)n * K.res_bs : nullptr);
        char* const ob = reinterpret_cast<char*>(K.out + (long long)n * K.out_bs);
        const bool has_res = K.res != nullptr, has_bias = K.bias != nullptr, has_aff = K.affs != nullptr;
        const int act = K.act;
        constexpr int NG = MT * NT, NS = 2 * NG, SB = 8;
        unsigned goff[NG];
        dcvic_static_for<0, NG>([&](auto g_) {
            constexpr int g = decltype(g_)::value, mt = g % MT, nt = g / MT;
            const int pix = p0 + wn * (NT * 32) + nt * 32 + lane_j;
            goff[g] = (unsigned)(4 * ((long long)(cotile * TCv + (wm * MT + mt) * 32 + 4 * lane_k) * HW + pix));
        });
        const unsigned rstep = (unsigned)(4 * HW);
        float rv[2][SB];
        auto loads = [&](auto s_, float (&dst)[SB]) {
            constexpr int sg = decltype(s_)::value, g = sg / 2, r0 = (sg & 1) * SB;
            dcvic_static_for<0, SB>([&](auto r_) {
                constexpr int r = r0 + decltype(r_)::value;
                dst[r - r0] = *reinterpret_cast<const float*>(rb + (goff[g] + (unsigned)((r & 3) + 8 * (r >> 2)) * rstep));
            });
        };
        if (has_res) loads(std::integral_constant<int, 0>{}, rv[0]);
        dcvic_static_for<0, NS>([&](auto s_) {
            constexpr int sg = decltype(s_)::value, g = sg / 2, r0 = (sg & 1) * SB, mt = g % MT;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (sg + 1 < NS) { if (has_res) loads(std::integral_constant<int, sg + 1>{}, rv[(sg + 1) & 1]); }
            __builtin_amdgcn_sched_barrier(0);
            float v[SB];
            dcvic_static_for<0, SB>([&](auto r_) {
                constexpr int r = r0 + decltype(r_)::value;
                const int cl = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lane_k;
                float e = acc[mt][g / MT][r];
                if (has_bias) e += sbias[cl];
                e = dcvic_act(e, act);
                if (has_res) e += rv[sg & 1][r - r0];
                if (has_aff) e = e * (1.f + sbias[TCv + cl]) + sbias[2 * TCv + cl];
                v[r - r0] = e;
            });
            __builtin_amdgcn_sched_barrier(0);
            dcvic_static_for<0, SB>([&](auto r_) {
                constexpr int r = r0 + decltype(r_)::value;
                *reinterpret_cast<float*>(ob + (goff[g] + (unsigned)((r & 3) + 8 * (r >> 2)) * rstep)) = v[r - r0];
            });
        });
        return;
    }
    dcvic_epilogue_dispatch(K, [&](auto res_, auto aff_) {
        constexpr bool RES = decltype(res_)::value, AFF = decltype(aff_)::value;
        dcvic_static_for<0, NT>([&](auto nt_) {
            constexpr int nt = decltype(nt_)::value;
            const int pix = p0 + wn * (NT * 32) + nt * 32 + lane_j;
            if (pix < HW) {
                dcvic_static_for<0, MT>([&](auto mt_) {
                    constexpr int mt = decltype(mt_)::value;
                    const int cob = cotile * TCv + (wm * MT + mt) * 32 + 4 * lane_k;
                    dcvic_conv_epilogue<16, (AFF ? 4 : 8), RES, AFF>(K, n, acc[mt][nt], [cob](int r) { return cob + (r & 3) + 8 * (r >> 2); }, (long long)pix, (long long)HW);
                });
            }
        });
    });
}

template <int TCv>
static int launch_1x1(const ConvKArgs& A, hipStream_t st) {
    static std::atomic<unsigned> attr_mask{0};
    auto k = conv1x1_dma_kernel<TCv>;
    if (DcvicAttrOnce once_{attr_mask}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const size_t lds = (size_t)(2 * (Q_CH * Q_P + Q_CH * TCv) + 3 * TCv) * sizeof(float);   // + bias / scale / shift rows
    k<<<A.nblocks, Q_THREADS, lds, st>>>(A);
    DCVIC_CHECK_LAUNCH("conv1x1_dma");
    return DCVIC_OK;
}

// returns DCVIC_OK after launching, 1 when the launch is not eligible (caller falls back to conv_mfma_kernel)
int dcvic_try_conv1x1_dma(const ConvKArgs& K, int n_src, bool upsample, int cls, int min_blocks, hipStream_t st) {
    if (K.T != 1 || K.stride != 1 || upsample || K.init || K.dy0 != 0 || K.dx0 != 0) return 1;
    if (K.osy != 1 || K.osx != 1 || K.ooy != 0 || K.oox != 0) return 1;
    if (K.Hout != K.H || K.Wout != K.W || K.Hfull != K.H || K.Wfull != K.W) return 1;
    const long long HW = (long long)K.H * K.W;
    if ((HW & 3) || HW >= (1ll << 30)) return 1;
    for (int i = 0; i < n_src; ++i)
        if ((K.src_bs[i] & 3) || (reinterpret_cast<uintptr_t>(K.src[i]) & 15)) return 1;
    if (cls != 0 && cls != 1 && cls != 3) return 1;
    ConvKArgs A = K;
    A.tiles_x = (int)((HW + Q_P - 1) / Q_P);
    A.tiles_y = 1;
    const long long blocks = (long long)K.N * A.tiles_x * K.n_cotiles;
    if (blocks >= (1ll << 31) || blocks < min_blocks) return 1;
    A.nblocks = (int)blocks;
    switch (cls) {
        case 0: return launch_1x1<128>(A, st);
        case 1: return launch_1x1<64>(A, st);
        default: return launch_1x1<96>(A, st);
    }
}
