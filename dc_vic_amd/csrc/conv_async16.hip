// conv_async16.hip -- the small-grid twin of conv.hip built on v_mfma_f32_16x16x4_f32.
//
// Same staging as conv_async.hip (LDS-DMA double buffering), but a wave's 32x32 output tile is held as FOUR 16x16
// accumulators and a 4-channel group is one MFMA (K = 4).  A lone wave per SIMD issues in order; with one 32x32x2 chain it
// stalls ~64 cycles on every MFMA (tools/lone_wave.hip: ~150 cycles per MFMA with its LDS reads and bookkeeping), with four
// independent 16x16x4 chains the matrix pipe stays fed while the wave issues the reads of the next item.
// Bit-identity: both MFMA shapes are exactly k-ordered fmaf chains on gfx950 (tools/mfma_order.hip: 0 mismatches against
// fmaf in 5 x 10^7 outputs, and against each other), and the channel order is unchanged (k ascending: channels 0-3 in one
// instruction, then 4-7), so results equal conv_mfma_kernel's bit for bit -- covered by test_conv_async_twin_bit_identical.
#include "conv_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ float dcvic_zero_pad16[16];   // zero-initialised: source of padded / out-of-range lanes

#define A_MAXSLOT 16

// LDS operand reads are inline asm with hand-counted waits: while an LDS-DMA is in flight hipcc turns every LDS wait
// into lgkmcnt(0), which would make "fetch item i+1, then compute item i" wait for item i+1 as well.
template <int OFF>
__device__ __forceinline__ float lds16_read_f32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void static16_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static16_for<I + 1, N>(f); }
}

// M16 x N16: the wave's tile in units of 16 output channels x 16 pixels; WM x WN waves per workgroup
template <int M16, int N16, int WM, int WN, bool H2>
__global__ __launch_bounds__(NTHREADS, 2) void conv_mfma_async16_kernel(const ConvKArgs K, const int xs_floats, const int ws_floats) {
    constexpr int TC = WM * M16 * 16;
    constexpr int P = WN * N16 * 16;
    constexpr int VPT = KC * TC / 4;                 // float4 per weight slab
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [2][xs_floats] patches, then [2][ws_floats] weight slabs
    float* const Xs0 = smem;
    float* const Ws0 = smem + 2 * xs_floats;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR: loops and LDS-DMA destinations stay scalar
    const int wm = wave / WN, wn = wave % WN;
    const int lane_k = lane >> 5, lane_j = lane & 31;

    int b;
    {
        const int orig = blockIdx.x, nb = K.nblocks;
        const int q = nb / NXCD, r = nb % NXCD, x = orig % NXCD;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + orig / NXCD;
    }
    const int cotile = b % K.n_cotiles; b /= K.n_cotiles;
    const int tile_x = b % K.tiles_x; b /= K.tiles_x;
    const int tile_y = b % K.tiles_y; b /= K.tiles_y;
    const int n = b;
    const int TW = 1 << K.TWlog;
    const int TH = P >> K.TWlog;
    const int oy0 = tile_y * TH, ox0 = tile_x * TW;
    const int iy0 = oy0 * K.stride + K.dy_min, ix0 = ox0 * K.stride + K.dx_min;

    // 16x16x4 operand layout: lane l supplies row/col (l & 15) of k = (l >> 4); a 16x16 accumulator holds rows
    // 4*(l >> 4) + r (r < 4) of column (l & 15)
    const int l16 = lane & 15, lq = lane >> 4;
    int pty[N16], ptx[N16], bbase[N16];
#pragma unroll
    for (int nt = 0; nt < N16; ++nt) {
        const int p = wn * (N16 * 16) + nt * 16 + l16;
        pty[nt] = p >> K.TWlog;
        ptx[nt] = p & (TW - 1);
        bbase[nt] = (pty[nt] * K.stride) * K.PW + ptx[nt] * K.stride + lq * K.plane;
    }

    f32x4 acc[M16][N16];
#pragma unroll
    for (int mt = 0; mt < M16; ++mt)
#pragma unroll
        for (int nt = 0; nt < N16; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;

    if (K.init) {
        const long long HWi = (long long)K.Hfull * K.Wfull;
#pragma unroll
        for (int nt = 0; nt < N16; ++nt) {
            const int oy = oy0 + pty[nt], ox = ox0 + ptx[nt];
            if (oy >= K.Hout || ox >= K.Wout) continue;
            const long long pix = (long long)(oy * K.osy + K.ooy) * K.Wfull + (ox * K.osx + K.oox);
#pragma unroll
            for (int mt = 0; mt < M16; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = cotile * TC + wm * (M16 * 16) + mt * 16 + 4 * lq + r;
                    if (co < K.Cout) acc[mt][nt][r] = K.init[(long long)n * K.init_bs + (long long)co * HWi + pix];
                }
        }
    }

    const long long HW = (long long)K.H * K.W;
    const float* wbase = K.wp + (long long)cotile * K.n_chunks * K.T * (KC * TC);

    // patch elements this thread moves for a stage of CPS chunks: element e = tid + s*256 -> (channel k, row, col);
    // the spatial part of the source offset is the same for every stage
    const int xslots = (K.CPS * KC * K.plane + NTHREADS - 1) / NTHREADS;
    int pk[A_MAXSLOT], poff[A_MAXSLOT];
#pragma unroll
    for (int s = 0; s < A_MAXSLOT; ++s) {
        pk[s] = 0; poff[s] = -1;
        if (s < xslots) {
            const int e = tid + s * NTHREADS;
            const int k = e / K.plane, r = e - k * K.plane;
            const int py = r / K.PW, px = r - py * K.PW;
            const int iy = iy0 + py, ix = ix0 + px;
            pk[s] = k;
            if (k < K.CPS * KC && iy >= 0 && iy < K.H && ix >= 0 && ix < K.W) poff[s] = iy * K.W + ix;
        }
    }

    // stage list: for chunk0 in steps of CPS: for tg in steps of TG   (same partition as conv_mfma_kernel)
    const int n_groups = (K.n_chunks + K.CPS - 1) / K.CPS;
    const int n_tgs = (K.T + K.TG - 1) / K.TG;
    const int n_stages = n_groups * n_tgs;

    // issue() runs on the lone wave's critical path (measured: 2.5k of 7.9k cycles per stage before this was trimmed), so
    // it keeps running (group, tap-group) counters instead of dividing, per-source base pointers are formed once, and
    // the weight slab is a plain linear copy
    const float* nsrc[DCVIC_MAX_SRC];
#pragma unroll
    for (int i = 0; i < DCVIC_MAX_SRC; ++i) nsrc[i] = K.src[i] + (long long)n * K.src_bs[i];
    const int c_s1 = K.srcC[0], c_s2 = K.srcC[0] + K.srcC[1];      // first channel of source 1 / 2 (2^30 when absent)
    int i_grp = 0, i_tgi = 0;                                       // coordinates of the NEXT stage to issue
    auto issue = [&]() {
        const int grp = i_grp, tgi = i_tgi;
        const int chunk0 = grp * K.CPS, tg = tgi * K.TG;
        const int ncs = min(K.CPS, K.n_chunks - chunk0);
        const int stage_par = (grp * n_tgs + tgi) & 1;
        if (tgi == 0) {
            float* xb = Xs0 + (grp & 1) * xs_floats;
            const int c0 = chunk0 * KC;
            const int cmax = min(K.Cin, c0 + ncs * KC);
#pragma unroll
            for (int s = 0; s < A_MAXSLOT; ++s) {
                if (s < xslots) {
                    const int c = c0 + pk[s];
                    const float* gp = dcvic_zero_pad16;
                    if (poff[s] >= 0 && c < cmax) {
                        const float* base = c < c_s1 ? nsrc[0] + (long long)c * HW
                                                     : (c < c_s2 ? nsrc[1] + (long long)(c - c_s1) * HW : nsrc[2] + (long long)(c - c_s2) * HW);
                        gp = base + poff[s];
                    }
                    __builtin_amdgcn_global_load_lds(gp, (lds_ptr_t)(xb + wave * 64 + s * NTHREADS), 4, 0, 0);
                }
            }
        }
        const int ntap = min(K.TG, K.T - tg);
        const int nslab = (K.TG >= K.T) ? ncs * K.T : ntap;
        const int total = nslab * VPT;                             // float4, a multiple of 64
        const float4* wsrc = reinterpret_cast<const float4*>(wbase + ((long long)chunk0 * K.T + tg) * (KC * TC)) + tid;
        float* wb = Ws0 + stage_par * ws_floats + wave * 256;
        for (int i0 = wave * 64; i0 < total; i0 += NTHREADS) {
            __builtin_amdgcn_global_load_lds(wsrc, (lds_ptr_t)wb, 16, 0, 0);
            wsrc += NTHREADS; wb += NTHREADS * 4;
        }
        if (++i_tgi == n_tgs) { i_tgi = 0; ++i_grp; }
    };

    issue();
    __syncthreads();

    int c_grp = 0, c_tgi = 0;                                       // coordinates of the stage being computed
    for (int stage = 0; stage < n_stages; ++stage) {
        if (stage + 1 < n_stages) issue();
        const int grp = c_grp, tgi = c_tgi;
        if (++c_tgi == n_tgs) { c_tgi = 0; ++c_grp; }
        const int chunk0 = grp * K.CPS, tg = tgi * K.TG;
        const int ncs = min(K.CPS, K.n_chunks - chunk0);
        const int ntap = min(K.TG, K.T - tg);
        const float* Xs = Xs0 + (grp & 1) * xs_floats;
        const float* Ws = Ws0 + (stage & 1) * ws_floats;
        // The stage is a sequence of ITEMS = (chunk cs, tap tt, 4-channel group h), one K = 4 MFMA per 16x16 accumulator,
        // in the layer's reduction order: (cs, tt, h) for most layers, (cs, h, tt) for the 3x3/stride-1 family (H2).
        // Two register buffers: the reads of the next item are issued, then the M16 x N16 independent MFMAs of the current
        // one; the lgkmcnt(0) at the top of an item finds its data there.  Tap offsets advance in SGPRs (no selects chains,
        // no divisions, no scalar loads -- SMEM shares lgkmcnt with LDS).
        const unsigned xs_addr = (unsigned)(uintptr_t)(lds_ptr_t)const_cast<float*>(Xs);
        const unsigned ws_addr = (unsigned)(uintptr_t)(lds_ptr_t)const_cast<float*>(Ws);
        const int tiy0 = tg / K.TX, tix0 = tg - tiy0 * K.TX;
        const int toff0 = (K.dy0 + tiy0 * K.dstep - K.dy_min) * K.PW + (K.dx0 + tix0 * K.dstep - K.dx_min);
        const int row_step = K.dstep * K.PW - K.TX * K.dstep;   // extra step from the last tap of a row to the next row
        const unsigned avec = 4u * (unsigned)(lq * TC + wm * (M16 * 16) + l16);
        unsigned bvec[N16];
#pragma unroll
        for (int nt = 0; nt < N16; ++nt) bvec[nt] = 4u * (unsigned)bbase[nt];
        const unsigned plane16 = 16u * (unsigned)K.plane;        // four channels, in bytes

#define ASYNC_FENCE() __builtin_amdgcn_sched_barrier(0)
#define ASYNC_WAIT_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ASYNC_FENCE(); } while (0)
        auto fetch = [&](float (&a)[M16], float (&bb)[N16], unsigned wtap, unsigned xtap, unsigned h) {
            const unsigned aaddr = ws_addr + wtap + h * (16u * TC) + avec;
            static16_for<0, M16>([&](auto mt_) {
                constexpr int mt = decltype(mt_)::value;
                a[mt] = lds16_read_f32<64 * mt>(aaddr);
            });
            const unsigned xb = xs_addr + xtap + h * plane16;
#pragma unroll
            for (int nt = 0; nt < N16; ++nt) bb[nt] = lds16_read_f32<0>(xb + bvec[nt]);
        };
        auto mma = [&](const float (&a)[M16], const float (&bb)[N16]) {
#pragma unroll
            for (int mt = 0; mt < M16; ++mt)
#pragma unroll
                for (int nt = 0; nt < N16; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], bb[nt], acc[mt][nt], 0, 0, 0);
        };
        // one item: the Q = M16 x N16 independent MFMAs of the current operands, with the R = M16 + N16 LDS reads of the
        // next item's operands spread between them -- a lone wave issues in order, so reads placed after the MFMAs would
        // only start once the last MFMA has left the issue stage, and MFMAs placed after the reads would find the pipe idle
        auto step = [&](const float (&ca)[M16], const float (&cb)[N16], float (&na)[M16], float (&nb)[N16],
                        unsigned wtap, unsigned xtap, unsigned h) {
            constexpr int Q = M16 * N16, R = M16 + N16;
            const unsigned aaddr = ws_addr + wtap + h * (16u * TC) + avec;
            unsigned baddr[N16];
            const unsigned xb = xs_addr + xtap + h * plane16;
#pragma unroll
            for (int nt = 0; nt < N16; ++nt) baddr[nt] = xb + bvec[nt];
            ASYNC_FENCE();
            static16_for<0, Q>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                constexpr int mt = q / N16, nt = q % N16;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[mt], cb[nt], acc[mt][nt], 0, 0, 0);
                ASYNC_FENCE();
                static16_for<0, R>([&](auto r_) {
                    constexpr int r = decltype(r_)::value;
                    if constexpr (r * Q / R == q) {
                        if constexpr (r < M16) na[r] = lds16_read_f32<64 * r>(aaddr);
                        else nb[r - M16] = lds16_read_f32<0>(baddr[r - M16]);
                    }
                });
                ASYNC_FENCE();
            });
        };
        float a0[M16], b0[N16], a1[M16], b1[N16];
        for (int cs = 0; cs < ncs; ++cs) {
            const unsigned wcs = 4u * (unsigned)(cs * ntap * KC * TC), xcs = 4u * (unsigned)(cs * KC * K.plane);
            for (int po = 0; po < (H2 ? 2 : 1); ++po) {
                // tap state of the NEXT tap to fetch.  The prefetch runs one or two taps past the end of the stage: those
                // reads land in the slab / patch padding the host adds (never used)
                int tix_n = tix0;
                unsigned wtap = wcs, xtap = xcs + 4u * (unsigned)toff0;
                const unsigned col_step = 4u * (unsigned)K.dstep, rowcol_step = 4u * (unsigned)(K.dstep + row_step);
                auto advance = [&]() {
                    const bool wrap = tix_n + 1 == K.TX;
                    xtap += wrap ? rowcol_step : col_step;
                    tix_n = wrap ? 0 : tix_n + 1;
                    wtap += 4u * (KC * TC);
                };
                if (H2) {
                    // items = taps of channel group po; pairs of taps ping-pong between the two register sets
                    fetch(a0, b0, wtap, xtap, (unsigned)po);
                    int t = 0;
                    for (; t + 1 < ntap; t += 2) {
                        advance();
                        ASYNC_WAIT_LDS();
                        step(a0, b0, a1, b1, wtap, xtap, (unsigned)po);      // computes tap t, fetches tap t+1
                        advance();
                        ASYNC_WAIT_LDS();
                        step(a1, b1, a0, b0, wtap, xtap, (unsigned)po);      // computes tap t+1, fetches tap t+2
                    }
                    ASYNC_WAIT_LDS();
                    if (t < ntap) mma(a0, b0);                               // odd tap count: the last one
                } else {
                    // items = (tap, group 0), (tap, group 1)
                    fetch(a0, b0, wtap, xtap, 0u);
                    for (int t = 0; t < ntap; ++t) {
                        ASYNC_WAIT_LDS();
                        step(a0, b0, a1, b1, wtap, xtap, 1u);                // computes (t, 0), fetches (t, 1)
                        advance();
                        ASYNC_WAIT_LDS();
                        step(a1, b1, a0, b0, wtap, xtap, 0u);                // computes (t, 1), fetches (t+1, 0)
                    }
                    ASYNC_WAIT_LDS();                                        // the extra fetch: hipcc does not count asm loads
                }
            }
        }
#undef ASYNC_FENCE
#undef ASYNC_WAIT_LDS
        __syncthreads();
    }

    // ---- epilogue: bias -> act -> (+res) -> (affine) -> store  (same order as conv.hip)
    const long long HWo = (long long)K.Hfull * K.Wfull;
    dcvic_epilogue_dispatch(K, [&](auto res_, auto aff_) {
        constexpr bool RES = decltype(res_)::value, AFF = decltype(aff_)::value;
        dcvic_static_for<0, N16>([&](auto nt_) {
            constexpr int nt = decltype(nt_)::value;
            const int oy = oy0 + pty[nt], ox = ox0 + ptx[nt];
            if (oy < K.Hout && ox < K.Wout) {
                const long long pix = (long long)(oy * K.osy + K.ooy) * K.Wfull + (ox * K.osx + K.oox);
                // the M16 groups of this pixel as one batch of 4 * M16 values
                float av[4 * M16];
#pragma unroll
                for (int mt = 0; mt < M16; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) av[mt * 4 + r] = acc[mt][nt][r];
                const int cob = cotile * TC + wm * (M16 * 16) + 4 * lq;
                dcvic_conv_epilogue<4 * M16, (4 * M16 < 8 ? 4 * M16 : 8), RES, AFF>(K, n, av, [cob](int q) { return cob + (q >> 2) * 16 + (q & 3); }, pix, HWo);
            }
        });
    });
}

template <int M16, int N16, int WM, int WN>
static int launch_async16(const ConvKArgs& K, int xs_floats, int ws_floats, hipStream_t st) {
    static std::atomic<unsigned> attr_mask{0};
    auto k1 = conv_mfma_async16_kernel<M16, N16, WM, WN, false>;
    auto k2 = conv_mfma_async16_kernel<M16, N16, WM, WN, true>;
    if (DcvicAttrOnce once_{attr_mask}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const size_t lds = (size_t)2 * (xs_floats + ws_floats) * sizeof(float);
    if (K.halves == 2) k2<<<K.nblocks, NTHREADS, lds, st>>>(K, xs_floats, ws_floats);
    else k1<<<K.nblocks, NTHREADS, lds, st>>>(K, xs_floats, ws_floats);
    DCVIC_CHECK_LAUNCH("conv2d_async16");
    return DCVIC_OK;
}

// returns DCVIC_OK after launching, 1 when the launch is not eligible (caller falls back to conv_mfma_kernel)
int dcvic_try_conv_async16(const ConvKArgs& Kin, int cls, int P, hipStream_t st) {
    static const int TCs[4] = {128, 64, 32, 96};
    static const int stage_kb = getenv("DCVIC_ASYNC_STAGE_KB") ? atoi(getenv("DCVIC_ASYNC_STAGE_KB")) : 40;
    const int TC = TCs[cls];
    ConvKArgs K = Kin;
    // Own stage partition (the reduction order does not depend on it): small stages keep a workgroup near 50 KiB of LDS
    // so that two or three of them -- e.g. the two CHARM parameter networks launched on two streams -- share a CU and
    // every SIMD has more than one independent MFMA chain to interleave.  The 3x3 family keeps whole taps per stage.
    int TG = (stage_kb * 1024) / (KC * TC * 4);
    if (TG < 1) TG = 1;
    if (K.halves == 2 && TG < K.T) TG = K.T;
    if (TG > K.T) TG = K.T;
    int CPS = 1;
    if (TG == K.T) {
        const int per_chunk = (KC * K.plane + K.T * KC * TC) * 4;
        while (CPS < 8 && (CPS + 1) * per_chunk <= stage_kb * 1024 + 8 * 1024 && (CPS + 1) * KC * K.plane <= A_MAXSLOT * NTHREADS) ++CPS;
        if (CPS > K.n_chunks) CPS = K.n_chunks;
    }
    K.TG = TG; K.CPS = CPS;
    const int stage_elems = K.CPS * KC * K.plane;
    const int xslots = (stage_elems + NTHREADS - 1) / NTHREADS;
    if (xslots > A_MAXSLOT) return 1;
    if ((long long)K.H * K.W * 1 >= (1ll << 30)) return 1;
    // + padding for the compute loop's prefetch, which runs up to two taps past the end of a stage (never used)
    const int xs_floats = xslots * NTHREADS + 2 * (K.PW + 2);
    const int slabs = (K.TG >= K.T) ? K.CPS * K.T : K.TG;
    const int ws_floats = (slabs + 2) * KC * TC;
    if ((size_t)2 * (xs_floats + ws_floats) * sizeof(float) > 156 * 1024) return 1;
    switch (cls * 1000 + P) {
        case 0 * 1000 + 64: return launch_async16<4, 2, 2, 2>(K, xs_floats, ws_floats, st);    // 128 ch x 64 px
        case 1 * 1000 + 128: return launch_async16<2, 4, 2, 2>(K, xs_floats, ws_floats, st);   //  64 ch x 128 px
        case 1 * 1000 + 64: return launch_async16<2, 2, 2, 2>(K, xs_floats, ws_floats, st);    //  64 ch x 64 px
        case 2 * 1000 + 256: return launch_async16<2, 4, 1, 4>(K, xs_floats, ws_floats, st);   //  32 ch x 256 px
        case 2 * 1000 + 128: return launch_async16<2, 2, 1, 4>(K, xs_floats, ws_floats, st);   //  32 ch x 128 px
        // tiles below the 32x32-per-wave grain, for launches that would leave CUs idle (N = 1 .. 4, 16x16 maps)
        case 2 * 1000 + 64: return launch_async16<2, 1, 1, 4>(K, xs_floats, ws_floats, st);    //  32 ch x 64 px
        case 2 * 1000 + 32: return launch_async16<1, 1, 2, 2>(K, xs_floats, ws_floats, st);    //  32 ch x 32 px
        default: return 1;
    }
}
