// conv_async.hip -- the generic implicit-GEMM convolution of conv.hip with the staging made asynchronous
// (LDS-DMA double buffering), for launches that cannot fill the chip with workgroups.
//
// The CHARM / hyperprior / Swin-side layers work on 16x16 ... 32x32 maps: a whole launch is a few hundred
// workgroups, i.e. about one per CU, so nothing hides a workgroup's global->LDS staging behind another
// workgroup's MFMAs and conv_mfma_kernel spends 70-80 % of its time waiting on loads.  Here the patch and the
// weight slabs of stage s+1 are written straight into the other LDS buffer by `global_load_lds` while stage s
// is computed (one barrier per stage; hipcc drains the DMA with vmcnt(0) in front of it).
// Tiling, packed-weight layout, stage partition (K.TG taps x K.CPS chunks), reduction order and epilogue are
// those of conv_mfma_kernel<MT,NT,WM,WN,false>: results are bit-identical, so the choice between the two is a
// pure scheduling decision (grid size) and never changes a value.
#include "conv_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

static __device__ float dcvic_zero_pad[16];   // zero-initialised: source of padded / out-of-range lanes

#define A_MAXSLOT 16

// LDS operand reads are inline asm with hand-counted waits: while an LDS-DMA is in flight hipcc turns every LDS wait
// into lgkmcnt(0), which would make "fetch item i+1, then compute item i" wait for item i+1 as well.
template <int OFF>
__device__ __forceinline__ float lds_read_f32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <int MT, int NT, int WM, int WN, bool H2>
__global__ __launch_bounds__(NTHREADS, 2) void conv_mfma_async_kernel(const ConvKArgs K, const int xs_floats, const int ws_floats) {
    constexpr int TC = WM * MT * 32;
    constexpr int P = WN * NT * 32;
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

    int pty[NT], ptx[NT], bbase[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int p = (wn * NT + nt) * 32 + lane_j;
        pty[nt] = p >> K.TWlog;
        ptx[nt] = p & (TW - 1);
        bbase[nt] = (pty[nt] * K.stride) * K.PW + ptx[nt] * K.stride + lane_k * K.plane;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    if (K.init) {
        const long long HWi = (long long)K.Hfull * K.Wfull;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int oy = oy0 + pty[nt], ox = ox0 + ptx[nt];
            if (oy >= K.Hout || ox >= K.Wout) continue;
            const long long pix = (long long)(oy * K.osy + K.ooy) * K.Wfull + (ox * K.osx + K.oox);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = cotile * TC + (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lane_k;
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
                    const float* gp = dcvic_zero_pad;
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
        // The stage is a sequence of ITEMS = (chunk cs, tap tt, 4-channel half h), two MFMA k-steps each, in the layer's
        // reduction order: (cs, tt, h) for most layers, (cs, h, tt) for the 3x3/stride-1 family (H2, see conv.hip).
        // With about one wave per SIMD nothing else hides the LDS latency, so the operands of the next item are fetched
        // while the MFMAs of the current one run: a two-deep register pipeline with hand-counted lgkmcnt, tap offsets
        // advanced incrementally in SGPRs (a handful of scalar instructions per tap, no divisions, no scalar loads --
        // SMEM shares lgkmcnt with LDS).
        const unsigned xs_addr = (unsigned)(uintptr_t)(lds_ptr_t)const_cast<float*>(Xs);
        const unsigned ws_addr = (unsigned)(uintptr_t)(lds_ptr_t)const_cast<float*>(Ws);
        const int tiy0 = tg / K.TX, tix0 = tg - tiy0 * K.TX;
        const int toff0 = (K.dy0 + tiy0 * K.dstep - K.dy_min) * K.PW + (K.dx0 + tix0 * K.dstep - K.dx_min);
        const int row_step = K.dstep * K.PW - K.TX * K.dstep;   // extra step from the last tap of a row to the next row
        const unsigned avec = 4u * (unsigned)(lane_k * TC + wm * (MT * 32) + lane_j);
        unsigned bvec[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bvec[nt] = 4u * (unsigned)bbase[nt];
        const unsigned plane8 = 8u * (unsigned)K.plane;          // two channels, in bytes

        // One wave issues in order and each MFMA waits for the previous one on the same accumulator (64 cycles), so the
        // loop is laid out so that everything else sits in those gaps: the LDS reads of the next item go out right after
        // an MFMA has been issued, the scalar tap bookkeeping and the next addresses after the last-but-one, and the only
        // wait -- lgkmcnt(0), at the top of an item -- finds its data already there.
        // fetch k-steps (2*hp, 2*hp+1) of the tap whose operand addresses are aaddr / baddr[]
        auto fetch2 = [&](float (&a)[2][MT], float (&bb)[2][NT], unsigned aaddr, const unsigned (&baddr)[NT]) {
            static_for<0, 2>([&](auto ks_) {
                constexpr int ks = decltype(ks_)::value;
                static_for<0, MT>([&](auto mt_) {
                    constexpr int mt = decltype(mt_)::value;
                    a[ks][mt] = lds_read_f32<4 * ((2 * ks) * TC + mt * 32)>(aaddr);
                });
            });
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bb[0][nt] = lds_read_f32<0>(baddr[nt]);
                bb[1][nt] = lds_read_f32<0>(baddr[nt] + plane8);
            }
        };
        auto mma1 = [&](const float (&a)[MT], const float (&bb)[NT]) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bb[nt], acc[mt][nt], 0, 0, 0);
        };
#define ASYNC_FENCE() __builtin_amdgcn_sched_barrier(0)
#define ASYNC_WAIT_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ASYNC_FENCE(); } while (0)
        for (int cs = 0; cs < ncs; ++cs) {
            const unsigned wcs = 4u * (unsigned)(cs * ntap * KC * TC), xcs = 4u * (unsigned)(cs * KC * K.plane);
            for (int po = 0; po < (H2 ? 2 : 1); ++po) {
                // tap state of the NEXT tap whose addresses are formed.  The prefetch runs one or two taps past the end of
                // the stage: those reads land in the slab / patch padding the host adds (never used), which keeps the
                // scalar bookkeeping to a short chain (a SALU chain of selects costs ~10 cycles per link on a lone wave).
                int tix_n = tix0;
                unsigned wtap = wcs + (H2 ? (unsigned)po * (16u * TC) : 0u);
                unsigned xtap = xcs + 4u * (unsigned)toff0 + (H2 ? (unsigned)po * (2u * plane8) : 0u);
                const unsigned col_step = 4u * (unsigned)K.dstep, rowcol_step = 4u * (unsigned)(K.dstep + row_step);
                auto advance = [&]() {
                    const bool wrap = tix_n + 1 == K.TX;
                    xtap += wrap ? rowcol_step : col_step;
                    tix_n = wrap ? 0 : tix_n + 1;
                    wtap += 4u * (KC * TC);
                };
                unsigned aaddr_n, baddr_n[NT];
                auto form = [&]() {
                    aaddr_n = ws_addr + wtap + avec;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) baddr_n[nt] = xs_addr + xtap + bvec[nt];
                };
                if (H2) {
                    // items = taps of half po, two k-steps each
                    float a0[2][MT], b0[2][NT], a1[2][MT], b1[2][NT];
                    form();
                    fetch2(a0, b0, aaddr_n, baddr_n);
                    advance(); form();
                    for (int t = 0; t < ntap; t += 2) {
                        ASYNC_WAIT_LDS();
                        mma1(a0[0], b0[0]);
                        ASYNC_FENCE();
                        fetch2(a1, b1, aaddr_n, baddr_n);          // tap t+1
                        ASYNC_FENCE();
                        mma1(a0[1], b0[1]);
                        ASYNC_FENCE();
                        advance(); form();                         // tap t+2
                        ASYNC_WAIT_LDS();
                        if (t + 1 < ntap) mma1(a1[0], b1[0]);
                        ASYNC_FENCE();
                        fetch2(a0, b0, aaddr_n, baddr_n);          // tap t+2
                        ASYNC_FENCE();
                        if (t + 1 < ntap) mma1(a1[1], b1[1]);
                        ASYNC_FENCE();
                        advance(); form();                         // tap t+3
                    }
                } else {
                    // items = taps, four k-steps each (h = 0: k-steps 0,1; h = 1: k-steps 2,3)
                    float a0[2][2][MT], b0[2][2][NT], a1[2][2][MT], b1[2][2][NT];
                    unsigned baddr_h[NT];
                    auto fetch_tap_h = [&](float (&a)[2][MT], float (&bb)[2][NT], int h) {
                        if (h == 0) fetch2(a, bb, aaddr_n, baddr_n);
                        else {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) baddr_h[nt] = baddr_n[nt] + 2u * plane8;
                            fetch2(a, bb, aaddr_n + 16u * TC, baddr_h);
                        }
                    };
                    form();
                    fetch_tap_h(a0[0], b0[0], 0);
                    fetch_tap_h(a0[1], b0[1], 1);
                    advance(); form();
                    for (int t = 0; t < ntap; t += 2) {
                        ASYNC_WAIT_LDS();
                        mma1(a0[0][0], b0[0][0]);
                        ASYNC_FENCE();
                        fetch_tap_h(a1[0], b1[0], 0);              // tap t+1, k-steps 0,1
                        ASYNC_FENCE();
                        mma1(a0[0][1], b0[0][1]);
                        ASYNC_FENCE();
                        fetch_tap_h(a1[1], b1[1], 1);              // tap t+1, k-steps 2,3
                        ASYNC_FENCE();
                        mma1(a0[1][0], b0[1][0]);
                        ASYNC_FENCE();
                        advance(); form();                         // tap t+2
                        ASYNC_FENCE();
                        mma1(a0[1][1], b0[1][1]);
                        ASYNC_WAIT_LDS();
                        if (t + 1 < ntap) mma1(a1[0][0], b1[0][0]);
                        ASYNC_FENCE();
                        fetch_tap_h(a0[0], b0[0], 0);              // tap t+2
                        ASYNC_FENCE();
                        if (t + 1 < ntap) mma1(a1[0][1], b1[0][1]);
                        ASYNC_FENCE();
                        fetch_tap_h(a0[1], b0[1], 1);
                        ASYNC_FENCE();
                        if (t + 1 < ntap) mma1(a1[1][0], b1[1][0]);
                        ASYNC_FENCE();
                        advance(); form();                         // tap t+3
                        ASYNC_FENCE();
                        if (t + 1 < ntap) mma1(a1[1][1], b1[1][1]);
                    }
                }
                ASYNC_WAIT_LDS();                                  // the extra fetch: hipcc does not count asm loads
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
        dcvic_static_for<0, NT>([&](auto nt_) {
            constexpr int nt = decltype(nt_)::value;
            const int oy = oy0 + pty[nt], ox = ox0 + ptx[nt];
            if (oy < K.Hout && ox < K.Wout) {
                const long long pix = (long long)(oy * K.osy + K.ooy) * K.Wfull + (ox * K.osx + K.oox);
                dcvic_static_for<0, MT>([&](auto mt_) {
                    constexpr int mt = decltype(mt_)::value;
                    const int cob = cotile * TC + (wm * MT + mt) * 32 + 4 * lane_k;
                    dcvic_conv_epilogue<16, (MT * NT >= 6 ? 4 : 8), RES, AFF>(K, n, acc[mt][nt], [cob](int r) { return cob + (r & 3) + 8 * (r >> 2); }, pix, HWo);
                });
            }
        });
    });
}

template <int MT, int NT, int WM, int WN>
static int launch_async(const ConvKArgs& K, int xs_floats, int ws_floats, hipStream_t st) {
    static std::atomic<unsigned> attr_mask{0};
    auto k1 = conv_mfma_async_kernel<MT, NT, WM, WN, false>;
    auto k2 = conv_mfma_async_kernel<MT, NT, WM, WN, true>;
    if (DcvicAttrOnce once_{attr_mask}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    const size_t lds = (size_t)2 * (xs_floats + ws_floats) * sizeof(float);
    if (K.halves == 2) k2<<<K.nblocks, NTHREADS, lds, st>>>(K, xs_floats, ws_floats);
    else k1<<<K.nblocks, NTHREADS, lds, st>>>(K, xs_floats, ws_floats);
    DCVIC_CHECK_LAUNCH("conv2d_async");
    return DCVIC_OK;
}

// returns DCVIC_OK after launching, 1 when the launch is not eligible (caller falls back to conv_mfma_kernel)
int dcvic_try_conv_async(const ConvKArgs& Kin, int cls, int P, hipStream_t st) {
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
        case 0 * 1000 + 256: return launch_async<2, 4, 2, 2>(K, xs_floats, ws_floats, st);
        case 0 * 1000 + 128: return launch_async<2, 2, 2, 2>(K, xs_floats, ws_floats, st);
        case 0 * 1000 + 64: return launch_async<2, 1, 2, 2>(K, xs_floats, ws_floats, st);
        case 1 * 1000 + 256: return launch_async<2, 2, 1, 4>(K, xs_floats, ws_floats, st);
        case 1 * 1000 + 128: return launch_async<1, 2, 2, 2>(K, xs_floats, ws_floats, st);
        case 1 * 1000 + 64: return launch_async<1, 1, 2, 2>(K, xs_floats, ws_floats, st);
        case 2 * 1000 + 256: return launch_async<1, 2, 1, 4>(K, xs_floats, ws_floats, st);
        case 2 * 1000 + 128: return launch_async<1, 1, 1, 4>(K, xs_floats, ws_floats, st);
        case 3 * 1000 + 256: return launch_async<3, 2, 1, 4>(K, xs_floats, ws_floats, st);
        case 3 * 1000 + 128: return launch_async<3, 1, 1, 4>(K, xs_floats, ws_floats, st);
        default: return 1;
    }
}
