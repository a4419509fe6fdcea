// conv.hip -- implicit-GEMM convolution family on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// One kernel template covers every Conv2d / ConvTranspose2d phase / upsample+conv on the DC-VIC path
// (see include/dcvic.h for the reference operators it replaces).
//
// GEMM view:   D[co][pixel] = sum_{ci,tap} Wt[tap][ci][co] * X[ci][pixel + tap]
//   A operand = weights (rows = output channels), B operand = input pixels (cols), so that the
//   accumulator's lane axis is the pixel axis and NCHW stores are 128-B row segments.
// Data movement per workgroup (256 threads = 4 waves) and stage of CPS x 8 input channels:
//   * the input PATCH (tile + halo, all taps) of the channels is staged once into LDS, zero padded
//     (coalesced reads of NCHW rows, 8 independent loads in flight per lane), and re-used by every
//     tap -> each input element is read from HBM/L2 once per output-channel tile, not once per tap;
//   * the pre-packed weight slabs [tap][8][TC] are linear float4 copies into LDS;
//   * every wave then issues MT x NT MFMA 32x32x2 per (tap, channel pair) with operands fetched by
//     conflict-free ds_read_b32 (consecutive lanes -> consecutive words).
// Tile variants (TC output channels x P pixels per workgroup) are picked per launch so that small
// feature maps (16x16 CHARM / hyperprior maps, N=1 decoding) still fill the 256 CUs; 1x1 convolutions
// stage several channel chunks per barrier pair.  Blocks are remapped so that the blocks sharing an
// XCD (b % 8) walk contiguous tiles: co-tiles of one patch hit the same L2.
// The fp32 MFMA is a k-ordered fmaf chain, so the reduction order of an output element is
// (chunk asc, tap asc, channel asc) for EVERY variant, grid and batch size: results are deterministic,
// batch-invariant and identical across variants (needed for encoder/decoder agreement).
#include "conv_common.h"

template <int MT, int NT, int WM, int WN, bool UPS>
__global__ __launch_bounds__(NTHREADS, (MT * NT >= 8) ? 2 : ((MT * NT >= 4) ? 3 : 4)) void conv_mfma_kernel(const ConvKArgs K) {
    constexpr int TC = WM * MT * 32;
    constexpr int P = WN * NT * 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                                            // [CPS*KC][plane]
    float* Ws = smem + ((K.CPS * KC * K.plane + 3) & ~3);        // [slabs][KC][TC]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lane_k = lane >> 5, lane_j = lane & 31;

    // XCD-aware remap (bijective): blocks b and b+8 share an XCD -> give each XCD a contiguous range
    int b;
    {
        const int orig = blockIdx.x, nb = K.nblocks;
        const int q = nb / NXCD, r = nb % NXCD, x = orig % NXCD;
        b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + orig / NXCD;
    }
    // block -> (n, tile_y, tile_x, cotile); cotile fastest: co-tiles of one patch run back to back
    const int cotile = b % K.n_cotiles; b /= K.n_cotiles;
    const int tile_x = b % K.tiles_x; b /= K.tiles_x;
    const int tile_y = b % K.tiles_y; b /= K.tiles_y;
    const int n = b;
    const int TW = 1 << K.TWlog;
    const int TH = P >> K.TWlog;
    const int oy0 = tile_y * TH, ox0 = tile_x * TW;

    // patch origin in input coordinates
    int iy0, ix0;
    if (UPS) { iy0 = (oy0 + K.dy_min) >> 1; ix0 = (ox0 + K.dx_min) >> 1; }
    else { iy0 = oy0 * K.stride + K.dy_min; ix0 = ox0 * K.stride + K.dx_min; }

    // per-lane pixel coordinates / LDS base offsets of each B fragment
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
        // continue a reduction started by another launch (CHARM: the hyperprior part of a transform's first conv is
        // computed for all slices up front); the fma chain then runs on exactly as in a single launch
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

    for (int chunk0 = 0; chunk0 < K.n_chunks; chunk0 += K.CPS) {
        const int ncs = min(K.CPS, K.n_chunks - chunk0);
        // ---- stage the input patch of channels [chunk0*8, (chunk0+ncs)*8): branch-free, 8 loads in flight
        for (int cs = 0; cs < ncs; ++cs) {
            const int c0 = (chunk0 + cs) * KC;
            const float* cp[KC];
            bool cok[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                int c = c0 + k;
                cok[k] = c < K.Cin;
                int si = 0;
                if (cok[k]) {
                    if (c >= K.srcC[0]) { c -= K.srcC[0]; si = 1; if (c >= K.srcC[1]) { c -= K.srcC[1]; si = 2; } }
                } else {
                    c = 0;
                }
                cp[k] = K.src[si] + (long long)n * K.src_bs[si] + (long long)c * HW;
            }
            float* xdst = Xs + cs * (KC * K.plane);
            for (int s = 0; s < K.nslots; ++s) {
                const int r = min(tid + s * NTHREADS, K.plane - 1);   // surplus lanes duplicate the last element (same value)
                const int py = r / K.PW, px = r - py * K.PW;
                const int iy = iy0 + py, ix = ix0 + px;
                const bool inb = (iy >= 0) & (iy < K.H) & (ix >= 0) & (ix < K.W);
                const int g = inb ? iy * K.W + ix : 0;
                float v[KC];
#pragma unroll
                for (int k = 0; k < KC; ++k) v[k] = cp[k][g];
#pragma unroll
                for (int k = 0; k < KC; ++k) xdst[k * K.plane + r] = (inb && cok[k]) ? v[k] : 0.f;
            }
        }
        for (int tg = 0; tg < K.T; tg += K.TG) {
            const int ntap = min(K.TG, K.T - tg);
            const int nslab = (K.TG >= K.T) ? ncs * K.T : ntap;      // CPS > 1 only when all taps fit one stage
            // ---- stage the weight slabs [nslab][KC][TC]: linear float4 copy, three slabs in flight per thread
            {
                constexpr int VPT = KC * TC / 4;   // float4 per slab
                const float4* wsrc = reinterpret_cast<const float4*>(wbase + ((long long)chunk0 * K.T + tg) * (KC * TC));
                float4* wdst = reinterpret_cast<float4*>(Ws);
                // the slabs of a stage are contiguous in the packed weights and in LDS: one linear float4 copy over all
                // 256 threads; small-accumulator variants (little MFMA work per stage to hide behind) keep more loads in flight
                constexpr int WIF = (MT * NT <= 2) ? 8 : 3;
                const int total = nslab * VPT;
                for (int i0 = 0; i0 < total; i0 += WIF * NTHREADS) {
                    float4 w[WIF];
                    int ii[WIF];
#pragma unroll
                    for (int j = 0; j < WIF; ++j) { ii[j] = min(i0 + j * NTHREADS + tid, total - 1); w[j] = wsrc[ii[j]]; }
#pragma unroll
                    for (int j = 0; j < WIF; ++j) wdst[ii[j]] = w[j];
                }
            }
            __syncthreads();
            // Reduction order inside an 8-channel chunk: (tap, channel) for most layers; the 3x3/stride-1 family
            // uses (4-channel half, tap, channel) -- the order of conv3x3.hip's 4-channel pipeline stages -- so that
            // both kernels give bit-identical results (K.halves == 2).
            const int n_outer = K.halves, n_inner = 3 - K.halves;      // (1, 2) or (2, 1)
            for (int cs = 0; cs < ncs; ++cs) {
                const float* xb0 = Xs + cs * (KC * K.plane);
                for (int po = 0; po < n_outer; ++po) {
                    int tiy = tg / K.TX, tix = tg - tiy * K.TX;
                    for (int tt = 0; tt < ntap; ++tt) {
                        const int dy = K.dy0 + tiy * K.dstep, dx = K.dx0 + tix * K.dstep;
                        if (++tix == K.TX) { tix = 0; ++tiy; }
                        int boff[NT];
                        const float* xb;
                        if (UPS) {
                            xb = xb0;
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                boff[nt] = (((oy0 + pty[nt] + dy) >> 1) - iy0) * K.PW + (((ox0 + ptx[nt] + dx) >> 1) - ix0) + lane_k * K.plane;
                        } else {
                            xb = xb0 + (dy - K.dy_min) * K.PW + (dx - K.dx_min);   // uniform tap offset
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) boff[nt] = bbase[nt];
                        }
                        const float* As0 = Ws + ((cs * ntap + tt) * KC + lane_k) * TC + wm * (MT * 32) + lane_j;
                        for (int pi = 0; pi < n_inner; ++pi) {
                            const int kb = 4 * (po + pi);                       // first channel of this half
                            const float* As = As0 + kb * TC;
                            const float* xk = xb + kb * K.plane;
#pragma unroll
                            for (int ks = 0; ks < KC / 4; ++ks) {
                                float a[MT], bb[NT];
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt) a[mt] = As[(2 * ks) * TC + mt * 32];
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) bb[nt] = xk[boff[nt] + (2 * ks) * K.plane];
#pragma unroll
                                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                                    for (int nt = 0; nt < NT; ++nt)
                                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], bb[nt], acc[mt][nt], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- epilogue: bias -> act -> (+res) -> (affine) -> store
    // (compile-time loops: with `#pragma unroll` around the inlined helper hipcc gave up scalarising acc / pty / ptx and
    // moved them to scratch)
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

// ------------------------------------------------------------------------------------------------
// weight packing: [cotile][chunk][tap][k][TC]
struct PackTaps { int8_t ky[DCVIC_MAX_TAPS], kx[DCVIC_MAX_TAPS]; };   // travels as a kernel argument: no allocation, no sync

__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int T,
                                 int KH, int KW, int transposed, int TC, int n_chunks, long long total,
                                 const PackTaps tapk) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long long r = i;
    const int col = r % TC; r /= TC;
    const int k = r % KC; r /= KC;
    const int t = r % T; r /= T;
    const int chunk = r % n_chunks; r /= n_chunks;
    const int cotile = (int)r;
    const int co = cotile * TC + col, ci = chunk * KC + k;
    float v = 0.f;
    if (co < Cout && ci < Cin) {
        const int ky = tapk.ky[t], kx = tapk.kx[t];
        const long long idx = transposed ? (((long long)ci * Cout + co) * KH + ky) * KW + kx
                                         : (((long long)co * Cin + ci) * KH + ky) * KW + kx;
        v = w[idx];
    }
    wp[i] = v;
}

// ------------------------------------------------------------------------------------------------
// tile variants: desc.cfg = TC class (fixes the weight pack), the pixel-tile size P is chosen per launch
static const int kClassTC[4] = {128, 64, 32, 96};
static const int kClassNP[4] = {3, 3, 4, 2};
// class 2 also has 64- and 32-pixel tiles (conv_async16.hip only): below the 32x32-per-wave grain, for launches that
// would otherwise leave most CUs idle (N = 1 .. 4 decoding, 16x16 maps)
static const int kClassP[4][4] = {{256, 128, 64, 0}, {256, 128, 64, 0}, {256, 128, 64, 32}, {256, 128, 0, 0}};
static inline int cfg_TC(int c) { return kClassTC[c]; }

// No process-global mutable state: the scheduling switches and the last-variant report are PER CALLING THREAD (debug /
// test aids that never change results), the CU count and kernel attributes are per device.
static thread_local int g_num_cu = 0;          // CUs of the device current at the last entry-point call of this thread
static thread_local int g_last_variant = -1;   // 9000: conv3x3_dma_kernel; else cls*1000 + P (+1 when UPS)
static thread_local int g_tuning_init = 0;
static thread_local int g_use_dma = 1;   // DCVIC_CONV_DMA=0 forces the generic kernel (A/B comparisons, debugging)
static thread_local int g_use_async = 1; // DCVIC_CONV_ASYNC=0 disables conv_async.hip
static thread_local int g_async_fill = 2; // async twin when workgroups <= g_async_fill x CUs (DCVIC_CONV_ASYNC_FILL)
static thread_local int g_async_fill256 = 1; // the same bound for the 256-pixel tiles (DCVIC_CONV_ASYNC_FILL256)
static thread_local int g_use_async16 = 1; // DCVIC_CONV_ASYNC16=0: keep the 32x32x2 build of the async twin for the small tiles too

static int tile_width_log(int Wout) {
    int TWlog = 5;
    while (TWlog > 2 && (1 << TWlog) > Wout && (1 << (TWlog - 1)) >= Wout) --TWlog;
    return TWlog;
}

// Expected relative throughput of a (TC class, P) variant on a launch: useful fraction of the output-channel
// tiles x how well the grid fills the CUs x a tile-efficiency prior (bigger tiles amortise staging better).
static double variant_score(int cls, int P, int Cout, int N, int Hout, int Wout, int upsample, int num_cu) {
    if (P <= 0) return -1.0;
    if (cls == 2 && P < 128 && (upsample || !g_use_async || !g_use_async16)) return -1.0;
    const int TC = kClassTC[cls];
    const int TW = 1 << tile_width_log(Wout);
    const int TH = P / TW;
    if (TH < 1 || (upsample && (TH & 1))) return -1.0;
    const int cot = (Cout + TC - 1) / TC;
    const double wg = (double)N * ((Hout + TH - 1) / TH) * ((Wout + TW - 1) / TW) * cot;
    const double useful = (double)Cout / (cot * TC) * ((double)Hout * Wout) / ((double)((Hout + TH - 1) / TH) * TH * ((Wout + TW - 1) / TW) * TW);
    const double fill = wg >= 1.5 * num_cu ? 1.0 : wg / (1.5 * num_cu);
    const double effP = P == 256 ? 1.0 : (P == 128 ? 0.93 : (P == 64 ? 0.85 : 0.75));
    const double effC = TC == 128 ? 1.0 : (TC == 96 ? 0.97 : (TC == 64 ? 0.93 : 0.80));
    return useful * fill * effP * effC;
}

static int choose_class(int Cout) {
    if (Cout <= 32) return 2;
    if (Cout <= 64) return 1;
    if (Cout == 96 || Cout == 192) return 3;
    if (Cout % 128 == 0 || Cout > 192) return 0;
    return 1;
}

extern "C" int dcvic_conv_desc_init(dcvic_conv_desc* d, int Cin, int Cout, int KH, int KW, int stride, int pad_t,
                                    int pad_l, int upsample) {
    DCVIC_CHECK_ARG(d && Cin > 0 && Cout > 0, "conv_desc_init: bad channels");
    DCVIC_CHECK_ARG(KH >= 1 && KW >= 1 && KH * KW <= DCVIC_MAX_TAPS, "conv_desc_init: kernel %dx%d unsupported", KH, KW);
    DCVIC_CHECK_ARG(stride == 1 || stride == 2, "conv_desc_init: stride %d unsupported", stride);
    DCVIC_CHECK_ARG(!(upsample && stride != 1), "conv_desc_init: upsample needs stride 1");
    memset(d, 0, sizeof(*d));
    d->Cin = Cin; d->Cout = Cout; d->KH = KH; d->KW = KW; d->stride = stride; d->upsample = upsample;
    d->T = KH * KW;
    for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) {
            const int t = ky * KW + kx;
            d->tap_ky[t] = (int8_t)ky; d->tap_kx[t] = (int8_t)kx;
            d->tap_dy[t] = (int8_t)(ky - pad_t); d->tap_dx[t] = (int8_t)(kx - pad_l);
        }
    d->cfg = choose_class(Cout);
    return DCVIC_OK;
}

extern "C" int dcvic_convT_phase_desc(dcvic_conv_desc* d, int Cin, int Cout, int k, int py, int px) {
    DCVIC_CHECK_ARG(d && Cin > 0 && Cout > 0, "convT_phase_desc: bad channels");
    memset(d, 0, sizeof(*d));
    d->Cin = Cin; d->Cout = Cout; d->KH = k; d->KW = k; d->stride = 1; d->upsample = 0; d->transposed_weight = 1;
    int t = 0;
    if (k == 3) {
        // ConvTranspose2d(k3,s1,p1): out[oy] += in[iy] * w[ky], oy = iy - 1 + ky  =>  dy = 1 - ky
        DCVIC_CHECK_ARG(py == 0 && px == 0, "convT k3 has a single phase");
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
                d->tap_ky[t] = (int8_t)ky; d->tap_kx[t] = (int8_t)kx;
                d->tap_dy[t] = (int8_t)(1 - ky); d->tap_dx[t] = (int8_t)(1 - kx); ++t;
            }
    } else if (k == 5) {
        // ConvTranspose2d(k5,s2,p2,op1): oy = 2*iy - 2 + ky.  For oy = 2m + py: ky = py (mod 2), iy = m + (py + 2 - ky)/2
        DCVIC_CHECK_ARG((py == 0 || py == 1) && (px == 0 || px == 1), "convT k5 phases are 0/1");
        for (int ky = py; ky < 5; ky += 2)
            for (int kx = px; kx < 5; kx += 2) {
                d->tap_ky[t] = (int8_t)ky; d->tap_kx[t] = (int8_t)kx;
                d->tap_dy[t] = (int8_t)((py + 2 - ky) / 2); d->tap_dx[t] = (int8_t)((px + 2 - kx) / 2); ++t;
            }
    } else {
        DCVIC_CHECK_ARG(false, "convT kernel %d unsupported", k);
    }
    d->T = t;
    d->cfg = choose_class(Cout);
    return DCVIC_OK;
}

static inline int n_chunks_of(const dcvic_conv_desc* d) { return (d->Cin + KC - 1) / KC; }
static inline int n_cotiles_of(const dcvic_conv_desc* d) { return (d->Cout + cfg_TC(d->cfg) - 1) / cfg_TC(d->cfg); }

extern "C" size_t dcvic_conv_packed_bytes(const dcvic_conv_desc* d) {
    if (!d || d->cfg < 0 || d->cfg > 3) return 0;
    return (size_t)n_cotiles_of(d) * n_chunks_of(d) * d->T * KC * cfg_TC(d->cfg) * sizeof(float);
}

extern "C" int dcvic_conv_pack_f32(const dcvic_conv_desc* d, const float* w, float* packed, void* stream) {
    DCVIC_CHECK_ARG(d && w && packed, "conv_pack: null pointer");
    DCVIC_CHECK_ARG(d->cfg >= 0 && d->cfg <= 3, "conv_pack: bad cfg %d", d->cfg);
    const int TC = cfg_TC(d->cfg);
    const long long total = (long long)n_cotiles_of(d) * n_chunks_of(d) * d->T * KC * TC;
    PackTaps taps;
    for (int t = 0; t < DCVIC_MAX_TAPS; ++t) { taps.ky[t] = t < d->T ? d->tap_ky[t] : 0; taps.kx[t] = t < d->T ? d->tap_kx[t] : 0; }
    conv_pack_kernel<<<dcvic_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, packed, d->Cin, d->Cout, d->T, d->KH, d->KW,
                                                                        d->transposed_weight, TC, n_chunks_of(d), total, taps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { dcvic_set_error("conv_pack: %s", hipGetErrorString(e)); return DCVIC_ELAUNCH; }
    return DCVIC_OK;
}

template <int MT, int NT, int WM, int WN>
static int launch_variant(const ConvKArgs& K, bool ups, size_t lds, hipStream_t st) {
    static std::atomic<unsigned> attr_mask{0};
    auto k0 = conv_mfma_kernel<MT, NT, WM, WN, false>;
    auto k1 = conv_mfma_kernel<MT, NT, WM, WN, true>;
    if (DcvicAttrOnce once_{attr_mask}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (ups) k1<<<K.nblocks, NTHREADS, lds, st>>>(K);
    else k0<<<K.nblocks, NTHREADS, lds, st>>>(K);
    DCVIC_CHECK_LAUNCH("conv2d");
    return DCVIC_OK;
}

static void init_num_cu() {
    g_num_cu = dcvic_num_cu();
    if (!g_tuning_init) {
        g_tuning_init = 1;
        const char* e = getenv("DCVIC_CONV_DMA");
        if (e && e[0] == '0') g_use_dma = 0;
        e = getenv("DCVIC_CONV_ASYNC");
        if (e && e[0] == '0') g_use_async = 0;
        e = getenv("DCVIC_CONV_ASYNC16");
        if (e && e[0] == '0') g_use_async16 = 0;
        e = getenv("DCVIC_CONV_ASYNC_FILL");
        if (e && atoi(e) > 0) g_async_fill = atoi(e);
        e = getenv("DCVIC_CONV_ASYNC_FILL256");
        if (e && atoi(e) > 0) g_async_fill256 = atoi(e);
    }
}

extern "C" int dcvic_conv_last_variant(void) { return g_last_variant; }

extern "C" int dcvic_conv_set_tuning(int use_dma, int use_async, int async_fill) {
    init_num_cu();
    if (use_dma >= 0) g_use_dma = use_dma != 0;
    if (use_async >= 0) { g_use_async = use_async != 0; g_use_async16 = use_async != 2; }   // 2: 32x32x2 build only
    if (async_fill > 0) g_async_fill = async_fill;
    return DCVIC_OK;
}

extern "C" int dcvic_conv_select_class(const dcvic_conv_desc* d, int N, int Hout, int Wout) {
    if (!d || N <= 0 || Hout <= 0 || Wout <= 0) return DCVIC_EINVAL;
    init_num_cu();
    int best_cls = choose_class(d->Cout);
    double best = -1.0;
#ifdef DCVIC_CONV_EXPERIMENTS
    if (const char* e = getenv("DCVIC_FORCE_CLS")) return atoi(e);
#endif
    for (int cls = 0; cls < 4; ++cls)
        for (int pi = 0; pi < kClassNP[cls]; ++pi) {
            const double sc = variant_score(cls, kClassP[cls][pi], d->Cout, N, Hout, Wout, d->upsample, g_num_cu);
            if (sc > best + 1e-9) { best = sc; best_cls = cls; }
        }
    return best_cls;
}

extern "C" int dcvic_conv2d_f32(const dcvic_conv_desc* d, const float* packed, const dcvic_conv_io* io, void* stream) {
    DCVIC_CHECK_ARG(d && packed && io && io->out, "conv2d: null pointer");
    DCVIC_CHECK_ARG(io->n_src >= 1 && io->n_src <= DCVIC_MAX_SRC, "conv2d: n_src %d", io->n_src);
    int csum = 0;
    for (int i = 0; i < io->n_src; ++i) {
        DCVIC_CHECK_ARG(io->src[i].ptr && io->src[i].C > 0, "conv2d: source %d empty", i);
        DCVIC_CHECK_ARG(io->src[i].batch_stride >= (long long)io->src[i].C * io->H * io->W, "conv2d: source %d batch stride too small", i);
        csum += io->src[i].C;
    }
    DCVIC_CHECK_ARG(csum == d->Cin, "conv2d: sources carry %d channels, layer expects %d", csum, d->Cin);
    DCVIC_CHECK_ARG(io->N > 0 && io->H > 0 && io->W > 0 && io->Hout > 0 && io->Wout > 0, "conv2d: bad sizes");
    DCVIC_CHECK_ARG(io->osy >= 1 && io->osx >= 1 && io->ooy >= 0 && io->oox >= 0, "conv2d: bad output scatter");
    DCVIC_CHECK_ARG((io->Hout - 1) * io->osy + io->ooy < io->Hfull && (io->Wout - 1) * io->osx + io->oox < io->Wfull,
                    "conv2d: output scatter exceeds the output plane");
    DCVIC_CHECK_ARG(io->out_batch_stride >= (long long)d->Cout * io->Hfull * io->Wfull, "conv2d: out batch stride too small");
    DCVIC_CHECK_ARG((long long)io->H * io->W < (1ll << 30) && (long long)io->Hfull * io->Wfull < (1ll << 30), "conv2d: plane too large");
    DCVIC_CHECK_ARG(!io->res || io->res_batch_stride >= (long long)d->Cout * io->Hfull * io->Wfull, "conv2d: res batch stride too small");
    DCVIC_CHECK_ARG((io->aff_scale == nullptr) == (io->aff_shift == nullptr), "conv2d: affine needs both scale and shift");
    DCVIC_CHECK_ARG(!io->init || io->init_batch_stride >= (long long)d->Cout * io->Hfull * io->Wfull, "conv2d: init batch stride too small");
    const int cls = d->cfg;
    DCVIC_CHECK_ARG(cls >= 0 && cls <= 3, "conv2d: bad cfg %d", cls);
    init_num_cu();

    ConvKArgs K;
    memset(&K, 0, sizeof(K));
    K.Cin = d->Cin; K.Cout = d->Cout; K.T = d->T; K.stride = d->stride;
    K.N = io->N; K.H = io->H; K.W = io->W; K.Hout = io->Hout; K.Wout = io->Wout; K.Hfull = io->Hfull; K.Wfull = io->Wfull;
    K.osy = io->osy; K.osx = io->osx; K.ooy = io->ooy; K.oox = io->oox;
    for (int i = 0; i < DCVIC_MAX_SRC; ++i) {
        if (i < io->n_src) { K.src[i] = io->src[i].ptr; K.srcC[i] = io->src[i].C; K.src_bs[i] = io->src[i].batch_stride; }
        else { K.src[i] = io->src[0].ptr; K.srcC[i] = 1 << 30; K.src_bs[i] = 0; }
    }
    K.out = io->out; K.out_bs = io->out_batch_stride; K.bias = io->bias; K.act = io->act;
    K.res = io->res; K.res_bs = io->res_batch_stride; K.affs = io->aff_scale; K.afft = io->aff_shift; K.aff_bs = io->aff_batch_stride;
    K.wp = packed;
    K.init = io->init; K.init_bs = io->init_batch_stride;
    int dy_min = 127, dy_max = -128, dx_min = 127, dx_max = -128;
    // taps must form a regular grid (true for every conv / transposed-conv phase the desc builders emit)
    {
        int TX = 1;
        while (TX < d->T && d->tap_dy[TX] == d->tap_dy[0]) ++TX;
        DCVIC_CHECK_ARG(d->T % TX == 0, "conv2d: taps are not a grid");
        const int step = d->T > 1 ? (TX > 1 ? d->tap_dx[1] - d->tap_dx[0] : d->tap_dy[1] - d->tap_dy[0]) : 1;
        DCVIC_CHECK_ARG(step == 1 || step == -1, "conv2d: tap step %d", step);
        for (int t = 0; t < d->T; ++t)
            DCVIC_CHECK_ARG(d->tap_dy[t] == d->tap_dy[0] + (t / TX) * step && d->tap_dx[t] == d->tap_dx[0] + (t % TX) * step,
                            "conv2d: taps are not a regular grid");
        K.TX = TX; K.dy0 = d->tap_dy[0]; K.dx0 = d->tap_dx[0]; K.dstep = step;
    }
    for (int t = 0; t < d->T; ++t) {
        dy_min = min(dy_min, (int)d->tap_dy[t]); dy_max = max(dy_max, (int)d->tap_dy[t]);
        dx_min = min(dx_min, (int)d->tap_dx[t]); dx_max = max(dx_max, (int)d->tap_dx[t]);
    }
    K.dy_min = dy_min; K.dx_min = dx_min;
    // layer-only predicate (never N / size / tile class): the 3x3 stride-1 pad-1 family shares conv3x3.hip's order
    {
        bool fam = d->T == 9 && K.TX == 3 && K.dstep == 1 && K.dy0 == -1 && K.dx0 == -1 && d->stride == 1 && !d->upsample &&
                   (d->Cin % KC) == 0;
        for (int i = 0; i < io->n_src; ++i) fam = fam && (io->src[i].C % KC) == 0;
        K.halves = fam ? 2 : 1;
    }
    const int TC = cfg_TC(cls);
    K.n_chunks = n_chunks_of(d);
    K.n_cotiles = n_cotiles_of(d);
    // tile width: the largest power of two <= 32 that does not exceed the (rounded-up) output width
    const int TWlog = tile_width_log(io->Wout);
    const int TW = 1 << TWlog;
    K.TWlog = TWlog;
    K.tiles_x = (io->Wout + TW - 1) / TW;
    // pixel-tile size by the occupancy / tile-efficiency score shared with dcvic_conv_select_class
    int P = 0;
    {
        double best = -1.0;
        for (int pi = 0; pi < kClassNP[cls]; ++pi) {
            const double sc = variant_score(cls, kClassP[cls][pi], d->Cout, io->N, io->Hout, io->Wout, d->upsample, g_num_cu);
            if (sc > best) { best = sc; P = kClassP[cls][pi]; }
        }
    }
#ifdef DCVIC_CONV_EXPERIMENTS
    if (const char* e = getenv("DCVIC_FORCE_P")) P = atoi(e);
#endif
    DCVIC_CHECK_ARG(P > 0, "conv2d: no tile variant fits");
    const int TH = P / TW;
    K.tiles_y = (io->Hout + TH - 1) / TH;
    if (d->upsample) {
        K.PH = TH / 2 + ((dy_max - dy_min + 1) >> 1) + 1;
        K.PW = TW / 2 + ((dx_max - dx_min + 1) >> 1) + 1;
    } else {
        K.PH = (TH - 1) * d->stride + (dy_max - dy_min) + 1;
        K.PW = (TW - 1) * d->stride + (dx_max - dx_min) + 1;
    }
    K.plane = K.PH * K.PW;
    DCVIC_CHECK_ARG(K.plane <= MAXSLOT * NTHREADS, "conv2d: patch %dx%d exceeds staging slots", K.PH, K.PW);
    K.nslots = (K.plane + NTHREADS - 1) / NTHREADS;
    // taps per weight stage: keep the slab <= 40 KiB; several channel chunks per stage when a chunk is small
    static const int tg_cap_kb = getenv("DCVIC_TG_CAP_KB") ? atoi(getenv("DCVIC_TG_CAP_KB")) : 40;
    int TG = (tg_cap_kb * 1024) / (KC * TC * 4);
    if (TG < 1) TG = 1;
    if (TG > d->T) TG = d->T;
    K.TG = TG;
    int CPS = 1;
    if (TG == d->T) {
        const int per_chunk = (KC * K.plane + d->T * KC * TC) * 4;
        const int mfma_per_chunk = d->T * (KC / 2) * ((TC / 32) * (P / 32)) / 4;   // per wave
        while (CPS < 8 && (CPS + 1) * per_chunk <= 48 * 1024 && CPS * mfma_per_chunk < 256) ++CPS;
        if (CPS > K.n_chunks) CPS = K.n_chunks;
    }
    K.CPS = CPS;
    const int slabs = (TG == d->T) ? CPS * d->T : TG;
    const size_t lds = (size_t)(((CPS * KC * K.plane + 3) & ~3) + slabs * KC * TC) * sizeof(float);
    DCVIC_CHECK_ARG(lds <= 160 * 1024, "conv2d: LDS %zu too large", lds);
    const long long blocks = (long long)io->N * K.tiles_y * K.tiles_x * K.n_cotiles;
    DCVIC_CHECK_ARG(blocks < (1ll << 31), "conv2d: grid too large");
    K.nblocks = (int)blocks;
    hipStream_t st = (hipStream_t)stream;
    const bool ups = d->upsample != 0;
    if (P == 256 && g_use_dma) {
        int var = 9000;
        const int rc = dcvic_try_conv3x3_dma(K, io->n_src, ups, cls, st, &var);
        if (rc <= 0) { g_last_variant = var; return rc; }
    }
    if (cls == 2 && P < 128) {
        const int rc = dcvic_try_conv_async16(K, cls, P, st);
        if (rc <= 0) { g_last_variant = 8500 + cls * 100 + P / 32; return rc; }
        dcvic_set_error("conv2d: %d-pixel tile not launchable for this layer", P);
        return DCVIC_EINVAL;
    }
    if (g_use_dma && d->T == 1) {
        // 1x1: flat 256-pixel tiles, DMA-pipelined GEMM (needs about a workgroup per CU to pay off)
        const int rc = dcvic_try_conv1x1_dma(K, io->n_src, ups, cls, g_num_cu, st);
        if (rc <= 0) { g_last_variant = 7000 + cls; return rc; }
    }
    // measured: the async twin wins with about one workgroup per CU, and up to g_async_fill per CU for the small tiles
    if (!ups && g_use_async && blocks <= (long long)(P == 256 ? g_async_fill256 : g_async_fill) * g_num_cu) {
        // about one workgroup per CU: nothing hides the staging -> the DMA double-buffered twin (same values)
        // small tiles: the 16x16x4 build (four independent accumulator chains per wave), else the 32x32x2 one
        int rc = g_use_async16 ? dcvic_try_conv_async16(K, cls, P, st) : 1;
        if (rc <= 0) { g_last_variant = 8500 + cls * 100 + P / 32; return rc; }
        rc = dcvic_try_conv_async(K, cls, P, st);
        if (rc <= 0) { g_last_variant = 8000 + cls * 100 + P / 32; return rc; }
    }
    g_last_variant = cls * 1000 + P + (ups ? 1 : 0);
    switch (cls * 1000 + P) {
        case 0 * 1000 + 256: return launch_variant<2, 4, 2, 2>(K, ups, lds, st);
        case 0 * 1000 + 128: return launch_variant<2, 2, 2, 2>(K, ups, lds, st);
        case 0 * 1000 + 64: return launch_variant<2, 1, 2, 2>(K, ups, lds, st);
        case 1 * 1000 + 256: return launch_variant<2, 2, 1, 4>(K, ups, lds, st);
        case 1 * 1000 + 128: return launch_variant<1, 2, 2, 2>(K, ups, lds, st);
        case 1 * 1000 + 64: return launch_variant<1, 1, 2, 2>(K, ups, lds, st);
        case 2 * 1000 + 256: return launch_variant<1, 2, 1, 4>(K, ups, lds, st);
        case 2 * 1000 + 128: return launch_variant<1, 1, 1, 4>(K, ups, lds, st);
        case 3 * 1000 + 256: return launch_variant<3, 2, 1, 4>(K, ups, lds, st);
        case 3 * 1000 + 128: return launch_variant<3, 1, 1, 4>(K, ups, lds, st);
        default: break;
    }
    dcvic_set_error("conv2d: no kernel for class %d P %d", cls, P);
    return DCVIC_EINVAL;
}
