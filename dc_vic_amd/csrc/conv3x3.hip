// conv3x3.hip -- the hot 3x3 / stride-1 / pad-1 convolution (90 % of the path's flops) with the LDS
// double-buffered by LDS-DMA (`global_load_lds`), gfx950.
//
// Same GEMM view, packed-weight layout, reduction order and epilogue as conv.hip (results are
// bit-identical to the generic kernel); what changes is the pipeline:
//   * 512 threads = 8 waves per workgroup on a 128 (out channels) x 256 (8 rows x 32 pixels) tile,
//     waves as 2 (channel halves) x 4 (row pairs), each wave 2x2 MFMA 32x32x2 accumulators, two waves
//     per SIMD so one wave's LDS latency hides behind the other's MFMAs;
//   * a pipeline stage is 4 input channels: the input patch (4 ch x 10 x 34, zero padded via a zero
//     word for out-of-image lanes) and the weight rows (9 taps x 4 x 128) of stage i+1 are written
//     straight into the other LDS buffer by `global_load_lds_dword / _dwordx4` (no VGPR staging, no
//     ds_write) while stage i is computed; one barrier per stage (hipcc drains the DMA with vmcnt(0)
//     in front of it); 48 KiB of LDS per workgroup so that 2-3 workgroups share a CU and one
//     workgroup's barrier / DMA issue hides behind the others' MFMAs;
//   * tap and channel offsets are compile-time immediates of ds_read_b32 -- no address VALU in the loop.
// LDS: 2 x (1536 + 4608) floats = 48 KiB per workgroup.
//
// The kernel is a template over the tap grid: <3,3,4> is the 3x3/stride-1/pad-1 family (4-channel stages, the order
// (chunk, half, tap, channel) that conv.hip mirrors for K.halves == 2); <2,2,8> runs the 2x2 sub-pixel phases of
// "nearest x2 upsample + conv3x3" (8-channel stages, the generic (chunk, tap, channel) order).
#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ float dcvic_zero_word[16];   // zero-initialised: source of padded lanes

// Diagnostic build only (-DDCVIC_STAMPS, tools/build_stamps.sh; never in libdcvic_hip.so): per-wave cycle sums of the three
// phases of a pipeline stage (DMA issue / MFMA loop / barrier), prologue and epilogue, written to a buffer of their own.
#ifdef DCVIC_STAMPS
__device__ unsigned long long* dcvic_stamp_buf;
extern "C" int dcvic_debug_set_stamp_buffer(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(dcvic_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define DCVIC_STAMP() __builtin_amdgcn_s_memtime()
#endif

#define D_TW 32
#define D_TH 8
#define D_THREADS 512
#ifndef DCVIC_CONV_STAGGER
#define DCVIC_CONV_STAGGER 1      // 0: the round-1 schedule (two buffers, all eight waves in step); kept for A/B runs
#endif

// TCV = 128: waves 2 (channel halves) x 4 (row pairs), each 64 ch x 2 rows (MT = NT = 2);
// TCV =  96: waves 1 x 8 (rows), each 96 ch x 1 row (MT = 3, NT = 1) -- the ELIC 96 / 192-channel layers
template <int TY, int TX, int SKC, int TCV>
__global__ __launch_bounds__(D_THREADS, 4) void conv3x3_dma_kernel(const ConvKArgs K) {
    constexpr int D_TC = TCV;
    constexpr int WM = (TCV == 128) ? 2 : 1, WN = 8 / WM;
    constexpr int MT = TCV / (32 * WM), NT = D_TH / WN;
    constexpr int T = TY * TX;
    constexpr int D_PW = D_TW + TX - 1;
    constexpr int D_PLANE = (D_TH + TY - 1) * D_PW;
    constexpr int D_SKC = SKC;                                  // channels per pipeline stage
    constexpr int D_SLOTS = (D_SKC * D_PLANE + D_THREADS - 1) / D_THREADS;
    constexpr int D_XS = D_SLOTS * D_THREADS;
    constexpr int D_WS = T * D_SKC * D_TC;
    constexpr int D_BUF = D_XS + D_WS;
    constexpr int D_CHUNK_W = T * KC * D_TC;                    // floats of one packed 8-channel chunk
    constexpr int NV = T * D_SKC * (D_TC / 4);                  // float4 of weights per stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef DCVIC_STAMPS
    const unsigned long long st_entry = DCVIC_STAMP();
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7, an SGPR: LDS-DMA destinations stay scalar
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
    const int oy0 = tile_y * D_TH, ox0 = tile_x * D_TW;
    const int iy0 = oy0 + K.dy0, ix0 = ox0 + K.dx0;
    const long long HW = (long long)K.H * K.W;

    // Per-slot source pointers of the patch elements this thread moves and per-piece pointers of its weight rows are kept
    // as running 64-bit pointers and advanced by (mostly uniform) strides each stage: the issue path then is two VALU adds
    // and one DMA per piece instead of re-deriving (source, channel, row, column) -- it sits in front of every stage's
    // MFMAs (measured: 4.3k of a wave's 19.7k cycles per stage before).  Out-of-image lanes point at a zero word, stride 0.
    const float* xp[D_SLOTS];
    unsigned xst[D_SLOTS];                                       // bytes per stage (0 for padding lanes)
    int poff[D_SLOTS];
#pragma unroll
    for (int s = 0; s < D_SLOTS; ++s) {
        const int e = tid + s * D_THREADS;
        int o = -1;
        if (e < D_SKC * D_PLANE) {
            const int k = e / D_PLANE, r = e - k * D_PLANE;
            const int py = r / D_PW, px = r - py * D_PW;
            const int iy = iy0 + py, ix = ix0 + px;
            if (iy >= 0 && iy < K.H && ix >= 0 && ix < K.W) o = (int)(k * HW) + iy * K.W + ix;
        }
        poff[s] = o;
        xst[s] = o >= 0 ? (unsigned)(D_SKC * HW * 4) : 0u;
    }
    int x_si = 0, x_left = 0;                                    // source of the NEXT stage, channels left in it
    auto x_rebase = [&](int c) {                                 // (re)derive the pointers for absolute channel c
        int si = 0;
        if (c >= K.srcC[0]) { c -= K.srcC[0]; si = 1; if (c >= K.srcC[1]) { c -= K.srcC[1]; si = 2; } }
        const float* base = K.src[si] + (long long)n * K.src_bs[si] + (long long)c * HW;
#pragma unroll
        for (int s = 0; s < D_SLOTS; ++s) xp[s] = poff[s] >= 0 ? base + poff[s] : dcvic_zero_word;
        x_si = si; x_left = K.srcC[si] - c;
    };
    x_rebase(0);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float* wbase = K.wp + (long long)cotile * K.n_chunks * (long long)D_CHUNK_W;
    const int n_stages = K.n_chunks * (KC / D_SKC);

    // weight pieces: float4 index v = tid + j * 512 of the stage's [tap][SKC][128] rows -> (tap t, element i) is fixed per
    // thread; the stage base walks the pack: + SKC rows inside a chunk, then to the next chunk
    constexpr int NWJ = (TY * TX * SKC * (TCV / 4) + D_THREADS - 1) / D_THREADS;
    const float* wp4[NWJ];                                       // float4 rows, kept as float*: with a dependent-size float4* array captured by the lambdas hipcc 7.2 silently drops the kernel's host stub
#pragma unroll
    for (int j = 0; j < NWJ; ++j) {
        const int v = tid + j * D_THREADS;
        const int t = v / (D_SKC * D_TC / 4), i = v % (D_SKC * D_TC / 4);
        wp4[j] = wbase + 4 * (t * (KC * D_TC / 4) + i);
    }
    int w_sub = 0;                                               // 4-channel sub-stage inside the chunk of the NEXT stage

    // One DMA piece of the NEXT stage: pieces [0, D_SLOTS) are patch slots, [D_SLOTS, D_SLOTS + NWJ) weight rows.  The pieces are
    // issued from INSIDE the MFMA loop of the current stage, one every other step: measured with in-kernel stamps
    // (tools/conv_stamps.py), a separate issue phase at the top of a stage took 4.8k of a wave's 17.8k cycles per stage -- each of
    // its ~75 instructions waits for an issue slot between the other waves' MFMAs -- and because the waves of a workgroup are
    // barrier-synchronised both co-resident workgroups regularly sat in their issue / barrier phases together: 14 % of the
    // matrix-pipe cycles idle.  Interleaved, a wave never stops issuing MFMAs for more than one instruction.
    constexpr int NPIECE = D_SLOTS + NWJ;
    auto issue_piece = [&](auto p_, int buf) {
        constexpr int p = decltype(p_)::value;
        float* xb = smem + buf * D_BUF;
        float* wb = xb + D_XS;
        if constexpr (p < D_SLOTS) {
            __builtin_amdgcn_global_load_lds(xp[p], (lds_ptr_t)(xb + wave * 64 + p * D_THREADS), 4, 0, 0);
        } else {
            constexpr int j = p - D_SLOTS;
            if (j * D_THREADS + tid < NV)                         // whole waves for TCV = 128; the last wave is partial for 96
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float4*>(wp4[j]), (lds_ptr_t)(wb + (wave * 64 + j * D_THREADS) * 4), 16, 0, 0);
        }
    };
    auto advance = [&](int stage) {                               // pointers of the stage after `stage`
        x_left -= D_SKC;
        if (x_left > 0) {
#pragma unroll
            for (int s = 0; s < D_SLOTS; ++s) xp[s] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(xp[s]) + xst[s]);
        } else if (stage + 1 < n_stages) {
            x_rebase((stage + 1) * D_SKC);                       // next source of the virtual concat
        }
        const int wstep = (++w_sub == KC / D_SKC) ? (D_CHUNK_W - (KC / D_SKC - 1) * (D_SKC * D_TC)) : (D_SKC * D_TC);
        if (w_sub == KC / D_SKC) w_sub = 0;
#pragma unroll
        for (int j = 0; j < NWJ; ++j) wp4[j] += wstep;
    };
    auto issue = [&](int stage, int buf) {                        // prologue form: all pieces at once
        dcvic_static_for<0, NPIECE>([&](auto p_) { issue_piece(p_, buf); });
        advance(stage);
    };

    const int xlane = lane_k * D_PLANE + (wn * NT) * D_PW + lane_j;
    const int alane = lane_k * D_TC + wm * (MT * 32) + lane_j;
    constexpr int NSTEP = T * D_SKC / 2;
#if DCVIC_CONV_STAGGER
    // ---- staggered schedule: THREE staging buffers, waves 4..7 run HALF A STAGE behind waves 0..3.
    // Waves w and w + 4 share a SIMD and run the same program with one barrier per stage: in lockstep both reach the barrier,
    // the post-barrier operand reads and the DMA issue together and leave the SIMD's matrix pipe to the other workgroup alone
    // (MI355X_MICROARCH "two waves per SIMD", item 9).  Half-periods h = 0 .. 2 n: the leading group runs (stage, half) =
    // (h >> 1, h & 1), the trailing group the same one half-period later; everybody meets at ONE barrier per period (even h).
    // In every odd half-period each wave issues its DMA pieces of stage (h + 1) / 2 into ring[stage % 3], whose previous
    // occupant (stage - 3) was last read two barriers ago; the barrier at h = 2 i then makes stage i complete for both groups.
    // Per-wave arithmetic is unchanged (same MFMA sequence), so results stay bit-identical.
    constexpr int HS = NSTEP / 2;
    static_assert(NSTEP % 2 == 0 && NPIECE <= HS, "half a stage must hold the DMA pieces");
    float* const sbias = smem + 3 * D_BUF;
    if (tid < D_TC) sbias[tid] = K.bias ? K.bias[min(cotile * D_TC + tid, K.Cout - 1)] : 0.f;
    issue(0, 0);
    const int grp = wave >> 2;
    auto run_half = [&](auto part_, auto issue_, int stage) {
        constexpr int part = decltype(part_)::value;
        constexpr bool do_issue = decltype(issue_)::value;
        const int buf = stage % 3, nbuf = (stage + 1) % 3;
        const bool more = stage + 1 < n_stages;
        const float* xb = smem + buf * D_BUF + xlane;
        const float* wb = smem + buf * D_BUF + D_XS + alane;
        float a_cur[MT], b_cur[NT], a_nxt[MT], b_nxt[NT];
        {
            constexpr int s0 = part * HS, t = s0 / (D_SKC / 2), ks = s0 % (D_SKC / 2), ky = t / TX, kx = t - TX * ky;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a_cur[mt] = wb[(t * D_SKC + 2 * ks) * D_TC + mt * 32];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b_cur[nt] = xb[(2 * ks) * D_PLANE + (nt + ky) * D_PW + kx];
        }
        dcvic_static_for<0, HS>([&](auto i_) {
            constexpr int i = decltype(i_)::value, step = part * HS + i;
            if constexpr (do_issue && i < NPIECE) {
                if (more) issue_piece(std::integral_constant<int, i>{}, nbuf);
            }
            if constexpr (i + 1 < HS) {
                constexpr int t = (step + 1) / (D_SKC / 2), ks = (step + 1) % (D_SKC / 2), ky = t / TX, kx = t - TX * ky;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = wb[(t * D_SKC + 2 * ks) * D_TC + mt * 32];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = xb[(2 * ks) * D_PLANE + (nt + ky) * D_PW + kx];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch of step s+1 ahead of the MFMAs of step s
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[mt], b_cur[nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < MT; ++k) a_cur[k] = a_nxt[k];
#pragma unroll
            for (int k = 0; k < NT; ++k) b_cur[k] = b_nxt[k];
        });
        if (do_issue && more) advance(stage + 1);
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
#ifdef DCVIC_STAMPS
    const unsigned long long st_loop0 = DCVIC_STAMP();
#endif
    // one straight-line loop per group (a single loop with per-half branches made hipcc shuffle the 64 accumulators through
    // scratch at every branch): n + 1 barriers on both sides
    if (grp == 0) {
        for (int stage = 0; stage < n_stages; ++stage) {
            __syncthreads();                                   // stage `stage` has landed; everyone is past stage - 2
            run_half(P0{}, std::false_type{}, stage);
            run_half(P1{}, std::true_type{}, stage);           // + DMA pieces of stage + 1
        }
        __syncthreads();
    } else {
        __syncthreads();
        for (int stage = 0; stage < n_stages; ++stage) {
            run_half(P0{}, std::true_type{}, stage);           // + DMA pieces of stage + 1
            __syncthreads();                                   // (the leading group is entering stage + 1)
            run_half(P1{}, std::false_type{}, stage);
        }
    }
#ifdef DCVIC_STAMPS
    const unsigned long long st_loop1 = DCVIC_STAMP();
#define DCVIC_STAMP_END()                                                                                             \
    if (dcvic_stamp_buf && lane == 0) {                                                                               \
        unsigned long long* o_ = dcvic_stamp_buf + ((long long)blockIdx.x * 8 + wave) * 8;                            \
        o_[0] = st_entry; o_[1] = st_loop0; o_[2] = st_loop1; o_[3] = DCVIC_STAMP();                                  \
        o_[4] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); o_[5] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); \
        o_[6] = (unsigned long long)n_stages;                                                                          \
    }
#else
#define DCVIC_STAMP_END()
#endif
#else
    // the workgroup's bias values go to LDS (read in the epilogue without touching vmcnt)
    float* const sbias = smem + 2 * D_BUF;
    if (tid < D_TC) sbias[tid] = K.bias ? K.bias[min(cotile * D_TC + tid, K.Cout - 1)] : 0.f;
    issue(0, 0);
    __syncthreads();

#ifdef DCVIC_STAMPS
    unsigned long long st_issue = 0, st_mfma = 0, st_bar = 0;
    const unsigned long long st_loop0 = DCVIC_STAMP();
#endif
    for (int stage = 0; stage < n_stages; ++stage) {
        const int buf = stage & 1;
#ifdef DCVIC_STAMPS
        const unsigned long long st_a = DCVIC_STAMP();
#endif
        const bool more = stage + 1 < n_stages;
#ifdef DCVIC_STAMPS
        const unsigned long long st_b = DCVIC_STAMP();
        st_issue += st_b - st_a;
#endif
        const float* xb = smem + buf * D_BUF + xlane;
        const float* wb = smem + buf * D_BUF + D_XS + alane;
        // T x SKC/2 steps (taps x channel pairs), software pipelined: the fragments of step s+1 are in flight
        // while the four MFMAs of step s issue
        float a_cur[MT], b_cur[NT], a_nxt[MT], b_nxt[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a_cur[mt] = wb[mt * 32];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b_cur[nt] = xb[nt * D_PW];
        static_assert(2 * NPIECE <= NSTEP, "one DMA piece every other MFMA step must fit the stage");
        dcvic_static_for<0, NSTEP>([&](auto step_) {
            constexpr int step = decltype(step_)::value;
            if constexpr ((step & 1) == 1 && step / 2 < NPIECE) {
                if (more) issue_piece(std::integral_constant<int, step / 2>{}, buf ^ 1);
            }
            if (step + 1 < T * D_SKC / 2) {
                const int t = (step + 1) / (D_SKC / 2), ks = (step + 1) % (D_SKC / 2);
                const int ky = t / TX, kx = t - TX * ky;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) a_nxt[mt] = wb[(t * D_SKC + 2 * ks) * D_TC + mt * 32];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b_nxt[nt] = xb[(2 * ks) * D_PLANE + (nt + ky) * D_PW + kx];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch of step s+1 ahead of the MFMAs of step s
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
        });
        if (more) advance(stage + 1);
#ifdef DCVIC_STAMPS
        const unsigned long long st_c = DCVIC_STAMP();
        st_mfma += st_c - st_b;
#endif
        if (stage + 1 < n_stages) __syncthreads();             // nothing reads the staging buffers after the last stage
#ifdef DCVIC_STAMPS
        st_bar += DCVIC_STAMP() - st_c;
#endif
    }

#endif
#ifndef DCVIC_STAMP_END
#define DCVIC_STAMP_END()
#endif
    // ---- epilogue: bias -> act -> (+res) -> (affine) -> store  (same order as conv.hip)
    const long long HWo = (long long)K.Hfull * K.Wfull;
    // Fast form for interior tiles without the affine: the residuals of the next 8-value sub-group are requested BEFORE
    // the current sub-group's stores, so the wait for them is a counted vmcnt that leaves those stores in flight (vmcnt
    // counts stores on gfx950: with loads and stores alternating batch by batch every load-wait drained the previous
    // stores -- 87k / 139k cycles per workgroup, bias only / bias + residual).  Bias comes from LDS; residual and output
    // share one 32-bit byte offset from scalar bases.
    if (oy0 + D_TH <= K.Hout && ox0 + D_TW <= K.Wout && (cotile + 1) * D_TC <= K.Cout && !K.affs &&
        (long long)K.Cout * HWo * 4 < (1ll << 32)) {
        const char* const rb = reinterpret_cast<const char*>(K.res ? K.res + (long long)n * K.res_bs : nullptr);
        char* const ob = reinterpret_cast<char*>(K.out + (long long)n * K.out_bs);
        const bool has_res = K.res != nullptr, has_bias = K.bias != nullptr;
        const int act = K.act;
        constexpr int NG = MT * NT;
        constexpr int NS = 2 * NG, SB = 8;                                 // sub-groups of 8 values: register budget (128)
        unsigned goff[NG];                                                 // byte offset of element r = 0 of group g
        dcvic_static_for<0, NG>([&](auto g_) {
            constexpr int g = decltype(g_)::value, mt = g % MT, nt = g / MT;
            const int oy = oy0 + wn * NT + nt, ox = ox0 + lane_j;
            const long long pix = (long long)(oy * K.osy + K.ooy) * K.Wfull + (ox * K.osx + K.oox);
            goff[g] = (unsigned)(4 * ((long long)(cotile * D_TC + (wm * MT + mt) * 32 + 4 * lane_k) * HWo + pix));
        });
        const unsigned rstep = (unsigned)(4 * HWo);                        // one channel, in bytes
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
                float e = acc[mt][g / MT][r];
                if (has_bias) e += sbias[(wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lane_k];   // (no "+ 0": keeps -0)
                e = dcvic_act(e, act);
                if (has_res) e += rv[sg & 1][r - r0];
                v[r - r0] = e;
            });
            __builtin_amdgcn_sched_barrier(0);
            dcvic_static_for<0, SB>([&](auto r_) {
                constexpr int r = r0 + decltype(r_)::value;
                *reinterpret_cast<float*>(ob + (goff[g] + (unsigned)((r & 3) + 8 * (r >> 2)) * rstep)) = v[r - r0];
            });
        });
        DCVIC_STAMP_END()
        return;
    }
    dcvic_epilogue_dispatch(K, [&](auto res_, auto aff_) {
        constexpr bool RES = decltype(res_)::value, AFF = decltype(aff_)::value;
        dcvic_static_for<0, NT>([&](auto nt_) {
            constexpr int nt = decltype(nt_)::value;
            const int oy = oy0 + wn * NT + nt, ox = ox0 + lane_j;
            if (oy < K.Hout && ox < K.Wout) {
                const long long pix = (long long)(oy * K.osy + K.ooy) * K.Wfull + (ox * K.osx + K.oox);
                dcvic_static_for<0, MT>([&](auto mt_) {
                    constexpr int mt = decltype(mt_)::value;
                    const int cob = cotile * D_TC + (wm * MT + mt) * 32 + 4 * lane_k;
                    dcvic_conv_epilogue<16, (AFF ? 4 : 8), RES, AFF>(K, n, acc[mt][nt], [cob](int r) { return cob + (r & 3) + 8 * (r >> 2); }, pix, HWo);
                });
            }
        });
    });
    DCVIC_STAMP_END()
}

template <int TY, int TX, int SKC, int TCV>
static int launch_tap_dma(const ConvKArgs& A, hipStream_t st) {
    static std::atomic<unsigned> attr_mask{0};
    auto k = conv3x3_dma_kernel<TY, TX, SKC, TCV>;
    constexpr int PLANE = (D_TH + TY - 1) * (D_TW + TX - 1);
    constexpr int XS = ((SKC * PLANE + D_THREADS - 1) / D_THREADS) * D_THREADS;
    const size_t lds = (size_t)((DCVIC_CONV_STAGGER ? 3 : 2) * (XS + TY * TX * SKC * TCV) + TCV) * sizeof(float);   // + the bias row
    if (DcvicAttrOnce once_{attr_mask}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    k<<<A.nblocks, D_THREADS, lds, st>>>(A);
    DCVIC_CHECK_LAUNCH("conv_tap_dma");
    return DCVIC_OK;
}

// returns DCVIC_OK (variant_out: 9000 = 3x3, 9001 = 2x2 phases, 9003 = 3x3 with 96-channel tiles) after launching, or 1 if the layer is not eligible
int dcvic_try_conv3x3_dma(const ConvKArgs& Kin, int n_src, bool upsample, int cls, hipStream_t st, int* variant_out) {
    const ConvKArgs& K = Kin;
    if (upsample || (cls != 0 && cls != 3) || K.TWlog != 5 || K.init || K.stride != 1 || K.dstep != 1) return 1;
    if ((long long)K.H * K.W * KC >= (1ll << 31)) return 1;
    const bool fam3 = K.halves == 2;                                  // 3x3/s1/p1 family, channels % 8 == 0
    bool ph2 = !fam3 && K.T == 4 && K.TX == 2;                        // 2x2 sub-pixel phase of upsample + conv3x3
    if (ph2) {
        if (K.Cin % KC) ph2 = false;
        for (int i = 0; i < n_src; ++i) if (K.srcC[i] % KC) ph2 = false;
    }
    if (!fam3 && !ph2) return 1;
    if (cls == 3 && !fam3) return 1;                                  // 96-channel tiles: the 3x3 family only
    ConvKArgs A = K;
    A.tiles_y = (K.Hout + D_TH - 1) / D_TH;
    A.tiles_x = (K.Wout + D_TW - 1) / D_TW;
    const long long blocks = (long long)K.N * A.tiles_y * A.tiles_x * K.n_cotiles;
    if (blocks >= (1ll << 31)) return 1;
    A.nblocks = (int)blocks;
    if (variant_out) *variant_out = cls == 3 ? 9003 : (fam3 ? 9000 : 9001);
    if (cls == 3) return launch_tap_dma<3, 3, 4, 96>(A, st);
    return fam3 ? launch_tap_dma<3, 3, 4, 128>(A, st) : launch_tap_dma<2, 2, 8, 128>(A, st);
}
