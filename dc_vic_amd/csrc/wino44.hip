// wino44.hip -- 3x3 / stride-1 / pad-1 convolution as Winograd F(4x4, 3x3) on fp32 MFMA, gfx950.
//
// For the layers AFTER the path's last integer decision only (dc_vic_amd.layers.allow_winograd(..., f44=True)): the frozen VQGAN
// decoder and the SFT fusion blocks (ldm/modules/diffusionmodules/model.py:82-141, 462-568; src/models/layer/codeformer_layers.py:20-67;
// src/models/subnet/vq_fusion_module.py:78-126).  Y = A^T [ sum_ci (G g G^T) . (B^T d B) ] A with 6x6 transformed tiles: 36 multiplies
// per 4x4 outputs and input channel = 2.25 per output (F(2x2, 3x3): 4, the direct sum: 9).  The price is accuracy: the transforms
// amplify fp32 rounding (about 3x the F(2x2) error per layer), which is why it never runs upstream of the VQ argmin, the symbol
// rounding or the estimator argmax: bitstreams and indices cannot change, only the reconstruction at the 1e-5 level (the contract
// is 1e-3).  Interpolation points 0, +-3/4, +-3/2, inf instead of the textbook 0, +-1, +-2, inf: every constant of B^T and A^T stays a
// dyadic rational (exact in fp32) and the measured error halves (rms 1.5e-6 against 3.2e-6 per layer; oracle/ notes in DESIGN.md).
//
//   B^T d (6 -> 6):  t0 = 81/64 d0 - 45/16 d2 + d4          t5 = 81/64 d1 - 45/16 d3 + d5
//                    t1, t2 = (d4 - 9/4 d2) +- 3/4 (d3 - 9/4 d1)    t3, t4 = (d4 - 9/16 d2) +- 3/2 (d3 - 9/16 d1)      (12 fma)
//   A^T m (6 -> 4):  y0 = m0 + (m1 + m2) + (m3 + m4)        y1 = 3/4 (m1 - m2) + 3/2 (m3 - m4)
//                    y2 = 9/16 (m1 + m2) + 9/4 (m3 + m4)    y3 = 27/64 (m1 - m2) + 27/8 (m3 - m4) + m5
//   G (fp64, at pack time): rows [64/81, 0, 0], [-128/243, -+32/81, -8/27] (x2), [32/243, +-16/81, 8/27] (x2), [0, 0, 1]
//
// One PERSISTENT workgroup per CU = 256 threads = 4 waves, ONE PER SIMD (512 registers each).
//   * workgroup tile: 64 output channels x (16 rows x 32 columns) = 32 tiles of 4x4; a pipeline stage is EIGHT input channels = two
//     k-steps of the MFMA (144 MFMAs per wave and barrier).  Wave cg owns output channels 16 cg .. 16 cg + 15 for all 32 tiles and all
//     36 positions: 72 accumulators of 16x16 (288 registers: 256 in the accumulator file, 32 in VGPRs), so the output transform
//     A^T M A is register-only, exactly as in wino.hip.
//   * the pre-transformed weights U never touch LDS: no two waves of a workgroup share a co group, so each wave fetches ITS 16 co x
//     4 ch x 36 positions of the next k-step straight into operand registers (nine `global_load_dwordx4` per k-step from the packed
//     image [k-step 2][position group 9][cg 4][k 4][m 16][4 positions]; L2-resident: the tile order keeps an XCD on one co-tile).
//     (First build: U through LDS by LDS-DMA like wino.hip -- 12 DMA pieces per 72 MFMAs cost 0.54 of 1.90 ms, ~105 cycles of the
//     SIMD's issue each: profiles/r3_wino44_experiments.md.)
//   * the raw input patch (8 ch x 18 rows x ten 16-byte segments) arrives by LDS-DMA (`global_load_lds_dwordx4`, six pieces per wave
//     and stage) two stages ahead.
//   * input transform: 256 (channel, tile) patches per stage = ONE 6x6 patch per thread and stage (18 LDS reads, column pass + row
//     pass = 144 fma, 12 stores into the V image [k-step][position group][block 2][k 4][n 16][4 positions]), its instructions placed
//     one quarter-pass (3 fma) behind every second MFMA.
//   * MFMA operands: A (U) from registers, B (V) by two ds_read_b128 (four positions x two 16-tile blocks) per 8 MFMAs.
// LDS: 2 x 24 KiB raw patch + 2 x 36 KiB V + two bias rows = 120.5 KiB.
// Deterministic and batch-invariant: per position the reduction runs over the 4-channel k-steps ascending inside the MFMA's ordered
// fmaf chain; the tiling never depends on N.
#include "conv_common.h"

// DCVIC_W44_DBG: timing experiments with WRONG results, only in the diagnostic builds of tools/build_w44_experiments.sh (never in
// libdcvic_hip.so): 1 no stage barrier, 2 no input transform, 4 no LDS-DMA, 8 no operand reads / waits, 16 no tile epilogue,
// 32 no MFMAs, 64 no vmcnt wait in front of the stage barrier
#ifndef DCVIC_W44_DBG
#define DCVIC_W44_DBG 0
#endif

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ float dcvic_w44_zero[16];   // zero-initialised: source of padded lanes

#define F4_TH 16
#define F4_TW 32
#define F4_PW 40           // LDS row: columns ox0 - 4 .. ox0 + 35 as ten 16-byte segments; the patch's 34 columns sit at 3 .. 36
#define F4_PROWS 18
#define F4_PLANE 720
#define F4_KC 8            // input channels per stage (two k-steps of four)
#define F4_SEGS 1440       // float4 segments of a stage: 8 ch x 18 rows x 10
#define F4_XSLOTS 6
#define F4_XS 6144         // floats (1536 lanes x 4: the last slot's idle lanes write zeros behind the patch)
#define F4_US 18432        // floats of a stage's packed weights: 2 k-steps x 36 positions x 4 ch x 64 co
#define F4_VS 9216         // 2 k-steps x 36 positions x 4 ch x 32 tiles
#define F4_CO 64
#define F4_THREADS 256
#define F4_OFF_V (2 * F4_XS)
#define F4_OFF_BIAS (F4_OFF_V + 2 * F4_VS)
#define F4_LDS_FLOATS (F4_OFF_BIAS + 2 * F4_CO)

// Slot of Winograd position (a, b) in the accumulator / LDS order: groups 0..5 hold (a, 1..4) -- the transform's natural float4 --
// groups 6..8 the twelve edge values (a, 0), (a, 5).
__host__ __device__ constexpr int f4_pos(int a, int b) { return (b >= 1 && b <= 4) ? a * 4 + (b - 1) : 24 + 2 * a + (b == 5 ? 1 : 0); }

__device__ __forceinline__ double f4_u(const float* g, int a, int b) {
    const double G[6][3] = {{64.0 / 81.0, 0.0, 0.0},
                            {-128.0 / 243.0, -32.0 / 81.0, -8.0 / 27.0}, {-128.0 / 243.0, 32.0 / 81.0, -8.0 / 27.0},
                            {32.0 / 243.0, 16.0 / 81.0, 8.0 / 27.0}, {32.0 / 243.0, -16.0 / 81.0, 8.0 / 27.0},
                            {0.0, 0.0, 1.0}};
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) s += G[a][r] * (double)g[r * 3 + c] * G[b][c];
    return s;
}

// packed[cotile][chunk][k-step 2][group 9][cg 4][k 4][m 16][slot 4]  <-  w[Cout][Cin][3][3]   (fp64 transform, rounded once)
__global__ void wino44_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int n_chunks, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long long r = i;
    const int slot = r & 3; r >>= 2;
    const int m = r & 15; r >>= 4;
    const int k = r & 3; r >>= 2;
    const int cg = r & 3; r >>= 2;
    const int pg = (int)(r % 9); r /= 9;
    const int ks = (int)(r & 1); r >>= 1;
    const int chunk = (int)(r % n_chunks);
    const int cotile = (int)(r / n_chunks);
    const int idx = pg * 4 + slot;
    int a, b;
    if (idx < 24) { a = idx >> 2; b = (idx & 3) + 1; } else { a = (idx - 24) >> 1; b = ((idx - 24) & 1) ? 5 : 0; }
    const int co = cotile * F4_CO + cg * 16 + m, ci = chunk * F4_KC + 4 * ks + k;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = (float)f4_u(w + ((long long)co * Cin + ci) * 9, a, b);
    wp[i] = v;
}

// one quarter (three fma) of the 1-D input transform B^T x, in place over the lvalues X0..X5 (floats or vector elements); T[9]
// carries the intermediates between the quarters
#define F4_BT_QUARTER(Q, X0, X1, X2, X3, X4, X5, T)                                               \
    do {                                                                                           \
        if constexpr ((Q) == 0) {                                                                  \
            T[0] = __builtin_fmaf(-2.25f, X2, X4);         /* e1 = d4 - 9/4 d2 */                  \
            T[1] = __builtin_fmaf(-2.25f, X1, X3);         /* o1 = d3 - 9/4 d1 */                  \
            T[2] = __builtin_fmaf(-0.5625f, X2, X4);       /* e2 = d4 - 9/16 d2 */                 \
        } else if constexpr ((Q) == 1) {                                                           \
            T[3] = __builtin_fmaf(-0.5625f, X1, X3);       /* o2 = d3 - 9/16 d1 */                 \
            T[4] = __builtin_fmaf(-2.8125f, X2, X4);       /* d4 - 45/16 d2 */                     \
            T[5] = __builtin_fmaf(-2.8125f, X3, X5);       /* d5 - 45/16 d3 */                     \
        } else if constexpr ((Q) == 2) {                                                           \
            T[6] = __builtin_fmaf(1.265625f, X0, T[4]);    /* t0 */                                \
            T[7] = __builtin_fmaf(1.265625f, X1, T[5]);    /* t5 */                                \
            T[8] = __builtin_fmaf(0.75f, T[1], T[0]);      /* t1 */                                \
        } else {                                                                                   \
            X2 = __builtin_fmaf(-0.75f, T[1], T[0]);       /* t2 */                                \
            X3 = __builtin_fmaf(1.5f, T[3], T[2]);         /* t3 */                                \
            X4 = __builtin_fmaf(-1.5f, T[3], T[2]);        /* t4 */                                \
            X0 = T[6]; X1 = T[8]; X5 = T[7];                                                       \
        }                                                                                          \
    } while (0)

// 1-D output transform A^T m (6 -> 4)
__device__ __forceinline__ void f4_at(float m0, float m1, float m2, float m3, float m4, float m5, float& y0, float& y1, float& y2, float& y3) {
    const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
    y0 = (m0 + s12) + s34;
    y1 = __builtin_fmaf(1.5f, d34, 0.75f * d12);
    y2 = __builtin_fmaf(2.25f, s34, 0.5625f * s12);
    y3 = __builtin_fmaf(3.375f, d34, __builtin_fmaf(0.421875f, d12, m5));
}

__global__ __launch_bounds__(F4_THREADS, 1) void conv3x3_wino44_kernel(const ConvKArgs K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..3

    const long long HW = (long long)K.H * K.W;
    const int S = K.n_chunks;                                     // stages (8-channel chunks) per tile
    const long long x_stride = (long long)F4_KC * HW;

    // ---- PERSISTENT workgroup (as wino.hip): XCD x = blockIdx.x % 8 owns a contiguous range of tile indices
    int xe;
    const int J = (int)gridDim.x / NXCD;
    int first;
    {
        const int nb = K.nblocks, q = nb / NXCD, r = nb % NXCD, x = (int)blockIdx.x % NXCD;
        const int xs = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        xe = xs + (x < r ? q + 1 : q);
        first = xs + (int)blockIdx.x / NXCD;
    }
    if (first >= xe) return;                                      // (uniform: the whole workgroup leaves before any barrier)
    const int ntile = (xe - first + J - 1) / J;
    const int total = ntile * S;
    // tile index b = (cotile, image, tile row, tile column), cotile SLOWEST: an XCD's contiguous range then lies inside one or two
    // co-tiles, whose weight slabs (36 positions x Cin x 64 co x 4 B = 2.4 MB at Cin = 256) stay in that XCD's 4 MiB L2 for the whole
    // launch, where every wave's direct weight loads find them.
    const int n_ptiles = K.nblocks / K.n_cotiles;
    auto decode = [&](int b, int& cotile, int& n, int& oy0, int& ox0) __attribute__((always_inline)) {
        cotile = b / n_ptiles; b -= cotile * n_ptiles;
        const int tile_x = b % K.tiles_x; b /= K.tiles_x;
        const int tile_y = b % K.tiles_y; b /= K.tiles_y;
        n = b; oy0 = tile_y * F4_TH; ox0 = tile_x * F4_TW;
    };
    auto cotile_of = [&](int b) __attribute__((always_inline)) { return b / n_ptiles; };

    // ---- raw-patch DMA: float4 segment e = tid + s*256 of [8 ch][18 rows][10 segments]
    const float* xp[F4_XSLOTS];
    int poff[F4_XSLOTS];
    int x_left = 0, x_n = 0, x_b = first, x_next = 0;
    auto x_rebase = [&](int c) __attribute__((always_inline)) {
        int si = 0;
        if (c >= K.srcC[0]) { c -= K.srcC[0]; si = 1; if (c >= K.srcC[1]) { c -= K.srcC[1]; si = 2; } }
        const float* base = K.src[si] + (long long)x_n * K.src_bs[si] + (long long)c * HW;
#pragma unroll
        for (int s = 0; s < F4_XSLOTS; ++s) xp[s] = poff[s] >= 0 ? base + poff[s] : dcvic_w44_zero;
        x_left = K.srcC[si] - c;
    };
    auto x_setup = [&](int b) __attribute__((always_inline)) {
        int cot, oy0, ox0;
        decode(b, cot, x_n, oy0, ox0);
#pragma unroll
        for (int s = 0; s < F4_XSLOTS; ++s) {
            const int e = tid + s * F4_THREADS;
            int o = -1;
            if (e < F4_SEGS) {
                const int k = e / 180, r = e - k * 180;
                const int py = r / 10, seg = r - py * 10;
                const int iy = oy0 - 1 + py, ix = ox0 - 4 + 4 * seg;   // W % 4 == 0: a segment is entirely inside or outside the row
                if (iy >= 0 && iy < K.H && ix >= 0 && ix < K.W) o = (int)(k * HW) + iy * K.W + ix;
            }
            poff[s] = o;
        }
        x_rebase(0);
    };
    x_setup(first);
    // ---- weights: this wave's slice of the packed image, fetched straight into operand registers.  wp_nxt = the slab of the NEXT
    // stage of the stream (its first k-step is loaded during the second k-step of the current stage).
    const int cg = wave;
    const unsigned u_voff = 16u * (unsigned)(cg * 64 + lane);     // byte offset inside a (k-step, group) block of 4 KiB
    const float* wp_cur;
    const float* wp_nxt;
    int u_b = first, u_next = 0;
    auto u_setup = [&](int b) __attribute__((always_inline)) { wp_nxt = K.wp + (long long)cotile_of(b) * S * (long long)F4_US; };
    u_setup(first);
    wp_cur = wp_nxt;
    auto u_advance = [&]() __attribute__((always_inline)) {       // wp_nxt: on to the next stage of the stream
        if (++u_next == S) {
            u_next = 0;
            u_b += J;
            if (u_b < xe) u_setup(u_b);
        } else {
            wp_nxt += F4_US;
        }
    };
    u_advance();

    // ---- input transform: thread -> channel k = tid / 32 (k-step k / 4), tile t = tid % 32 (block t / 16, column n = t % 16 of the
    //      MFMA's B operand); tile (row ty = t / 8, column tx = t % 8) of the 4 x 8 tile grid
    const int t_k = tid >> 5, t_t = tid & 31;
    const int t_blk = t_t >> 4, t_n = t_t & 15, t_ty = t_t >> 3, t_tx = t_t & 7;
    const unsigned t_src = 4u * (unsigned)(t_k * F4_PLANE + (4 * t_ty) * F4_PW + 4 * t_tx + 3);
    const unsigned t_dst = 4u * (unsigned)(F4_OFF_V + (t_k >> 2) * 4608 + (t_blk * 64 + (t_k & 3) * 16 + t_n) * 4);
    const unsigned op_v = 4u * (unsigned)(F4_OFF_V + lane * 4);

    // 72 accumulators of 16x16 = 288 registers, but the accumulator file (AGPRs) holds 256 and hipcc (ROCm 7.2), given the builtin,
    // spills the rest to scratch around every MFMA instead of keeping them in VGPRs, or shuttles all 256 between the two files once
    // per iteration (seen in the .s: 230 scratch / 314 v_accvgpr instructions per stage).  So every MFMA is inline asm: position slots
    // 0..31 with the accumulator constrained to AGPRs ("+a": exactly the 256), slots 32..35 (`accv`) to VGPRs ("+v").  Nothing reads an
    // accumulator before the tile epilogue, 36+ MFMAs later, so the asm needs no hazard padding inside the loop; the epilogue pads once
    // and reads the AGPRs with explicit v_accvgpr_read.
    f32x4 acc[32][2];                                             // [position slot][16-tile block]
    f32x4 accv[4][2];

    // All LDS traffic of the loop is inline asm with hand-placed waits (see wino.hip: hipcc guards every LDS access it can see with
    // `s_waitcnt vmcnt(0)` while an LDS-DMA is in flight); so are the weight loads (their landing is awaited once per k-step).
#define F4_FENCE() __builtin_amdgcn_sched_barrier(0)
#define F4_WAIT_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); F4_FENCE(); } while (0)
    // the 6x6 patch of this thread, row r = (dl[r] | dm[r][0..3] | dr[r]); transformed in place; after the row pass the edge
    // columns live in de[r] = (column 0, column 5)
    float dl[6], dr[6];
    f32x4 dm[6];
    f32x2 de[6];
    float tt[9];
    auto t_load = [&](auto r_, unsigned xaddr) {
        constexpr int r = decltype(r_)::value;
        float &l = dl[r], &rr = dr[r];
        f32x4 &m = dm[r];
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(l) : "v"(xaddr), "n"(4 * (r * F4_PW)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(m) : "v"(xaddr), "n"(4 * (r * F4_PW + 1)));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(rr) : "v"(xaddr), "n"(4 * (r * F4_PW + 5)));
    };
    auto t_col = [&](auto c_, auto q_) {                           // quarter q of the column pass of column c (over the six rows)
        constexpr int c = decltype(c_)::value, q = decltype(q_)::value;
        if constexpr (c == 0) F4_BT_QUARTER(q, dl[0], dl[1], dl[2], dl[3], dl[4], dl[5], tt);
        else if constexpr (c == 5) F4_BT_QUARTER(q, dr[0], dr[1], dr[2], dr[3], dr[4], dr[5], tt);
        else { constexpr int e = c == 0 || c == 5 ? 0 : c - 1; F4_BT_QUARTER(q, dm[0][e], dm[1][e], dm[2][e], dm[3][e], dm[4][e], dm[5][e], tt); }
    };
    auto t_row = [&](auto a_, auto q_) {                           // quarter q of the row pass of row a
        constexpr int a = decltype(a_)::value, q = decltype(q_)::value;
        F4_BT_QUARTER(q, dl[a], dm[a][0], dm[a][1], dm[a][2], dm[a][3], dr[a], tt);
        if constexpr (q == 3) de[a] = f32x2{dl[a], dr[a]};
    };
    auto t_store = [&](auto i_, unsigned vaddr) {                  // i = 0..5: positions (i, 1..4); i = 6..11: positions (i - 6, 0), (i - 6, 5)
        constexpr int i = decltype(i_)::value;
        if constexpr (i < 6) {
            const f32x4 v = dm[i];
            asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(vaddr), "v"(v), "n"(i * 2048) : "memory");
        } else {
            constexpr int a = i - 6;
            const f32x2 v = de[a];
            asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(vaddr), "v"(v), "n"((6 + a / 2) * 2048 + 8 * (a & 1)) : "memory");
        }
    };
    f32x4 uA[2][9];                                               // [k-step][position group]: four positions of this lane's (co, channel)
    f32x4 opB[3][2];                                              // [set = group % 3][block]: four positions of this lane's (channel, tile)
    auto u_load = [&](auto ks_, auto pg_, const float* slab) {     // U of (k-step, group) of the stage whose slab is given -> uA[ks][pg]
        constexpr int ks = decltype(ks_)::value, pg = decltype(pg_)::value;
        f32x4& dst = uA[ks][pg];
        const float* base = slab + (ks * 9 + pg) * 1024;          // (uniform: scalar base + 32-bit lane offset, the saddr form)
        const unsigned vo = u_voff;                               // (asm operands inside a generic lambda do not capture)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(vo), "s"(base) : "memory");
    };
    auto op_load = [&](auto G_, unsigned va) {                    // B operands of group G = 9 ks + pg of a stage
        constexpr int G = decltype(G_)::value, set = G % 3, ks = G / 9, pg = G % 9;
        f32x4 &b0 = opB[set][0], &b1 = opB[set][1];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b0) : "v"(va), "n"(ks * 18432 + 1024 * (2 * pg)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b1) : "v"(va), "n"(ks * 18432 + 1024 * (2 * pg + 1)));
    };
    auto dma_x = [&](auto s_, int buf) {
        constexpr int sl = decltype(s_)::value;
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float4*>(xp[sl]), (lds_ptr_t)(smem + buf * F4_XS + (wave * 64 + sl * F4_THREADS) * 4), 16, 0, 0);
    };
    auto x_advance = [&]() __attribute__((always_inline)) {
        if (++x_next == S) {
            x_next = 0;
            x_b += J;
            if (x_b < xe) x_setup(x_b);
        } else {
            x_left -= F4_KC;
            if (x_left > 0) {
#pragma unroll
                for (int sl = 0; sl < F4_XSLOTS; ++sl) xp[sl] += poff[sl] >= 0 ? x_stride : 0ll;   // (padding lanes stay on the zero word)
            } else {
                x_rebase(x_next * F4_KC);
            }
        }
    };

    // ---- epilogue of one tile, in registers: lane holds element (co = 16 cg + 4 (lane / 16) + r, tile = 16 blk + lane % 16) of all 36
    // positions.  A^T M A per (block, r): 6 column passes + 4 row passes, bias -> act -> (+ res) -> four 16-byte row stores.
    float* const sbias = smem + F4_OFF_BIAS;                      // [2][64], by tile parity
    const int e_n = lane & 15, lq = lane >> 4;
    const float neg_slope = K.act == DCVIC_ACT_RELU ? 0.f : K.act == DCVIC_ACT_LRELU02 ? 0.2f : 1.f;
    const bool has_bias = K.bias != nullptr, has_res = K.res != nullptr;
    auto tile_epilogue = [&](int cotile, int n, int oy0, int ox0, int par) __attribute__((always_inline)) {
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");          // the inline-asm MFMAs' results are read below
        auto A = [&](auto idx_, auto blk_, int r) __attribute__((always_inline)) -> float {
            constexpr int idx = decltype(idx_)::value, blk = decltype(blk_)::value;
            if constexpr (idx < 32) {
                float v;                                          // (explicit read: every use of `acc` names the accumulator file, so the
                const float src = acc[idx][blk][r];               //  allocator never moves the 256 registers into VGPRs and back)
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(src));
                return v;
            } else return accv[idx - 32][blk][r];
        };
        const int co0 = cotile * F4_CO + cg * 16 + 4 * lq;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = has_bias ? sbias[par * F4_CO + cg * 16 + 4 * lq + r] : 0.f;
        float gs[4] = {0.f, 0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};   // GroupNorm partials of this lane's four channels
        dcvic_static_for<0, 2>([&](auto blk_) {
            constexpr int blk = decltype(blk_)::value;
            const int t = blk * 16 + e_n;
            const int oy = oy0 + 4 * (t >> 3), ox = ox0 + 4 * (t & 7);
            const int rows = K.H - oy;                            // output rows of this tile inside the image (>= 4: all of them)
            const bool in_img = ox < K.W && rows > 0;             // W % 4 == 0: all four columns or none
            const long long pix = (long long)oy * K.W + ox;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float tm[4][6];                                   // A^T M: [output row i][position column b]
                dcvic_static_for<0, 6>([&](auto b_) {
                    constexpr int b = decltype(b_)::value;
                    f4_at(A(std::integral_constant<int, f4_pos(0, b)>{}, blk_, r), A(std::integral_constant<int, f4_pos(1, b)>{}, blk_, r),
                          A(std::integral_constant<int, f4_pos(2, b)>{}, blk_, r), A(std::integral_constant<int, f4_pos(3, b)>{}, blk_, r),
                          A(std::integral_constant<int, f4_pos(4, b)>{}, blk_, r), A(std::integral_constant<int, f4_pos(5, b)>{}, blk_, r),
                          tm[0][b], tm[1][b], tm[2][b], tm[3][b]);
                });
                float y[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    f4_at(tm[i][0], tm[i][1], tm[i][2], tm[i][3], tm[i][4], tm[i][5], y[i][0], y[i][1], y[i][2], y[i][3]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[i][j] += bv[r];
                }
                // activation: none / ReLU / LeakyReLU(0.2) only (host check) = one negative-side slope, branch-free: slope 1 is the
                // identity bit for bit.  (The transcendental activations would inline 128 expf expansions into this epilogue.)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[i][j] = y[i][j] > 0.f ? y[i][j] : neg_slope * y[i][j];
                if (in_img && co0 + r < K.Cout) {
                    float* const ob = K.out + (long long)n * K.out_bs + (long long)(co0 + r) * HW + pix;
                    const bool want_gn = K.gn_part != nullptr;
                    if (has_res) {
                        const float* const rb = K.res + (long long)n * K.res_bs + (long long)(co0 + r) * HW + pix;
                        f32x4 rv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) rv[i] = *reinterpret_cast<const f32x4*>(rb + (long long)min(i, rows - 1) * K.W);   // (clamped row: in bounds)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 4; ++j) y[i][j] += rv[i][j];
                    }
                    if (want_gn) {                                // statistics of exactly the values stored (fixed order: row, column)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (i < rows) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) { gs[r] += y[i][j]; gq[r] = __builtin_fmaf(y[i][j], y[i][j], gq[r]); }
                            }
                    }
                    if (rows >= 4) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(ob + (long long)i * K.W) = f32x4{y[i][0], y[i][1], y[i][2], y[i][3]};
                    } else {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
                            if (i < rows) *reinterpret_cast<f32x4*>(ob + (long long)i * K.W) = f32x4{y[i][0], y[i][1], y[i][2], y[i][3]};
                    }
                }
            }
        });
        if (K.gn_part) {
            // GroupNorm statistics from the producer: per (image, channel, pixel tile) the sum and the sum of squares of the
            // 512 stored values -- 32 per lane, then the 16 lanes of a row (same four channels) by xor shuffles, everything in a fixed order.
            // dcvic_groupnorm_part_f32 adds the tiles of a group in fp64 and skips its own statistics pass (one read of the map less).
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { gs[r] += __shfl_xor(gs[r], o, 64); gq[r] += __shfl_xor(gq[r], o, 64); }
            }
            if (e_n == 0) {
                const int pt = (oy0 / F4_TH) * K.tiles_x + ox0 / F4_TW;
                const int npt = K.tiles_x * K.tiles_y;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (co0 + r < K.Cout)
                        *reinterpret_cast<f32x2*>(K.gn_part + (((long long)n * K.Cout + co0 + r) * npt + pt) * 2) = f32x2{gs[r], gq[r]};
            }
        }
    };
    auto stage_bias = [&](int b, int par) __attribute__((always_inline)) {
        if (tid < F4_CO) sbias[par * F4_CO + tid] = has_bias ? K.bias[min(cotile_of(b) * F4_CO + tid, K.Cout - 1)] : 0.f;
    };

    // ---- pipeline
    int c_b = first, c_par = 0;                                   // compute stream: tile, tile parity
    int c_cotile, c_n, c_oy0, c_ox0;
    decode(first, c_cotile, c_n, c_oy0, c_ox0);
    stage_bias(first, 0);
    // prologue: X(0) -> Xr[0], X(1) -> Xr[1], U(stage 0, k-step 0) -> uA[0]; every thread transforms its patch of stage 0 -> V[0]
    dcvic_static_for<0, F4_XSLOTS>([&](auto s_) { dma_x(s_, 0); });
    x_advance();
    if (total > 1) {
        dcvic_static_for<0, F4_XSLOTS>([&](auto s_) { dma_x(s_, 1); });
        x_advance();
    }
    dcvic_static_for<0, 9>([&](auto pg_) { u_load(std::integral_constant<int, 0>{}, pg_, wp_cur); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    F4_FENCE();
    dcvic_static_for<0, 6>([&](auto r_) { t_load(r_, t_src); });
    F4_WAIT_LDS();
    dcvic_static_for<0, 6>([&](auto c_) { dcvic_static_for<0, 4>([&](auto q_) { t_col(c_, q_); }); });
    dcvic_static_for<0, 6>([&](auto a_) { dcvic_static_for<0, 4>([&](auto q_) { t_row(a_, q_); }); });
    dcvic_static_for<0, 12>([&](auto i_) { t_store(i_, t_dst); });
    F4_WAIT_LDS();
    __syncthreads();
    F4_FENCE();
    op_load(std::integral_constant<int, 0>{}, op_v);

    // One stage = 18 operand groups G = 9 ks + pg (k-step, position group) x 8 MFMAs (4 positions x 2 blocks) = 144 slots.  The body has
    // NO run-time condition: past the end of the stream the loads simply re-fetch the last slab / patch (the advance functions stop),
    // the transform works on stale LDS data and writes a V image nobody reads.
    //   in front of a group's MFMAs: wait for its B operands (lgkmcnt), request the next group's; group 9 (the second k-step) also waits
    //   for the weights loaded during the first (vmcnt);
    //   the stage BARRIER sits in front of the LAST group: every LDS read of this stage has returned, the transform's stores, this wave's
    //   DMA pieces and the next stage's first weights have landed; behind it the first B operands of stage s + 1 are requested;
    //   B operands rotate through THREE register sets (group G -> set G % 3: group 17 and the next stage's group 0 never share one).
    auto run_stage = [&](int s) __attribute__((always_inline)) {
        const int cur = s & 1, nxt = cur ^ 1;
        const unsigned va = op_v + (unsigned)(cur * F4_VS * 4);
        const unsigned xaddr = t_src + (unsigned)(nxt * F4_XS * 4);      // X(s + 1) lives in Xr[(s + 1) & 1]
        const unsigned vaddr = t_dst + (unsigned)(nxt * F4_VS * 4);      // V(s + 1)
        dcvic_static_for<0, 18>([&](auto G_) {
            constexpr int G = decltype(G_)::value, ks = G / 9, pg = G % 9, set = G % 3;
            if constexpr (G < 17) {
                if constexpr (!(DCVIC_W44_DBG & 8)) {
                    if constexpr (G == 9) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    F4_FENCE();
                    op_load(std::integral_constant<int, G + 1>{}, va);
                }
            } else {
                if constexpr (DCVIC_W44_DBG & 64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if constexpr (!(DCVIC_W44_DBG & 1)) __syncthreads();
                F4_FENCE();
                if constexpr (!(DCVIC_W44_DBG & 8)) op_load(std::integral_constant<int, 0>{}, op_v + (unsigned)(nxt * F4_VS * 4));
            }
            F4_FENCE();
            dcvic_static_for<0, 8>([&](auto q_) {
                constexpr int q = decltype(q_)::value, ps = q >> 1, blk = q & 1;
                if constexpr (!(DCVIC_W44_DBG & 32)) {
                    const float a_ = uA[ks][pg][ps], b_ = opB[set][blk][ps];
                    if constexpr (pg < 8) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[pg * 4 + ps][blk]) : "v"(a_), "v"(b_));
                    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(accv[ps][blk]) : "v"(a_), "v"(b_));
                }
                F4_FENCE();
                constexpr int sl = 8 * G + q;                     // 0 .. 143
                constexpr int h = sl % 72;                        // slot inside the k-step
                // memory: per k-step three DMA pieces of X(s + 2) (slots 1, 5, 9), then the nine weight loads of the NEXT k-step
                // (slots 13 .. 45): during k-step 0 those of (s, k-step 1), during k-step 1 those of (s + 1, k-step 0)
                if constexpr (!(DCVIC_W44_DBG & 4) && (h & 3) == 1 && h < 12) dma_x(std::integral_constant<int, 3 * ks + h / 4>{}, cur);
                if constexpr (!(DCVIC_W44_DBG & 4) && (h & 3) == 1 && h >= 12 && h < 48) {
                    if constexpr (ks == 0) u_load(std::integral_constant<int, 1>{}, std::integral_constant<int, (h / 4 - 3)>{}, wp_cur);
                    else u_load(std::integral_constant<int, 0>{}, std::integral_constant<int, (h / 4 - 3)>{}, wp_nxt);
                }
                if constexpr (!(DCVIC_W44_DBG & 2)) {
                    // transform of X(s + 1): six row loads at slots 2, 3, 6, 7, 10, 11 (landed by the wait in front of group 2), column pass
                    // at the even slots of 16 .. 62, row pass at the even slots of 72 .. 118, the twelve stores at slots 120 .. 131
                    if constexpr (sl < 12 && (sl & 3) >= 2) t_load(std::integral_constant<int, ((sl / 4) * 2 + (sl & 1))>{}, xaddr);
                    if constexpr (sl >= 16 && sl < 64 && (sl & 1) == 0)
                        t_col(std::integral_constant<int, (((sl - 16) / 2) / 4)>{}, std::integral_constant<int, (((sl - 16) / 2) % 4)>{});
                    if constexpr (sl >= 72 && sl < 120 && (sl & 1) == 0)
                        t_row(std::integral_constant<int, (((sl - 72) / 2) / 4)>{}, std::integral_constant<int, (((sl - 72) / 2) % 4)>{});
                    if constexpr (sl >= 120 && sl < 132) t_store(std::integral_constant<int, sl - 120>{}, vaddr);
                }
                F4_FENCE();
            });
        });
        F4_FENCE();
    };
    {
        int s = 0;
        for (int t = 0; t < ntile; ++t) {
            // the accumulators are (re)defined HERE, outside the stage loop, and die in the epilogue: one plain loop-carried live range
            // each.  (Zeroing them inside a conditional epilogue in a flat stage loop made hipcc shuffle all 256 AGPRs through scratch.)
#pragma unroll
            for (int i = 0; i < 32; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accv[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < S; ++c, ++s) {
                run_stage(s);
                if (s + 2 < total) x_advance();                   // the DMA of this stage fetched X(s + 2): on to X(s + 3)
                wp_cur = wp_nxt;
                if (s + 2 < total) u_advance();                   // wp_nxt: slab of stage s + 2
            }
            if constexpr (!(DCVIC_W44_DBG & 16)) tile_epilogue(c_cotile, c_n, c_oy0, c_ox0, c_par);
            c_b += J; c_par ^= 1;
            if (c_b < xe) {
                decode(c_b, c_cotile, c_n, c_oy0, c_ox0);
                stage_bias(c_b, c_par);
            }
        }
    }
#undef F4_FENCE
#undef F4_WAIT_LDS
}

extern "C" size_t dcvic_wino44_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)((Cout + F4_CO - 1) / F4_CO) * ((Cin + F4_KC - 1) / F4_KC) * F4_US * sizeof(float);
}

extern "C" int dcvic_wino44_pack_f32(const float* w, float* packed, int Cin, int Cout, void* stream) {
    DCVIC_CHECK_ARG(w && packed && Cin > 0 && Cout > 0, "wino44_pack: bad argument");
    const int n_chunks = (Cin + F4_KC - 1) / F4_KC;
    const long long total = (long long)((Cout + F4_CO - 1) / F4_CO) * n_chunks * F4_US;
    wino44_pack_kernel<<<dcvic_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, packed, Cin, Cout, n_chunks, total);
    DCVIC_CHECK_LAUNCH("wino44_pack");
    return DCVIC_OK;
}

extern "C" int dcvic_wino44_stats_tiles(int H, int W) { return H > 0 && W > 0 ? ((H + F4_TH - 1) / F4_TH) * ((W + F4_TW - 1) / F4_TW) : 0; }

static int wino44_launch(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, float* gn_part, void* stream);

extern "C" int dcvic_conv3x3_wino44_f32(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, void* stream) {
    return wino44_launch(Cin, Cout, packed, io, nullptr, stream);
}

extern "C" int dcvic_conv3x3_wino44_stats_f32(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, float* gn_part, void* stream) {
    DCVIC_CHECK_ARG(gn_part, "conv3x3_wino44_stats: null statistics buffer");
    return wino44_launch(Cin, Cout, packed, io, gn_part, stream);
}

static int wino44_launch(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, float* gn_part, void* stream) {
    DCVIC_CHECK_ARG(packed && io && io->out && Cin > 0 && Cout > 0, "conv3x3_wino44: null pointer");
    DCVIC_CHECK_ARG(io->n_src >= 1 && io->n_src <= DCVIC_MAX_SRC, "conv3x3_wino44: n_src %d", io->n_src);
    int csum = 0;
    for (int i = 0; i < io->n_src; ++i) {
        DCVIC_CHECK_ARG(io->src[i].ptr && io->src[i].C > 0 && io->src[i].C % F4_KC == 0, "conv3x3_wino44: source %d needs a multiple of 8 channels", i);
        DCVIC_CHECK_ARG(io->src[i].batch_stride >= (long long)io->src[i].C * io->H * io->W, "conv3x3_wino44: source %d batch stride too small", i);
        DCVIC_CHECK_ARG((reinterpret_cast<uintptr_t>(io->src[i].ptr) & 15) == 0 && (io->src[i].batch_stride & 3) == 0,
                        "conv3x3_wino44: source %d must be 16-byte aligned (16-byte LDS-DMA segments)", i);
        csum += io->src[i].C;
    }
    DCVIC_CHECK_ARG(csum == Cin, "conv3x3_wino44: sources carry %d channels, layer expects %d", csum, Cin);
    DCVIC_CHECK_ARG(io->N > 0 && io->H > 0 && io->W > 0, "conv3x3_wino44: bad sizes");
    DCVIC_CHECK_ARG(io->Hout == io->H && io->Wout == io->W && io->Hfull == io->H && io->Wfull == io->W && io->osy == 1 && io->osx == 1 &&
                    io->ooy == 0 && io->oox == 0, "conv3x3_wino44: stride-1 pad-1 geometry only");
    DCVIC_CHECK_ARG((io->W & 3) == 0, "conv3x3_wino44: width must be a multiple of 4");
    DCVIC_CHECK_ARG(!io->aff_scale && !io->aff_shift && !io->init, "conv3x3_wino44: affine / init epilogues are not supported");
    DCVIC_CHECK_ARG(io->act == DCVIC_ACT_NONE || io->act == DCVIC_ACT_RELU || io->act == DCVIC_ACT_LRELU02,
                    "conv3x3_wino44: activation %d not supported (none / ReLU / LeakyReLU(0.2) only)", io->act);
    DCVIC_CHECK_ARG((long long)io->H * io->W * F4_KC < (1ll << 31), "conv3x3_wino44: plane too large");
    DCVIC_CHECK_ARG(io->out_batch_stride >= (long long)Cout * io->H * io->W && (io->out_batch_stride & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(io->out) & 15) == 0, "conv3x3_wino44: output view must be 16-byte aligned");
    DCVIC_CHECK_ARG(!io->res || (io->res_batch_stride >= (long long)Cout * io->H * io->W && (io->res_batch_stride & 3) == 0 &&
                                 (reinterpret_cast<uintptr_t>(io->res) & 15) == 0), "conv3x3_wino44: residual view must be 16-byte aligned");
    ConvKArgs K;
    memset(&K, 0, sizeof(K));
    K.Cin = Cin; K.Cout = Cout; K.T = 9; K.stride = 1;
    K.N = io->N; K.H = io->H; K.W = io->W; K.Hout = io->H; K.Wout = io->W; K.Hfull = io->H; K.Wfull = io->W;
    K.osy = K.osx = 1;
    for (int i = 0; i < DCVIC_MAX_SRC; ++i) {
        if (i < io->n_src) { K.src[i] = io->src[i].ptr; K.srcC[i] = io->src[i].C; K.src_bs[i] = io->src[i].batch_stride; }
        else { K.src[i] = io->src[0].ptr; K.srcC[i] = 1 << 30; K.src_bs[i] = 0; }
    }
    K.out = io->out; K.out_bs = io->out_batch_stride; K.bias = io->bias; K.act = io->act;
    K.res = io->res; K.res_bs = io->res_batch_stride;
    K.wp = packed;
    K.gn_part = gn_part;
    K.n_chunks = (Cin + F4_KC - 1) / F4_KC;
    K.n_cotiles = (Cout + F4_CO - 1) / F4_CO;
    K.tiles_y = (io->H + F4_TH - 1) / F4_TH;
    K.tiles_x = (io->W + F4_TW - 1) / F4_TW;
    const long long blocks = (long long)io->N * K.tiles_y * K.tiles_x * K.n_cotiles;
    DCVIC_CHECK_ARG(blocks < (1ll << 31), "conv3x3_wino44: grid too large");
    K.nblocks = (int)blocks;
    static std::atomic<unsigned> attr_mask{0};
    if (DcvicAttrOnce once_{attr_mask})
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino44_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // persistent grid: one workgroup per CU, a multiple of the 8 XCDs; each walks its share of the tiles
    int grid = (dcvic_num_cu() / NXCD) * NXCD;
    if (grid < NXCD) grid = NXCD;
    if ((long long)grid > blocks) grid = (int)((blocks + NXCD - 1) / NXCD) * NXCD;
    conv3x3_wino44_kernel<<<grid, F4_THREADS, F4_LDS_FLOATS * sizeof(float), (hipStream_t)stream>>>(K);
    DCVIC_CHECK_LAUNCH("conv3x3_wino44");
    return DCVIC_OK;
}
