// wino.hip -- 3x3 / stride-1 / pad-1 convolution as Winograd F(2x2, 3x3) on fp32 MFMA, gfx950.
//
// Replaces the same torch.nn.Conv2d(k3, s1, p1) layers as conv3x3.hip, but only where the caller opted in
// (dc_vic_amd.layers.allow_winograd): the frozen VQGAN decoder + SFT fusion after the estimator's argmax and the VQGAN encoder
// (ldm/modules/diffusionmodules/model.py:82-141, 368-568; src/models/layer/codeformer_layers.py:20-67;
// src/models/subnet/vq_fusion_module.py:78-126).  Hyper-decoder and CHARM (whose results encoder and decoder must reproduce
// bit for bit), both ELIC networks and the estimator stay on the direct kernels' layer-defined fmaf order.  Winograd
// re-associates: Y = A^T [ sum_ci (G g G^T) . (B^T d B) ] A executes 16 multiplies per 2x2 outputs and input channel instead of
// 36 (4/9 of the MFMA work) and differs from the direct sum at the 1e-6 relative level (it is closer to an fp64 convolution:
// tests/test_gpu_kernels.py::test_wino_*).  Further down: conv3x3_wino_ups_kernel, the 9-position structured form for
// nearest-x2 upsample + conv3x3.
//
// One PERSISTENT workgroup per CU = 512 threads = 8 waves (2 per SIMD); a tile is 64 output channels x (8 rows x 32 columns)
// = 64 tiles of 2x2, all stages of all its tiles one continuous stream.
//   * a pipeline stage is 8 input channels.  Per stage the raw input patch (8 ch x (10 rows x ten 16-byte segments + 2 of padding),
//     zero padded through a zero source) and the pre-transformed weights U (16 positions x 8 ch x 64 co = 32 KiB, packed by
//     wino_pack_kernel in the exact LDS image) arrive by LDS-DMA (`global_load_lds_dwordx4`), two stages / one stage ahead;
//   * every thread transforms ONE (channel, tile) 4x4 patch per stage (12 conflict-free ds_read_b64, 32 adds, 8 ds_write2st64_b32) into the
//     V image of the NEXT stage, its instructions placed behind the MFMAs of the current stage;
//   * wave w = (co group cg = w % 4 of 16 channels, tile half th = w / 4 of 32 tiles) owns ALL 16 Winograd positions of its
//     16 x 32 block: per position two `v_mfma_f32_16x16x4_f32` accumulators (128 accumulator registers per lane), fed by
//     ONE ds_read_b128 per position pair (U: two positions x both k-steps) and ONE ds_read_b128 per position (V: two
//     16-tile blocks x both k-steps) -- U pair-slab word ((cg*4 + k)*16 + m)*4 + pq*2 + ks, V slab word
//     ((th*4 + k)*16 + n)*4 + blk*2 + ks hold channel 4ks + k;
//   * since a lane then holds the same (co, tile) element of all 16 positions, the output transform A^T M A runs in
//     registers: no LDS exchange and no barrier after the last stage; bias -> act -> (+res) -> 16-byte stores (DPP row swap
//     between horizontally adjacent tiles).
// LDS: 2 x 13 KiB raw patch + 2 x 32 KiB U + 2 x 32 KiB V + two bias rows = 154.5 KiB: one workgroup per CU.
// Deterministic and batch-invariant: per position the reduction runs over chunks ascending, then k-steps ascending inside
// the MFMA's ordered fmaf chain; the tiling never depends on N.  Build log, measurements and the timing experiments behind the
// DBG / RS template paths: profiles/r2_wino_experiments.md.
#include "conv_common.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ float dcvic_wino_zero[16];   // zero-initialised: source of padded lanes

#define WN_TH 8
#define WN_TW 32
#define WN_PW 40          // LDS row: columns ox0 - 4 .. ox0 + 35 as ten 16-byte segments; the patch's 34 columns sit at 3 .. 36
#define WN_PLANE 408       // 102 segments per channel plane: 100 of data + 2 of padding, so that FOUR planes are 32 (mod 64) banks apart (see t_load)
#define WN_PSEGS 102
#define WN_SEGS 816        // float4 segments of a stage: 8 ch x (10 rows x 10 + 2)
#define WN_CO 64
#define WN_THREADS 512
#define WN_XSLOTS 2
#define WN_XS 3328         // floats: slot 1 is issued by waves 0..4 only, its idle lanes write zeros behind the patch
#define WN_US 8192
#define WN_VS 8192
#define WN_OFF_U (2 * WN_XS)
#define WN_OFF_V (WN_OFF_U + 2 * WN_US)
#define WN_OFF_BIAS (WN_OFF_V + 2 * WN_VS)
#define WN_LDS_FLOATS (WN_OFF_BIAS + 2 * WN_CO)

// U = G g G^T of one (co, ci) 3x3 kernel, position p = 4a + b
__device__ __forceinline__ double wino_u(const float* g, int a, int b) {
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) s += G[a][r] * (double)g[r * 3 + c] * G[b][c];
    return s;
}

// packed[cotile][chunk][pair 8][cg 4][k 4][m 16][pq 2][ks 2]  <-  w[Cout][Cin][3][3]   (fp64 transform, rounded once);
// position p = 2 pair + pq, channel 4 ks + k, output channel 16 cg + m
__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int n_chunks, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long long r = i;
    const int ks = r & 1; r >>= 1;
    const int pq = r & 1; r >>= 1;
    const int m = r & 15; r >>= 4;
    const int k = r & 3; r >>= 2;
    const int cg = r & 3; r >>= 2;
    const int p = 2 * (int)(r & 7) + pq; r >>= 3;
    const int chunk = (int)(r % n_chunks);
    const int cotile = (int)(r / n_chunks);
    const int co = cotile * WN_CO + cg * 16 + m, ci = chunk * KC + 4 * ks + k;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = (float)wino_u(w + ((long long)co * Cin + ci) * 9, p >> 2, p & 3);
    wp[i] = v;
}

// DBG: timing experiments with WRONG results, instantiated only in the diagnostic build (-DDCVIC_WINO_EXPERIMENTS,
// tools/build_wino_experiments.sh -> tools/libdcvic_wino_exp.so, never in libdcvic_hip.so), selected by DCVIC_WINO_DEBUG=16*DBG: 1 no stage barrier, 2 no transform, 4 no DMA, 8 no operand waits,
// 16 no X DMA, 32 no U DMA, 64 no vmcnt wait in front of the stage barrier
#ifdef DCVIC_WINO_EXPERIMENTS
#define WN_DBG_ARG(bit) (K.TG & (bit))     // 1: every X load from the cached zero source, 2: the same U slab every stage
#else
#define WN_DBG_ARG(bit) false
#endif
template <int DBG>
__global__ __launch_bounds__(WN_THREADS, 2) void conv3x3_wino_kernel(const ConvKArgs K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7

    const long long HW = (long long)K.H * K.W;
    const int S = K.n_chunks;                                     // stages (8-channel chunks) per tile
    const long long x_stride = (long long)KC * HW;                // floats between two stages of one source

    // ---- PERSISTENT workgroup: XCD x = blockIdx.x % 8 owns the contiguous range [xs, xe) of tile indices (cotile fastest, so
    // the workgroups of one L2 share input patches and weight slabs); slot j = blockIdx.x / 8 takes tiles xs + j, xs + j + J, ...
    // All stages of all its tiles form ONE stream: the DMA of the next tile's first stages is in flight during the last stages
    // and the (register-only) epilogue of the current tile, so nothing drains at a tile boundary.
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
    auto decode = [&](int b, int& cotile, int& n, int& oy0, int& ox0) __attribute__((always_inline)) {
        cotile = b % K.n_cotiles; b /= K.n_cotiles;
        const int tile_x = b % K.tiles_x; b /= K.tiles_x;
        const int tile_y = b % K.tiles_y; b /= K.tiles_y;
        n = b; oy0 = tile_y * WN_TH; ox0 = tile_x * WN_TW;
    };

    // ---- raw-patch DMA: element e = tid + s*512 of [8 ch][10][34]; running pointers, advanced per stage, re-derived per tile
    const float* xp[WN_XSLOTS];
    int poff[WN_XSLOTS];
    int x_left = 0, x_n = 0, x_b = first, x_next = 0;             // X stream: image, tile index, chunk inside the tile
    auto x_rebase = [&](int c) __attribute__((always_inline)) {                                  // pointers for absolute input channel c of image x_n
        int si = 0;
        if (c >= K.srcC[0]) { c -= K.srcC[0]; si = 1; if (c >= K.srcC[1]) { c -= K.srcC[1]; si = 2; } }
        const float* base = K.src[si] + (long long)x_n * K.src_bs[si] + (long long)c * HW;
#pragma unroll
        for (int s = 0; s < WN_XSLOTS; ++s) xp[s] = (poff[s] >= 0 && !WN_DBG_ARG(1)) ? base + poff[s] : dcvic_wino_zero;
        x_left = K.srcC[si] - c;
    };
    auto x_setup = [&](int b) __attribute__((always_inline)) {
        int cot, oy0, ox0;
        decode(b, cot, x_n, oy0, ox0);
        const int iy0 = oy0 - 1, ix0 = ox0 - 1;
#pragma unroll
        for (int s = 0; s < WN_XSLOTS; ++s) {
            const int e = tid + s * WN_THREADS;                    // float4 segment e of [8 ch][10 rows][10 segments]
            int o = -1;
            if (e < WN_SEGS) {
                const int k = e / WN_PSEGS, r = e - k * WN_PSEGS;  // (r >= 100: the plane's two padding segments, fed from the zero word)
                const int py = r / 10, seg = r - py * 10;
                const int iy = iy0 + py, ix = ix0 - 3 + 4 * seg;   // W % 4 == 0: a segment is entirely inside or outside the row
                if (r < 100 && iy >= 0 && iy < K.H && ix >= 0 && ix < K.W) o = (int)(k * HW) + iy * K.W + ix;
            }
            poff[s] = o;
        }
        x_rebase(0);
    };
    x_setup(first);
    // ---- weight DMA: the stage's 32 KiB slab is already the LDS image; thread moves float4 #(tid + j*512)
    const float* wp0;                                             // the stage's slab (UNIFORM: scalar base + 32-bit lane offset -> saddr form of the DMA)
    const unsigned u_lane = 16u * (unsigned)tid;                  // this thread's first float4, in bytes
    int u_b = first, u_next = 0;                                  // U stream
    auto u_setup = [&](int b) __attribute__((always_inline)) {
        const float* wbase = K.wp + (long long)(b % K.n_cotiles) * S * (long long)WN_US;
        wp0 = wbase;
    };
    u_setup(first);
    // ---- input transform: this thread's (channel, tile) of a stage
    //   wave -> (th, k); lane -> (n, ks, blk);  channel 4ks + k, tile row 2th + blk, tile column n.
    //   The patch's four columns 2n + 3 .. 2n + 6 start on an ODD dword (16-byte DMA segments of a pad-1 convolution), so dword reads
    //   of a 32-lane group can only ever touch the 16 odd (or even) banks: a 2-way conflict on every access (28.5 % of the LDS cycles
    //   of the round-2 build).  They are fetched instead as THREE aligned 8-byte pairs (2n + 2 .. 2n + 7; `ds_read_b64`: 64 banks per
    //   32-lane group): lanes 0..15 of a group cover banks 2 .. 33 and lanes 16..31 hold the channel four planes further
    //   (4 x 408 dwords = 32 mod 64): conflict-free.
    const int t_th = wave >> 2, t_k = wave & 3;
    const int t_n = lane & 15, t_ks = (lane >> 4) & 1, t_blk = lane >> 5;
    const unsigned t_src = 4u * (unsigned)((4 * t_ks + t_k) * WN_PLANE + (2 * (2 * t_th + t_blk)) * WN_PW + 2 * t_n + 2);
    const unsigned t_dst = 4u * (unsigned)(WN_OFF_V + ((t_th * 4 + t_k) * 16 + t_n) * 4 + t_blk * 2 + t_ks);
    // ---- MFMA operands: wave -> (cg = co group, th = tile half)
    const int cg = wave & 3, th = wave >> 2;
    const unsigned op_u = 4u * (unsigned)(WN_OFF_U + cg * 256 + lane * 4);   // this lane's float4 of position pair 0
    const unsigned op_v = 4u * (unsigned)(WN_OFF_V + th * 256 + lane * 4);   // this lane's float4 of position 0

    f32x4 acc[16][2];                                             // [position][16-tile block]
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

    // All LDS traffic of the loop is inline asm with hand-placed `s_waitcnt lgkmcnt(0)`: hipcc guards every LDS access it
    // can see with `s_waitcnt vmcnt(0)` while an LDS-DMA is in flight (it cannot prove the DMA's destination does not alias),
    // which serialises the stage into "DMA latency + transform + MFMA" (measured: 52 % -> MFMA-busy).  Nothing below is
    // visible to it as an LDS access, so the only vmcnt wait is the explicit one in front of the stage barrier.
#define WN_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WN_WAIT_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); WN_FENCE(); } while (0)
    // raw 4x4 patch of the transform.  The asm loads write the very variables t_compute reads after the wait: a copy made
    // between a load and the `s_waitcnt` would read the register before the LDS data has landed.
    f32x2 tp0[4], tp1[4], tp2[4];                                 // row r: dwords (2n+2, 2n+3) | (2n+4, 2n+5) | (2n+6, 2n+7); the patch is the middle four
    float td[4][4];
    auto t_load = [&](auto r_, unsigned xaddr) {                  // row r of the patch: three aligned dword pairs
        constexpr int r = decltype(r_)::value;
        f32x2 &p0 = tp0[r], &p1 = tp1[r], &p2 = tp2[r];
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p0) : "v"(xaddr), "n"(4 * (r * WN_PW)));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p1) : "v"(xaddr), "n"(4 * (r * WN_PW + 2)));
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p2) : "v"(xaddr), "n"(4 * (r * WN_PW + 4)));
    };
    float tv[16];
    auto t_compute = [&](auto c_) {                               // column c of B^T d, then nothing else: rows are finished in t_rows
        constexpr int c = decltype(c_)::value;
        auto col = [&](int r) __attribute__((always_inline)) { return c == 0 ? tp0[r][1] : c == 1 ? tp1[r][0] : c == 2 ? tp1[r][1] : tp2[r][0]; };
        const float d0 = col(0), d1 = col(1), d2 = col(2), d3 = col(3);
        td[0][c] = d0 - d2; td[1][c] = d1 + d2; td[2][c] = d2 - d1; td[3][c] = d1 - d3;
    };
    auto t_rows = [&](auto a_) {                                  // row a of (B^T d) B
        constexpr int a = decltype(a_)::value;
        tv[4 * a + 0] = td[a][0] - td[a][2];
        tv[4 * a + 1] = td[a][1] + td[a][2];
        tv[4 * a + 2] = td[a][2] - td[a][1];
        tv[4 * a + 3] = td[a][1] - td[a][3];
    };
    auto t_store = [&](auto p_, unsigned vaddr) {                 // positions 2p and 2p + 1 (slabs 2 KiB apart = 8 x 64 dwords)
        constexpr int p = decltype(p_)::value;
        const float v0 = tv[2 * p], v1 = tv[2 * p + 1];           // (asm operands inside a generic lambda do not capture)
        asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(vaddr), "v"(v0), "v"(v1), "n"(16 * p), "n"(16 * p + 8) : "memory");
    };
    f32x4 opA[2];                                                 // [set = pair parity]: (position 2j: ks 0, 1 | position 2j + 1: ks 0, 1)
    f32x4 opB[2][2];                                              // [set][position inside the pair]: (blk 0: ks 0, 1 | blk 1: ks 0, 1)
    auto op_load = [&](auto j_, unsigned ua, unsigned va) {       // operands of positions 2j and 2j + 1
        constexpr int j = decltype(j_)::value;
        f32x4 &a = opA[j & 1];
        f32x4 &b0 = opB[j & 1][0], &b1 = opB[j & 1][1];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(ua), "n"(4 * 1024 * j));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b0) : "v"(va), "n"(4 * 512 * (2 * j)));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b1) : "v"(va), "n"(4 * 512 * (2 * j + 1)));
    };
    auto dma_x = [&](auto s_, int buf) {
        constexpr int sl = decltype(s_)::value;
        if (sl == 0 || wave < 5)                                  // (wave-uniform: segments 512 .. 799 live in waves 0 .. 4 of slot 1)
            __builtin_amdgcn_global_load_lds(reinterpret_cast<const float4*>(xp[sl]), (lds_ptr_t)(smem + buf * WN_XS + (wave * 64 + sl * WN_THREADS) * 4), 16, 0, 0);
    };
    auto dma_u = [&](auto j_, int buf) {
        constexpr int j = decltype(j_)::value;
        {   // saddr form by hand (hipcc re-materialises 64-bit per-lane addresses inside the loop): scalar base + 32-bit lane offset
            // instead of a 64-bit address per lane -- 2.38 -> 2.32 ms on the 256 -> 256 @ 128^2 x 32 layer
            const unsigned voff = u_lane + 16u * WN_THREADS * j;
            const unsigned long long sb = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(reinterpret_cast<unsigned long long>(wp0) & 0xffffffffull)) & 0xffffffffull
                                        | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(reinterpret_cast<unsigned long long>(wp0) >> 32)) << 32);
            const unsigned lds = (unsigned)(4 * (WN_OFF_U + buf * WN_US + (wave * 64 + j * WN_THREADS) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sb), "s"(lds) : "memory", "m0");
        }
    };
    // RS (DBG bit 256) = REGISTER STAGING instead of LDS-DMA: the stage's U slab and X patch are fetched into 24 registers by plain
    // 16-byte global loads one stage earlier and written to LDS with ds_write_b128 (an LDS-DMA piece holds the SIMD's vector issue
    // for 60 - 185 cycles, MI355X_MICROARCH "LDS-DMA piece issue cost"; a load + a store are two short instructions that hide
    // inside MFMA gaps).  Loads are ordinary C++ loads: hipcc places the counted vmcnt waits in front of the stores itself.
    constexpr bool RS = (DBG & 256) != 0;
    f32x4 su[4], sx[2];
    const unsigned st_base = 16u * (unsigned)tid;
    auto ld_u = [&](auto j_) {
        constexpr int j = decltype(j_)::value;
        su[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(wp0) + (size_t)(u_lane + 16u * WN_THREADS * j));
    };
    auto st_u = [&](auto j_, int buf) {
        constexpr int j = decltype(j_)::value;
        const f32x4 v = su[j];
        const unsigned a = st_base + 4u * (unsigned)(WN_OFF_U + buf * WN_US);
        asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(16 * WN_THREADS * j) : "memory");
    };
    auto ld_x = [&](auto s_) {
        constexpr int sl = decltype(s_)::value;
        if (sl == 0 || wave < 5) sx[sl] = *reinterpret_cast<const f32x4*>(xp[sl]);
    };
    auto st_x = [&](auto s_, int buf) {
        constexpr int sl = decltype(s_)::value;
        if (sl == 0 || wave < 5) {
            const f32x4 v = sx[sl];
            const unsigned a = st_base + 4u * (unsigned)(buf * WN_XS);
            asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a), "v"(v), "n"(16 * WN_THREADS * sl) : "memory");
        }
    };
    auto x_advance = [&]() __attribute__((always_inline)) {                                     // after the DMA of an X stage: on to the next stage of the stream
        if (++x_next == S) {
            x_next = 0;
            x_b += J;
            if (x_b < xe) x_setup(x_b);
        } else {
            x_left -= KC;
            if (x_left > 0) {
#pragma unroll
                for (int sl = 0; sl < WN_XSLOTS; ++sl) xp[sl] += (poff[sl] >= 0 && !WN_DBG_ARG(1)) ? x_stride : 0ll;   // (padding lanes stay on the zero word)
            } else {
                x_rebase(x_next * KC);
            }
        }
    };
    auto u_advance = [&]() __attribute__((always_inline)) {
        if (++u_next == S) {
            u_next = 0;
            u_b += J;
            if (u_b < xe) u_setup(u_b);
        } else {
            if (!WN_DBG_ARG(2)) wp0 += WN_US;
        }
    };

    // ---- epilogue of one tile, in registers: lane holds element (co = 16cg + 4(lane/16) + r, tile = (row 2th + blk, column
    // lane%16)) of all 16 positions.  A^T M A, bias -> act -> (+res) -> float2 stores; a block's residuals are requested up front.
    float* const sbias = smem + WN_OFF_BIAS;                      // [2][64], by tile parity
    const int tx = lane & 15, lq = lane >> 4;
    const int act = K.act;
    const bool has_bias = K.bias != nullptr, has_res = K.res != nullptr;
    auto tile_epilogue = [&](int cotile, int n, int oy0, int ox0, int par) __attribute__((always_inline)) {
        // Lanes tx and tx ^ 1 hold the 2x2 outputs of two horizontally adjacent tiles: they swap one row each (DPP quad_perm
        // [1,0,3,2]) so that the even lane owns FOUR consecutive columns of the upper row and the odd lane of the lower row --
        // one 16-byte store (and residual load) per lane and channel instead of two 8-byte ones (VMEM instructions are the
        // expensive part of this kernel's side work).
        const bool odd = tx & 1;
        const int ox = ox0 + 2 * (tx & ~1);
        const bool in_x = ox < K.W;                               // W % 4 == 0: all four columns or none
        const int co0 = cotile * WN_CO + cg * 16 + 4 * lq;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = has_bias ? sbias[par * WN_CO + cg * 16 + 4 * lq + r] : 0.f;
        dcvic_static_for<0, 2>([&](auto blk_) {
            constexpr int blk = decltype(blk_)::value;
            const int oy = oy0 + 2 * (2 * th + blk) + (odd ? 1 : 0);
            const bool live = in_x && oy < K.H;
            const long long pix = (long long)oy * K.W + ox;
            float* const ob = K.out + (long long)n * K.out_bs + pix;
            const float* const rb = has_res ? K.res + (long long)n * K.res_bs + pix : nullptr;
            f32x4 rv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_res && live && co0 + r < K.Cout) rv[r] = *reinterpret_cast<const f32x4*>(rb + (long long)(co0 + r) * HW);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s0[4], s1[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    s0[c] = acc[c][blk][r] + acc[4 + c][blk][r] + acc[8 + c][blk][r];
                    s1[c] = acc[4 + c][blk][r] - acc[8 + c][blk][r] - acc[12 + c][blk][r];
                }
                float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
                float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
                y00 = dcvic_act(y00 + bv[r], act); y01 = dcvic_act(y01 + bv[r], act);
                y10 = dcvic_act(y10 + bv[r], act); y11 = dcvic_act(y11 + bv[r], act);
                const float g0 = odd ? y00 : y10, g1 = odd ? y01 : y11;            // what the neighbour needs from this lane
                const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, g0), 0xB1, 0xF, 0xF, true));
                const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, g1), 0xB1, 0xF, 0xF, true));
                const f32x4 o = odd ? f32x4{n0, n1, y10, y11} : f32x4{y00, y01, n0, n1};
                if (live && co0 + r < K.Cout) *reinterpret_cast<f32x4*>(ob + (long long)(co0 + r) * HW) = o + rv[r];
            }
        });
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    };
    auto stage_bias = [&](int b, int par) __attribute__((always_inline)) {                       // bias row of tile b -> sbias[par] (read >= one barrier later)
        if (tid < WN_CO) sbias[par * WN_CO + tid] = has_bias ? K.bias[min((b % K.n_cotiles) * WN_CO + tid, K.Cout - 1)] : 0.f;
    };

    // ---- pipeline
    int c_b = first, c_chunk = 0, c_par = 0;                      // compute stream: tile, chunk inside it, tile parity
    int c_cotile, c_n, c_oy0, c_ox0;
    decode(first, c_cotile, c_n, c_oy0, c_ox0);
    stage_bias(first, 0);
    if constexpr (RS) {
        // X(0) -> Xr[0], X(1) -> Xr[1], U(0) -> U[0]; U(1) and X(2) stay in the staging registers for stage 0 to store
        dcvic_static_for<0, WN_XSLOTS>([&](auto s_) { ld_x(s_); });
        dcvic_static_for<0, WN_XSLOTS>([&](auto s_) { st_x(s_, 0); });
        x_advance();
        if (total > 1) {
            dcvic_static_for<0, WN_XSLOTS>([&](auto s_) { ld_x(s_); });
            dcvic_static_for<0, WN_XSLOTS>([&](auto s_) { st_x(s_, 1); });
            x_advance();
        }
        dcvic_static_for<0, 4>([&](auto j_) { ld_u(j_); });
        dcvic_static_for<0, 4>([&](auto j_) { st_u(j_, 0); });
        u_advance();
        if (total > 1) { dcvic_static_for<0, 4>([&](auto j_) { ld_u(j_); }); u_advance(); }
        if (total > 2) { dcvic_static_for<0, WN_XSLOTS>([&](auto s_) { ld_x(s_); }); x_advance(); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
        dcvic_static_for<0, WN_XSLOTS>([&](auto s_) { dma_x(s_, 0); });
        x_advance();
        dcvic_static_for<0, 4>([&](auto j_) { dma_u(j_, 0); });
        u_advance();
        if (total > 1) {
            dcvic_static_for<0, WN_XSLOTS>([&](auto s_) { dma_x(s_, 1); });
            x_advance();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    WN_FENCE();
    dcvic_static_for<0, 4>([&](auto r_) { t_load(r_, t_src); });
    WN_WAIT_LDS();
    dcvic_static_for<0, 4>([&](auto c_) { t_compute(c_); });
    dcvic_static_for<0, 4>([&](auto a_) { t_rows(a_); });
    dcvic_static_for<0, 8>([&](auto p_) { t_store(p_, t_dst); });
    WN_WAIT_LDS();
    __syncthreads();
    WN_FENCE();
    op_load(std::integral_constant<int, 0>{}, op_u, op_v);        // first operands of stage 0
    // more1 / more2: a stage g + 1 / g + 2 exists in the stream (compile-time: no branches between the MFMAs)
    auto run_stage = [&](auto more1_, auto more2_, int g) __attribute__((always_inline)) {
        constexpr bool more1 = decltype(more1_)::value, more2 = decltype(more2_)::value;
        const int cur = g & 1, nxt = cur ^ 1;
        const unsigned ua = op_u + (unsigned)(cur * WN_US * 4), va = op_v + (unsigned)(cur * WN_VS * 4);
        const unsigned xaddr = t_src + (unsigned)(nxt * WN_XS * 4), vaddr = t_dst + (unsigned)(nxt * WN_VS * 4);
        // 64 MFMA slots = 8 position pairs x (position, k-step, block).  In front of a pair's eight MFMAs: wait for its operands,
        // then request the next pair's (they have eight MFMAs to arrive); behind each MFMA a piece of the stage's other work
        // issues in its shadow: the DMA of U(g+1) and X(g+2), the transform of X(g+1) into V(g+1).
        // The stage BARRIER sits in front of the LAST pair: by then every operand read of this stage has returned, the
        // transform's stores and this wave's DMA pieces have landed; behind it the first operands of stage g + 1 are requested
        // and arrive under the last pair's MFMAs -- no LDS round trip between two stages' MFMAs (pair 0 of the very first stage
        // is requested by the prologue).
        dcvic_static_for<0, 8>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j < 7) {
                if constexpr (!(DBG & 8)) WN_WAIT_LDS();
                op_load(std::integral_constant<int, j + 1>{}, ua, va);
            } else {
                if constexpr ((DBG & 64) || RS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (RS: nothing lands by DMA)
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                if constexpr (!(DBG & 1)) __syncthreads();
                WN_FENCE();
                if constexpr (more1) op_load(std::integral_constant<int, 0>{}, op_u + (unsigned)(nxt * WN_US * 4), op_v + (unsigned)(nxt * WN_VS * 4));
            }
            WN_FENCE();
            dcvic_static_for<0, 8>([&](auto i_) {
                constexpr int i = decltype(i_)::value, pq = i >> 2, ks = (i >> 1) & 1, blk = i & 1, pp = 2 * j + pq;
                acc[pp][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(opA[j & 1][pq * 2 + ks], opB[j & 1][pq][blk * 2 + ks], acc[pp][blk], 0, 0, 0);
                WN_FENCE();
                // slot sl = 8j + i.  DMA pieces one every 4th slot (all eight waves run this schedule in step: ten pieces in ten
                // consecutive slots put 80 VMEM instructions into the CU's address unit at once and stalled the issuing waves --
                // measured 16 % of the kernel), U first (needed right after the barrier), then X; transform work after them.
                constexpr int sl = 8 * j + i;
                auto fillers = [&]() __attribute__((always_inline)) {
                    if constexpr (RS) {
                        // stage g: store U(g+1) / X(g+2) (loaded during stage g-1), then reload the registers with U(g+2) / X(g+3)
                        if constexpr ((sl & 3) == 1 && sl < 16) {
                            if constexpr (more1) st_u(std::integral_constant<int, sl / 4>{}, nxt);
                            if constexpr (more2) ld_u(std::integral_constant<int, sl / 4>{});
                        }
                        if constexpr ((sl & 3) == 1 && sl >= 16 && sl < 24) {
                            if constexpr (more2) st_x(std::integral_constant<int, sl / 4 - 4>{}, cur);
                            if constexpr (more2) { if (g + 3 < total) ld_x(std::integral_constant<int, sl / 4 - 4>{}); }
                        }
                    } else {
                    if constexpr (!(DBG & 36) && more1 && (sl & 3) == 1 && sl < 16) dma_u(std::integral_constant<int, sl / 4>{}, nxt);
                    if constexpr (!(DBG & 20) && more2 && (sl & 3) == 1 && sl >= 16 && sl < 24) dma_x(std::integral_constant<int, sl / 4 - 4>{}, cur);
                    }
                    if constexpr (!(DBG & 2) && more1 && j == 2 && (i & 3) >= 2) t_load(std::integral_constant<int, (i >> 2) * 2 + (i & 1)>{}, xaddr);
                    if constexpr (!(DBG & 2) && more1 && j == 3 && (i & 3) >= 2) t_compute(std::integral_constant<int, (i >> 2) * 2 + (i & 1)>{});
                    if constexpr (!(DBG & 2) && more1 && j == 4 && (i & 3) >= 2) t_rows(std::integral_constant<int, (i >> 2) * 2 + (i & 1)>{});
                    if constexpr (!(DBG & 2) && more1 && (j == 5 || j == 6) && (i & 1)) t_store(std::integral_constant<int, 4 * (j - 5) + (i >> 1)>{}, vaddr);
                };
                if constexpr (DBG & 128) {          // experiment: waves 4..7 do the side work of a SIMD twice, waves 0..3 none
                    if (wave >= 4) { fillers(); WN_FENCE(); fillers(); }
                } else {
                    fillers();
                }
                WN_FENCE();
            });
        });
        if constexpr (RS) {
            if constexpr (more2) { u_advance(); if (g + 3 < total) x_advance(); }
        } else {
            if constexpr (more2) x_advance();
            if constexpr (more1) u_advance();
        }
        if (++c_chunk == S) {                                     // the tile is complete: write it out, move the compute stream on
            tile_epilogue(c_cotile, c_n, c_oy0, c_ox0, c_par);
            c_chunk = 0; c_b += J; c_par ^= 1;
            if (c_b < xe) {
                decode(c_b, c_cotile, c_n, c_oy0, c_ox0);
                stage_bias(c_b, c_par);
            }
        }
        WN_FENCE();
    };
    {
        int g = 0;
        for (; g + 2 < total; ++g) run_stage(std::true_type{}, std::true_type{}, g);
        if (g + 1 < total) { run_stage(std::true_type{}, std::false_type{}, g); ++g; }
        run_stage(std::false_type{}, std::false_type{}, g);
    }
#undef WN_FENCE
#undef WN_WAIT_LDS
}

// ------------------------------------------------------------------------------------------------------------------------
// Nearest-x2 upsample + Conv2d(k3, s1, p1) (ldm Upsample, model.py:42-57) as a STRUCTURED Winograd F(2x2, 3x3).
// A 2x2 output tile aligned to the 2x2 blocks of the upsampled image sees input rows [a, b, b, c] (three low-resolution rows),
// so B^T d = [a - b, 2b, 0, b - c]: the third transformed row (and column) is identically zero and only 9 of the 16 Winograd
// positions carry data -- 9 multiplies per 4 outputs and input channel (2.25 per output; the four 2x2 sub-pixel phase
// convolutions execute 4, the plain 3x3 sum 9).  The factors 2 move into the weights (U'[a][b] = U[A(a)][A(b)] * 2^[a=1] * 2^[b=1],
// A = {0, 1, 3}: exact in fp32), which leaves V' = R d R^T with R = [[1,-1,0],[0,1,0],[0,1,-1]]: twelve subtractions per patch.
// Same workgroup shape, persistent stage stream, LDS images, barrier protocol and register-only epilogue as conv3x3_wino_kernel;
// the tile grid runs over the OUTPUT (8 x 32 pixels = 4 x 16 low-resolution pixels), the raw patch is 8 ch x 6 rows x six 16-byte
// segments of the LOW-resolution map, a stage has 36 MFMAs per wave (positions 0..8 as pairs (0,1) .. (6,7), (8)).
#define WU_NPOS 9
#define WU_PW 24
#define WU_PLANE 144
#define WU_SEGS 288
#define WU_US 5120          // floats of a stage's U slab: 5 pair slabs x 1024 (the tenth position is zero padding)

// packed[cotile][chunk][pair 5][cg 4][k 4][m 16][pq 2][ks 2]  <-  w[Cout][Cin][3][3]
__global__ void wino_ups_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int n_chunks, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    long long r = i;
    const int ks = r & 1; r >>= 1;
    const int pq = r & 1; r >>= 1;
    const int m = r & 15; r >>= 4;
    const int k = r & 3; r >>= 2;
    const int cg = r & 3; r >>= 2;
    const int pair = (int)(r % 5); r /= 5;
    const int chunk = (int)(r % n_chunks);
    const int cotile = (int)(r / n_chunks);
    const int p9 = 2 * pair + pq;
    const int co = cotile * WN_CO + cg * 16 + m, ci = chunk * KC + 4 * ks + k;
    float v = 0.f;
    if (p9 < WU_NPOS && co < Cout && ci < Cin) {
        const int a = p9 / 3, b = p9 - 3 * a;
        const int A[3] = {0, 1, 3};
        v = (float)(wino_u(w + ((long long)co * Cin + ci) * 9, A[a], A[b]) * (a == 1 ? 2.0 : 1.0) * (b == 1 ? 2.0 : 1.0));
    }
    wp[i] = v;
}

__global__ __launch_bounds__(WN_THREADS, 2) void conv3x3_wino_ups_kernel(const ConvKArgs K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long HW = (long long)K.H * K.W;                    // low-resolution input plane
    const long long HWo = (long long)K.Hfull * K.Wfull;           // output plane
    const int S = K.n_chunks;
    const long long x_stride = (long long)KC * HW;
    int xe, first;
    const int J = (int)gridDim.x / NXCD;
    {
        const int nb = K.nblocks, q = nb / NXCD, r = nb % NXCD, x = (int)blockIdx.x % NXCD;
        const int xs = x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        xe = xs + (x < r ? q + 1 : q);
        first = xs + (int)blockIdx.x / NXCD;
    }
    if (first >= xe) return;
    const int total = ((xe - first + J - 1) / J) * S;
    auto decode = [&](int b, int& cotile, int& n, int& oy0, int& ox0) __attribute__((always_inline)) {
        cotile = b % K.n_cotiles; b /= K.n_cotiles;
        const int tile_x = b % K.tiles_x; b /= K.tiles_x;
        const int tile_y = b % K.tiles_y; b /= K.tiles_y;
        n = b; oy0 = tile_y * WN_TH; ox0 = tile_x * WN_TW;          // OUTPUT coordinates
    };
    // ---- raw low-resolution patch: segment e = tid of [8 ch][6 rows][6 segments] (waves 0..4)
    const float* xp;
    int poff;
    int x_left = 0, x_n = 0, x_b = first, x_next = 0;
    auto x_rebase = [&](int c) __attribute__((always_inline)) {
        int si = 0;
        if (c >= K.srcC[0]) { c -= K.srcC[0]; si = 1; if (c >= K.srcC[1]) { c -= K.srcC[1]; si = 2; } }
        const float* base = K.src[si] + (long long)x_n * K.src_bs[si] + (long long)c * HW;
        xp = poff >= 0 ? base + poff : dcvic_wino_zero;
        x_left = K.srcC[si] - c;
    };
    auto x_setup = [&](int b) __attribute__((always_inline)) {
        int cot, oy0, ox0;
        decode(b, cot, x_n, oy0, ox0);
        int o = -1;
        if (tid < WU_SEGS) {
            const int k = tid / 36, r = tid - k * 36;
            const int py = r / 6, seg = r - py * 6;
            const int iy = oy0 / 2 - 1 + py, ix = ox0 / 2 - 4 + 4 * seg;
            if (iy >= 0 && iy < K.H && ix >= 0 && ix < K.W) o = (int)(k * HW) + iy * K.W + ix;
        }
        poff = o;
        x_rebase(0);
    };
    x_setup(first);
    const float* wp0;
    int u_b = first, u_next = 0;
    auto u_setup = [&](int b) __attribute__((always_inline)) {
        wp0 = K.wp + (long long)(b % K.n_cotiles) * S * (long long)WU_US;      // uniform: the DMA uses the saddr form
    };
    u_setup(first);
    // ---- input transform: wave -> (th, k); lane -> (n, blk, ks); channel 4ks + k, low-resolution pixel (row 2th + blk, column n)
    const int t_th = wave >> 2, t_k = wave & 3;
    const int t_n = lane & 15, t_blk = (lane >> 4) & 1, t_ks = lane >> 5;
    const unsigned t_src = 4u * (unsigned)((4 * t_ks + t_k) * WU_PLANE + (2 * t_th + t_blk) * WU_PW + t_n + 3);
    const unsigned t_dst = 4u * (unsigned)(WN_OFF_V + ((t_th * 4 + t_k) * 16 + t_n) * 4 + t_blk * 2 + t_ks);
    const int cg = wave & 3, th = wave >> 2;
    const unsigned op_u = 4u * (unsigned)(WN_OFF_U + cg * 256 + lane * 4);
    const unsigned op_v = 4u * (unsigned)(WN_OFF_V + th * 256 + lane * 4);
    f32x4 acc[WU_NPOS][2];
#pragma unroll
    for (int i = 0; i < WU_NPOS; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
#define WN_FENCE() __builtin_amdgcn_sched_barrier(0)
#define WN_WAIT_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); WN_FENCE(); } while (0)
    // (the asm loads write the very variables the transform reads after the wait: a copy made between the load and the
    // `s_waitcnt` would read the register before the LDS data has landed)
    f32x2 tlo[3];
    float tc2[3];
    float td[3][3], tv[10];
    auto t_load = [&](auto r_, unsigned xaddr) {
        constexpr int r = decltype(r_)::value;
        f32x2& lo = tlo[r];
        float& c2 = tc2[r];
        asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(lo) : "v"(xaddr), "n"(r * WU_PW), "n"(r * WU_PW + 1));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(c2) : "v"(xaddr), "n"(4 * (r * WU_PW + 2)));
    };
    auto t_cols = [&]() {                                         // R d: rows [d0 - d1, d1, d1 - d2]
        const float d0[3] = {tlo[0][0], tlo[0][1], tc2[0]}, d1[3] = {tlo[1][0], tlo[1][1], tc2[1]}, d2[3] = {tlo[2][0], tlo[2][1], tc2[2]};
#pragma unroll
        for (int c = 0; c < 3; ++c) { td[0][c] = d0[c] - d1[c]; td[1][c] = d1[c]; td[2][c] = d1[c] - d2[c]; }
    };
    auto t_rows = [&]() {                                         // (R d) R^T
#pragma unroll
        for (int a = 0; a < 3; ++a) { tv[3 * a] = td[a][0] - td[a][1]; tv[3 * a + 1] = td[a][1]; tv[3 * a + 2] = td[a][1] - td[a][2]; }
        tv[9] = 0.f;
    };
    auto t_store = [&](auto p_, unsigned vaddr) {                 // positions 2p, 2p + 1; p = 4: position 8 alone
        constexpr int p = decltype(p_)::value;
        const float v0 = tv[2 * p], v1 = tv[2 * p + 1];
        if constexpr (p < 4) asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" :: "v"(vaddr), "v"(v0), "v"(v1), "n"(16 * p), "n"(16 * p + 8) : "memory");
        else asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(vaddr), "v"(v0), "n"(4 * 512 * 8) : "memory");
    };
    f32x4 opA[2];
    f32x4 opB[2][2];
    // operand register set of pair j in a stage of parity PAR: (j + PAR) & 1.  Five pairs per stage: the last pair of a stage and
    // the first pair of the next one (requested behind the barrier, before the last pair's MFMAs) must not share a set, so the
    // sets swap roles every stage and the stage body is instantiated for both parities.
    auto op_load = [&](auto j_, auto set_, unsigned ua, unsigned va) {
        constexpr int j = decltype(j_)::value, set = decltype(set_)::value;
        f32x4 &a = opA[set];
        f32x4 &b0 = opB[set][0], &b1 = opB[set][1];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a) : "v"(ua), "n"(4 * 1024 * j));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b0) : "v"(va), "n"(4 * 512 * (2 * j)));
        if constexpr (2 * j + 1 < WU_NPOS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b1) : "v"(va), "n"(4 * 512 * (2 * j + 1)));
    };
    auto dma_x = [&](int buf) {
        if (wave < 5) __builtin_amdgcn_global_load_lds(reinterpret_cast<const float4*>(xp), (lds_ptr_t)(smem + buf * WN_XS + wave * 64 * 4), 16, 0, 0);
    };
    auto dma_u = [&](auto j_, int buf) {
        constexpr int j = decltype(j_)::value;
        if (j < 2 || wave < 4) {
            const unsigned voff = 16u * (unsigned)tid + 16u * WN_THREADS * j;
            const unsigned long long sb = (unsigned long long)__builtin_amdgcn_readfirstlane((int)(reinterpret_cast<unsigned long long>(wp0) & 0xffffffffull)) & 0xffffffffull
                                        | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(reinterpret_cast<unsigned long long>(wp0) >> 32)) << 32);
            const unsigned lds = (unsigned)(4 * (WN_OFF_U + buf * WN_US + (wave * 64 + j * WN_THREADS) * 4));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sb), "s"(lds) : "memory", "m0");
        }
    };
    auto x_advance = [&]() __attribute__((always_inline)) {
        if (++x_next == S) {
            x_next = 0; x_b += J;
            if (x_b < xe) x_setup(x_b);
        } else {
            x_left -= KC;
            if (x_left > 0) xp += poff >= 0 ? x_stride : 0ll;
            else x_rebase(x_next * KC);
        }
    };
    auto u_advance = [&]() __attribute__((always_inline)) {
        if (++u_next == S) { u_next = 0; u_b += J; if (u_b < xe) u_setup(u_b); }
        else wp0 += WU_US;
    };
    float* const sbias = smem + WN_OFF_BIAS;
    const int tx = lane & 15, lq = lane >> 4;
    const int act = K.act;
    const bool has_bias = K.bias != nullptr, has_res = K.res != nullptr;
    auto tile_epilogue = [&](int cotile, int n, int oy0, int ox0, int par) __attribute__((always_inline)) {
        const bool odd = tx & 1;
        const int ox = ox0 + 2 * (tx & ~1);
        const bool in_x = ox < K.Wfull;
        const int co0 = cotile * WN_CO + cg * 16 + 4 * lq;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = has_bias ? sbias[par * WN_CO + cg * 16 + 4 * lq + r] : 0.f;
        dcvic_static_for<0, 2>([&](auto blk_) {
            constexpr int blk = decltype(blk_)::value;
            const int oy = oy0 + 2 * (2 * th + blk) + (odd ? 1 : 0);
            const bool live = in_x && oy < K.Hfull;
            const long long pix = (long long)oy * K.Wfull + ox;
            float* const ob = K.out + (long long)n * K.out_bs + pix;
            const float* const rb = has_res ? K.res + (long long)n * K.res_bs + pix : nullptr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f32x4 rv = f32x4{0.f, 0.f, 0.f, 0.f};
                if (has_res && live && co0 + r < K.Cout) rv = *reinterpret_cast<const f32x4*>(rb + (long long)(co0 + r) * HWo);
                float s0[3], s1[3];                               // A^T M with the third Winograd row / column absent
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    s0[c] = acc[c][blk][r] + acc[3 + c][blk][r];
                    s1[c] = acc[3 + c][blk][r] - acc[6 + c][blk][r];
                }
                float y00 = s0[0] + s0[1], y01 = s0[1] - s0[2];
                float y10 = s1[0] + s1[1], y11 = s1[1] - s1[2];
                y00 = dcvic_act(y00 + bv[r], act); y01 = dcvic_act(y01 + bv[r], act);
                y10 = dcvic_act(y10 + bv[r], act); y11 = dcvic_act(y11 + bv[r], act);
                const float g0 = odd ? y00 : y10, g1 = odd ? y01 : y11;
                const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, g0), 0xB1, 0xF, 0xF, true));
                const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, g1), 0xB1, 0xF, 0xF, true));
                const f32x4 o = odd ? f32x4{n0, n1, y10, y11} : f32x4{y00, y01, n0, n1};
                if (live && co0 + r < K.Cout) *reinterpret_cast<f32x4*>(ob + (long long)(co0 + r) * HWo) = o + rv;
            }
        });
#pragma unroll
        for (int i = 0; i < WU_NPOS; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    };
    auto stage_bias = [&](int b, int par) __attribute__((always_inline)) {
        if (tid < WN_CO) sbias[par * WN_CO + tid] = has_bias ? K.bias[min((b % K.n_cotiles) * WN_CO + tid, K.Cout - 1)] : 0.f;
    };
    int c_b = first, c_chunk = 0, c_par = 0;
    int c_cotile, c_n, c_oy0, c_ox0;
    decode(first, c_cotile, c_n, c_oy0, c_ox0);
    stage_bias(first, 0);
    dma_x(0); x_advance();
    dcvic_static_for<0, 3>([&](auto j_) { dma_u(j_, 0); });
    u_advance();
    if (total > 1) { dma_x(1); x_advance(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    WN_FENCE();
    dcvic_static_for<0, 3>([&](auto r_) { t_load(r_, t_src); });
    WN_WAIT_LDS();
    t_cols(); t_rows();
    dcvic_static_for<0, 5>([&](auto p_) { t_store(p_, t_dst); });
    WN_WAIT_LDS();
    __syncthreads();
    WN_FENCE();
    op_load(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, op_u, op_v);
    auto run_stage = [&](auto more1_, auto more2_, auto par_, int g) __attribute__((always_inline)) {
        constexpr bool more1 = decltype(more1_)::value, more2 = decltype(more2_)::value;
        constexpr int PAR = decltype(par_)::value;
        const int cur = g & 1, nxt = cur ^ 1;
        const unsigned ua = op_u + (unsigned)(cur * WN_US * 4), va = op_v + (unsigned)(cur * WN_VS * 4);
        const unsigned xaddr = t_src + (unsigned)(nxt * WN_XS * 4), vaddr = t_dst + (unsigned)(nxt * WN_VS * 4);
        dcvic_static_for<0, 5>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j < 4) {
                WN_WAIT_LDS();
                op_load(std::integral_constant<int, j + 1>{}, std::integral_constant<int, (j + 1 + PAR) & 1>{}, ua, va);
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __syncthreads();
                WN_FENCE();
                if constexpr (more1) op_load(std::integral_constant<int, 0>{}, std::integral_constant<int, PAR ^ 1>{}, op_u + (unsigned)(nxt * WN_US * 4), op_v + (unsigned)(nxt * WN_VS * 4));
            }
            WN_FENCE();
            dcvic_static_for<0, 8>([&](auto i_) {
                constexpr int i = decltype(i_)::value, pq = i >> 2, ks = (i >> 1) & 1, blk = i & 1, pp = 2 * j + pq;
                if constexpr (pp < WU_NPOS) {
                    acc[pp][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(opA[(j + PAR) & 1][pq * 2 + ks], opB[(j + PAR) & 1][pq][blk * 2 + ks], acc[pp][blk], 0, 0, 0);
                    WN_FENCE();
                }
                if constexpr (more1 && j == 0 && i == 1) dma_u(std::integral_constant<int, 0>{}, nxt);
                if constexpr (more1 && j == 0 && i == 5) dma_u(std::integral_constant<int, 1>{}, nxt);
                if constexpr (more1 && j == 1 && i == 1) dma_u(std::integral_constant<int, 2>{}, nxt);
                if constexpr (more2 && j == 1 && i == 5) dma_x(cur);
                if constexpr (more1 && j == 1 && (i == 2 || i == 3 || i == 6)) t_load(std::integral_constant<int, i == 6 ? 2 : i - 2>{}, xaddr);
                if constexpr (more1 && j == 2 && i == 1) t_cols();
                if constexpr (more1 && j == 2 && i == 3) t_rows();
                if constexpr (more1 && j == 3 && (i & 1)) t_store(std::integral_constant<int, (i >> 1)>{}, vaddr);
                if constexpr (more1 && j == 3 && i == 6) t_store(std::integral_constant<int, 4>{}, vaddr);
                WN_FENCE();
            });
        });
        if constexpr (more2) x_advance();
        if constexpr (more1) u_advance();
        if (++c_chunk == S) {
            tile_epilogue(c_cotile, c_n, c_oy0, c_ox0, c_par);
            c_chunk = 0; c_b += J; c_par ^= 1;
            if (c_b < xe) {
                decode(c_b, c_cotile, c_n, c_oy0, c_ox0);
                stage_bias(c_b, c_par);
            }
        }
        WN_FENCE();
    };
    {
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        int g = 0;
        for (; g + 2 < total; ++g) { if (g & 1) run_stage(std::true_type{}, std::true_type{}, P1{}, g); else run_stage(std::true_type{}, std::true_type{}, P0{}, g); }
        if (g + 1 < total) { if (g & 1) run_stage(std::true_type{}, std::false_type{}, P1{}, g); else run_stage(std::true_type{}, std::false_type{}, P0{}, g); ++g; }
        if (g & 1) run_stage(std::false_type{}, std::false_type{}, P1{}, g); else run_stage(std::false_type{}, std::false_type{}, P0{}, g);
    }
#undef WN_FENCE
#undef WN_WAIT_LDS
}

extern "C" size_t dcvic_wino_ups_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)((Cout + WN_CO - 1) / WN_CO) * ((Cin + KC - 1) / KC) * WU_US * sizeof(float);
}

extern "C" int dcvic_wino_ups_pack_f32(const float* w, float* packed, int Cin, int Cout, void* stream) {
    DCVIC_CHECK_ARG(w && packed && Cin > 0 && Cout > 0, "wino_ups_pack: bad argument");
    const int n_chunks = (Cin + KC - 1) / KC;
    const long long total = (long long)((Cout + WN_CO - 1) / WN_CO) * n_chunks * WU_US;
    wino_ups_pack_kernel<<<dcvic_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, packed, Cin, Cout, n_chunks, total);
    DCVIC_CHECK_LAUNCH("wino_ups_pack");
    return DCVIC_OK;
}

extern "C" int dcvic_conv3x3_wino_ups_f32(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, void* stream) {
    DCVIC_CHECK_ARG(packed && io && io->out && Cin > 0 && Cout > 0, "conv3x3_wino_ups: null pointer");
    DCVIC_CHECK_ARG(io->n_src >= 1 && io->n_src <= DCVIC_MAX_SRC, "conv3x3_wino_ups: n_src %d", io->n_src);
    int csum = 0;
    for (int i = 0; i < io->n_src; ++i) {
        DCVIC_CHECK_ARG(io->src[i].ptr && io->src[i].C > 0 && io->src[i].C % KC == 0, "conv3x3_wino_ups: source %d needs a multiple of 8 channels", i);
        DCVIC_CHECK_ARG(io->src[i].batch_stride >= (long long)io->src[i].C * io->H * io->W, "conv3x3_wino_ups: source %d batch stride too small", i);
        DCVIC_CHECK_ARG((reinterpret_cast<uintptr_t>(io->src[i].ptr) & 15) == 0 && (io->src[i].batch_stride & 3) == 0,
                        "conv3x3_wino_ups: source %d must be 16-byte aligned", i);
        csum += io->src[i].C;
    }
    DCVIC_CHECK_ARG(csum == Cin, "conv3x3_wino_ups: sources carry %d channels, layer expects %d", csum, Cin);
    DCVIC_CHECK_ARG(io->N > 0 && io->H > 0 && io->W > 0 && (io->W & 3) == 0, "conv3x3_wino_ups: bad sizes (input width must be a multiple of 4)");
    DCVIC_CHECK_ARG(io->Hfull == 2 * io->H && io->Wfull == 2 * io->W && io->Hout == io->Hfull && io->Wout == io->Wfull && io->osy == 1 && io->osx == 1 &&
                    io->ooy == 0 && io->oox == 0, "conv3x3_wino_ups: output must be the x2 plane");
    DCVIC_CHECK_ARG(!io->aff_scale && !io->aff_shift && !io->init, "conv3x3_wino_ups: affine / init epilogues are not supported");
    DCVIC_CHECK_ARG((long long)io->H * io->W * KC < (1ll << 31), "conv3x3_wino_ups: plane too large");
    const long long HWo = (long long)io->Hfull * io->Wfull;
    DCVIC_CHECK_ARG(io->out_batch_stride >= (long long)Cout * HWo && (io->out_batch_stride & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(io->out) & 15) == 0, "conv3x3_wino_ups: output view must be 16-byte aligned");
    DCVIC_CHECK_ARG(!io->res || (io->res_batch_stride >= (long long)Cout * HWo && (io->res_batch_stride & 3) == 0 &&
                                 (reinterpret_cast<uintptr_t>(io->res) & 15) == 0), "conv3x3_wino_ups: residual view must be 16-byte aligned");
    ConvKArgs K;
    memset(&K, 0, sizeof(K));
    K.Cin = Cin; K.Cout = Cout; K.T = 9; K.stride = 1;
    K.N = io->N; K.H = io->H; K.W = io->W; K.Hout = io->Hfull; K.Wout = io->Wfull; K.Hfull = io->Hfull; K.Wfull = io->Wfull;
    K.osy = K.osx = 1;
    for (int i = 0; i < DCVIC_MAX_SRC; ++i) {
        if (i < io->n_src) { K.src[i] = io->src[i].ptr; K.srcC[i] = io->src[i].C; K.src_bs[i] = io->src[i].batch_stride; }
        else { K.src[i] = io->src[0].ptr; K.srcC[i] = 1 << 30; K.src_bs[i] = 0; }
    }
    K.out = io->out; K.out_bs = io->out_batch_stride; K.bias = io->bias; K.act = io->act;
    K.res = io->res; K.res_bs = io->res_batch_stride;
    K.wp = packed;
    K.n_chunks = (Cin + KC - 1) / KC;
    K.n_cotiles = (Cout + WN_CO - 1) / WN_CO;
    K.tiles_y = (io->Hfull + WN_TH - 1) / WN_TH;
    K.tiles_x = (io->Wfull + WN_TW - 1) / WN_TW;
    const long long blocks = (long long)io->N * K.tiles_y * K.tiles_x * K.n_cotiles;
    DCVIC_CHECK_ARG(blocks < (1ll << 31), "conv3x3_wino_ups: grid too large");
    K.nblocks = (int)blocks;
    static std::atomic<unsigned> attr_mask{0};
    if (DcvicAttrOnce once_{attr_mask})
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino_ups_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int grid = (dcvic_num_cu() / NXCD) * NXCD;
    if (grid < NXCD) grid = NXCD;
    if ((long long)grid > blocks) grid = (int)((blocks + NXCD - 1) / NXCD) * NXCD;
    conv3x3_wino_ups_kernel<<<grid, WN_THREADS, WN_LDS_FLOATS * sizeof(float), (hipStream_t)stream>>>(K);
    DCVIC_CHECK_LAUNCH("conv3x3_wino_ups");
    return DCVIC_OK;
}

extern "C" size_t dcvic_wino_packed_bytes(int Cin, int Cout) {
    if (Cin <= 0 || Cout <= 0) return 0;
    return (size_t)((Cout + WN_CO - 1) / WN_CO) * ((Cin + KC - 1) / KC) * WN_US * sizeof(float);
}

extern "C" int dcvic_wino_pack_f32(const float* w, float* packed, int Cin, int Cout, void* stream) {
    DCVIC_CHECK_ARG(w && packed && Cin > 0 && Cout > 0, "wino_pack: bad argument");
    const int n_chunks = (Cin + KC - 1) / KC;
    const long long total = (long long)((Cout + WN_CO - 1) / WN_CO) * n_chunks * WN_US;
    wino_pack_kernel<<<dcvic_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, packed, Cin, Cout, n_chunks, total);
    DCVIC_CHECK_LAUNCH("wino_pack");
    return DCVIC_OK;
}

extern "C" int dcvic_conv3x3_wino_f32(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, void* stream) {
    DCVIC_CHECK_ARG(packed && io && io->out && Cin > 0 && Cout > 0, "conv3x3_wino: null pointer");
    DCVIC_CHECK_ARG(io->n_src >= 1 && io->n_src <= DCVIC_MAX_SRC, "conv3x3_wino: n_src %d", io->n_src);
    int csum = 0;
    for (int i = 0; i < io->n_src; ++i) {
        DCVIC_CHECK_ARG(io->src[i].ptr && io->src[i].C > 0 && io->src[i].C % KC == 0, "conv3x3_wino: source %d needs a multiple of 8 channels", i);
        DCVIC_CHECK_ARG(io->src[i].batch_stride >= (long long)io->src[i].C * io->H * io->W, "conv3x3_wino: source %d batch stride too small", i);
        DCVIC_CHECK_ARG((reinterpret_cast<uintptr_t>(io->src[i].ptr) & 15) == 0 && (io->src[i].batch_stride & 3) == 0,
                        "conv3x3_wino: source %d must be 16-byte aligned (16-byte LDS-DMA segments)", i);
        csum += io->src[i].C;
    }
    DCVIC_CHECK_ARG(csum == Cin, "conv3x3_wino: sources carry %d channels, layer expects %d", csum, Cin);
    DCVIC_CHECK_ARG(io->N > 0 && io->H > 0 && io->W > 0, "conv3x3_wino: bad sizes");
    DCVIC_CHECK_ARG(io->Hout == io->H && io->Wout == io->W && io->Hfull == io->H && io->Wfull == io->W && io->osy == 1 && io->osx == 1 &&
                    io->ooy == 0 && io->oox == 0, "conv3x3_wino: stride-1 pad-1 geometry only");
    DCVIC_CHECK_ARG((io->W & 3) == 0, "conv3x3_wino: width must be a multiple of 4");
    DCVIC_CHECK_ARG(!io->aff_scale && !io->aff_shift && !io->init, "conv3x3_wino: affine / init epilogues are not supported");
    DCVIC_CHECK_ARG((long long)io->H * io->W * KC < (1ll << 31), "conv3x3_wino: plane too large");
    DCVIC_CHECK_ARG(io->out_batch_stride >= (long long)Cout * io->H * io->W && (io->out_batch_stride & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(io->out) & 15) == 0, "conv3x3_wino: output view must be 16-byte aligned");
    DCVIC_CHECK_ARG(!io->res || (io->res_batch_stride >= (long long)Cout * io->H * io->W && (io->res_batch_stride & 3) == 0 &&
                                 (reinterpret_cast<uintptr_t>(io->res) & 15) == 0), "conv3x3_wino: residual view must be 16-byte aligned");
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
#ifdef DCVIC_WINO_EXPERIMENTS
    { const char* e = getenv("DCVIC_WINO_DEBUG"); K.TG = e ? atoi(e) : 0; }   // timing experiments only (wrong results)
#endif
    K.n_chunks = (Cin + KC - 1) / KC;
    K.n_cotiles = (Cout + WN_CO - 1) / WN_CO;
    K.tiles_y = (io->H + WN_TH - 1) / WN_TH;
    K.tiles_x = (io->W + WN_TW - 1) / WN_TW;
    const long long blocks = (long long)io->N * K.tiles_y * K.tiles_x * K.n_cotiles;
    DCVIC_CHECK_ARG(blocks < (1ll << 31), "conv3x3_wino: grid too large");
    K.nblocks = (int)blocks;
    static std::atomic<unsigned> attr_mask{0};
#ifdef DCVIC_WINO_EXPERIMENTS
    const int dbg = K.TG >> 4;
    K.TG &= 15;
    auto kern = dbg == 1 ? conv3x3_wino_kernel<1> : dbg == 2 ? conv3x3_wino_kernel<2> : dbg == 4 ? conv3x3_wino_kernel<4> : dbg == 8 ? conv3x3_wino_kernel<8> :
                dbg == 6 ? conv3x3_wino_kernel<6> : dbg == 16 ? conv3x3_wino_kernel<16> : dbg == 32 ? conv3x3_wino_kernel<32> : dbg == 64 ? conv3x3_wino_kernel<64> : dbg == 128 ? conv3x3_wino_kernel<128> : dbg == 256 ? conv3x3_wino_kernel<256> : conv3x3_wino_kernel<0>;
    if (dbg) hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#else
    auto kern = conv3x3_wino_kernel<0>;
#endif
    if (DcvicAttrOnce once_{attr_mask})
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    // persistent grid: one workgroup per CU (152 KiB of LDS each), a multiple of the 8 XCDs; each walks its share of the tiles
    int grid = (dcvic_num_cu() / NXCD) * NXCD;
    if (grid < NXCD) grid = NXCD;
    if ((long long)grid > blocks) grid = (int)((blocks + NXCD - 1) / NXCD) * NXCD;
    kern<<<grid, WN_THREADS, WN_LDS_FLOATS * sizeof(float), (hipStream_t)stream>>>(K);
    DCVIC_CHECK_LAUNCH("conv3x3_wino");
    return DCVIC_OK;
}
