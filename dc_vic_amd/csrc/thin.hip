// thin.hip -- Conv2d(k3, s1, p1) layers with almost no channels on one side: HBM-bound VALU kernels, gfx950.
//
// Two layers of the path are "thin": the VQGAN encoder's conv_in (3 -> 128 at full resolution, ldm model.py:388-392) and the VQGAN
// decoder's conv_out (128 -> 3, model.py:553-557).  Their contraction is 27 / 1 152 terms per output -- no GEMM to speak of -- but
// they read or write a full-resolution 128-channel map (1.07 GB for 32 images of 256 x 256).  On the MFMA kernels they ran with 3 of
// 32 tile rows / columns occupied: 13 and 8.5 TFLOP/s, i.e. 1.0 - 1.7 ms per launch where the bytes need 0.25 ms.  Here they are plain
// fp32 fmaf chains on the vector ALU IN THE SAME ORDER as conv.hip's MFMA chains (chunk ascending, then (4-channel half, tap, channel)
// for the 3x3 / stride-1 family with Cin % 8 == 0, (tap, channel) otherwise; an fp32 MFMA is an exactly ordered fmaf chain on this
// chip), so the results are BIT-IDENTICAL to the kernels they replace: nothing upstream or downstream of them changes by a single bit.
//   * thin_cout_kernel<CO>: Cout = CO <= 4.  Per 4-channel half the (18 x 66)-pixel input patch is staged in LDS (coalesced row reads,
//     zero padding applied by the loader, the next half's loads in flight during the arithmetic); a thread holds the 3 x 6 window of
//     its 4 pixels for the 4 channels in registers (24 ds_read_b128) and runs 36 x CO x 4 fmaf on it, weights as LDS broadcasts.
//   * thin_cin_kernel<CI>: Cin = CI <= 4.  4 pixels per thread: their 3 x 6 x CI input values live in registers, then per output
//     channel one fmaf chain per pixel (weights through the scalar cache) and one 16-byte store (1 KiB per wave-instruction).
#include "common.h"
#include <type_traits>

struct ThinArgs {
    const float* x; long long x_bs;
    const float* w;            // [Cout][Cin][3][3], unpacked
    const float* bias;
    const float* res; long long res_bs;
    float* out; long long out_bs;
    int N, H, W, Cin, Cout, act;
    int tiles_x, tiles_y;
    int vec;                   // rows of out / res are 16-byte aligned (W % 4 == 0, aligned bases and batch strides)
};

__device__ float dcvic_thin_zero[4];   // zero-initialised: source of out-of-image elements

#define TH_ROWS 16
#define TH_COLS 64
#define TH_PW 68            // LDS row stride (66 used; 68 = 4 mod 64 banks: the 16-lane passes of a ds_read_b128 never collide)
#define TH_PR (TH_ROWS + 2)
#define TH_PLANE (TH_PR * TH_PW)

// Both kernels: workgroup = 16 rows x 64 columns of output pixels, thread = 4 consecutive pixels of one row (16-byte stores / loads
// wherever the row allows), wave = 4 rows x 64 columns.

template <int CO>
__global__ __launch_bounds__(256, 2) void thin_cout_kernel(const ThinArgs A) {
    __shared__ __attribute__((aligned(16))) float Xs[4 * TH_PLANE];
    __shared__ __attribute__((aligned(16))) float Wsm[36 * 4];     // [tap 9][channel 4][co padded to 4]
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tile_x = b % A.tiles_x; b /= A.tiles_x;
    const int tile_y = b % A.tiles_y; b /= A.tiles_y;
    const int n = b;
    const int oy0 = tile_y * TH_ROWS, ox0 = tile_x * TH_COLS;
    const int sx = tid & 15, ty = tid >> 4;                       // pixels (ty, 4 sx .. 4 sx + 3) of the tile
    const long long HW = (long long)A.H * A.W;
    const float* xn = A.x + (long long)n * A.x_bs;
    float acc[4][CO];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[p][co] = 0.f;
    const int n_halves = A.Cin / 4;
    const int wave = tid >> 6, lane = tid & 63;
    // a stage is a 4-channel half of conv.hip's 8-channel chunk: the (18 x 66)-pixel patch of its channels (zero outside the image)
    // and its 36 x CO weights.  72 patch rows: wave w takes rows w, w + 4, ... (one coalesced 64-lane load + a 2-lane tail each).
    // The loads of stage h + 1 are issued before the arithmetic of stage h and stored to LDS after it: one memory round trip per
    // stage, hidden behind 432 fmaf per thread.
    float pv[18], ptail, wreg;
    auto fetch = [&](int h) {
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int rr = wave + 4 * i;
            const int c = rr / TH_PR, py = rr - c * TH_PR;
            const int iy = oy0 - 1 + py;
            const int ix = ox0 - 1 + lane;
            // (branch-free: an out-of-image element reads the zero word.  A select on the loaded VALUE makes hipcc put every load in its
            // own exec-masked block with a vmcnt(0) behind it: eighteen serial round trips instead of one)
            const float* src = (iy >= 0 && iy < A.H && ix >= 0 && ix < A.W) ? xn + ((long long)(h * 4 + c) * HW + (long long)iy * A.W + ix) : dcvic_thin_zero;
            pv[i] = *src;
        }
        {   // the two columns behind the 64-lane row loads: thread t < 144 takes column 64 + (t & 1) of patch row t / 2
            const int rr = tid >> 1;
            const int c = rr / TH_PR, py = rr - c * TH_PR;
            const int iy = oy0 - 1 + py, ix = ox0 + 63 + (tid & 1);
            const float* src = (tid < 144 && iy >= 0 && iy < A.H && ix < A.W) ? xn + ((long long)(h * 4 + c) * HW + (long long)iy * A.W + ix) : dcvic_thin_zero;
            ptail = *src;
        }
        const int co = tid & 3, c4 = (tid >> 2) & 3, tap = tid >> 4;
        const float* wsrc = (tid < 144 && co < CO) ? A.w + (((long long)co * A.Cin + h * 4 + c4) * 9 + tap) : dcvic_thin_zero;
        wreg = *wsrc;
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int rr = wave + 4 * i;
            const int c = rr / TH_PR, py = rr - c * TH_PR;
            Xs[c * TH_PLANE + py * TH_PW + lane] = pv[i];
        }
        if (tid < 144) {
            const int rr = tid >> 1;
            const int c = rr / TH_PR, py = rr - c * TH_PR;
            Xs[c * TH_PLANE + py * TH_PW + 64 + (tid & 1)] = ptail;
            Wsm[tid] = wreg;
        }
    };
    fetch(0);
    for (int h = 0; h < n_halves; ++h) {
        commit();
        __syncthreads();
        if (h + 1 < n_halves) fetch(h + 1);
        // this thread's window: 3 rows x 6 columns (two aligned 16-byte reads per row) of the 4 channels, in registers
        float win[4][3][8];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float4 lo = *reinterpret_cast<const float4*>(Xs + c * TH_PLANE + (ty + r) * TH_PW + 4 * sx);
                const float4 hi = *reinterpret_cast<const float4*>(Xs + c * TH_PLANE + (ty + r) * TH_PW + 4 * sx + 4);
                win[c][r][0] = lo.x; win[c][r][1] = lo.y; win[c][r][2] = lo.z; win[c][r][3] = lo.w;
                win[c][r][4] = hi.x; win[c][r][5] = hi.y; win[c][r][6] = hi.z; win[c][r][7] = hi.w;
            }
        // conv.hip's order inside a chunk for this layer family: 4-channel half, tap, channel.  The weights of tap t + 1 (four LDS
        // broadcasts) are read while tap t computes; the fences keep hipcc from hoisting all 36 reads (144 registers) to the top.
        float4 wcur[4], wnxt[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) wcur[c] = *reinterpret_cast<const float4*>(Wsm + c * 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            if (tap < 8) {
#pragma unroll
                for (int c = 0; c < 4; ++c) wnxt[c] = *reinterpret_cast<const float4*>(Wsm + ((tap + 1) * 4 + c) * 4);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float wv[4] = {wcur[c].x, wcur[c].y, wcur[c].z, wcur[c].w};
#pragma unroll
                for (int co = 0; co < CO; ++co)
#pragma unroll
                    for (int p = 0; p < 4; ++p) acc[p][co] = __builtin_fmaf(wv[co], win[c][ky][p + kx], acc[p][co]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) wcur[c] = wnxt[c];
        }
        __syncthreads();
    }
    const int oy = oy0 + ty, ox = ox0 + 4 * sx;
    if (oy >= A.H || ox >= A.W) return;
    const long long pix = (long long)oy * A.W + ox;
    const bool vec = A.vec && ox + 3 < A.W;
#pragma unroll
    for (int co = 0; co < CO; ++co) {
        float v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            v[p] = acc[p][co];
            if (A.bias) v[p] += A.bias[co];                       // ("+ 0" would turn -0 into +0: keep the no-bias path exact)
            v[p] = dcvic_act(v[p], A.act);
        }
        float* op = A.out + (long long)n * A.out_bs + (long long)co * HW + pix;
        const float* rp = A.res ? A.res + (long long)n * A.res_bs + (long long)co * HW + pix : nullptr;
        if (vec) {
            if (rp) { const float4 r4 = *reinterpret_cast<const float4*>(rp); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (ox + p < A.W) op[p] = rp ? v[p] + rp[p] : v[p];
        }
    }
}

template <int CI>
__global__ __launch_bounds__(256) void thin_cin_kernel(const ThinArgs A) {
    constexpr int WSTR = (CI * 9 + 3) & ~3;                        // floats per output channel in LDS (16-byte rows)
    __shared__ __attribute__((aligned(16))) float Wl[128 * WSTR];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tile_x = b % A.tiles_x; b /= A.tiles_x;
    const int tile_y = b % A.tiles_y; b /= A.tiles_y;
    const int n = b;
    const int ox = tile_x * TH_COLS + 4 * (tid & 15), oy = tile_y * TH_ROWS + (tid >> 4);
    const bool live = ox < A.W && oy < A.H;
    const long long HW = (long long)A.H * A.W;
    const float* xn = A.x + (long long)n * A.x_bs;
    float xv[3][CI][6];                                           // [row][channel][column ox - 1 .. ox + 4]
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = oy - 1 + r;
        const bool row_in = live && iy >= 0 && iy < A.H;
#pragma unroll
        for (int c = 0; c < CI; ++c)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int ix = ox - 1 + j;
                const float* src = (row_in && ix >= 0 && ix < A.W) ? xn + ((long long)c * HW + (long long)iy * A.W + ix) : dcvic_thin_zero;
                xv[r][c][j] = *src;
            }
    }
    const long long pix = (long long)oy * A.W + ox;
    float* op = A.out + (long long)n * A.out_bs + pix;
    const float* rp = A.res ? A.res + (long long)n * A.res_bs + pix : nullptr;
    const bool vec = A.vec && ox + 3 < A.W;
    // the weights of (up to) 128 output channels sit in LDS and are read back as 16-byte broadcasts: inside the channel loop they
    // would otherwise be vector loads (hipcc does not use the scalar cache for a pointer it sees stores next to)
    for (int co0 = 0; co0 < A.Cout; co0 += 128) {
        const int nco = min(128, A.Cout - co0);
        if (co0) __syncthreads();
        for (int e = tid; e < nco * CI * 9; e += 256) {
            const int co = e / (CI * 9), k = e - co * (CI * 9);
            Wl[co * WSTR + k] = A.w[(long long)(co0 + co) * (CI * 9) + k];
        }
        __syncthreads();
        if (!live) continue;
        auto channels = [&](auto plain) {
#pragma unroll 2
            for (int cl = 0; cl < nco; ++cl) {
                const int co = co0 + cl;
                float wl[WSTR];
#pragma unroll
                for (int q = 0; q < WSTR / 4; ++q) {
                    const float4 t = *reinterpret_cast<const float4*>(Wl + cl * WSTR + 4 * q);
                    wl[4 * q] = t.x; wl[4 * q + 1] = t.y; wl[4 * q + 2] = t.z; wl[4 * q + 3] = t.w;
                }
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                // conv.hip's order for a layer whose Cin is not a multiple of 8: tap, then channel
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                    for (int c = 0; c < CI; ++c)
#pragma unroll
                        for (int p = 0; p < 4; ++p) acc[p] = __builtin_fmaf(wl[c * 9 + tap], xv[tap / 3][c][p + tap % 3], acc[p]);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    if (A.bias) acc[p] += A.bias[co];
                    if (!decltype(plain)::value) acc[p] = dcvic_act(acc[p], A.act);
                }
                if (vec) {
                    if (rp) { const float4 r4 = *reinterpret_cast<const float4*>(rp + (long long)co * HW); acc[0] += r4.x; acc[1] += r4.y; acc[2] += r4.z; acc[3] += r4.w; }
                    *reinterpret_cast<float4*>(op + (long long)co * HW) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                } else {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
                        if (ox + p < A.W) op[(long long)co * HW + p] = rp ? acc[p] + rp[(long long)co * HW + p] : acc[p];
                }
            }
        };
        if (A.act == 0) channels(std::true_type{}); else channels(std::false_type{});
    }
}

// Conv2d(k3, s1, p1) with Cout <= 4 (Cin % 8 == 0) or Cin <= 4: bit-identical to dcvic_conv2d_f32 on the same layer.  One source, no
// affine / init epilogue.  Returns DCVIC_EINVAL for any other layer (callers test dcvic_conv3x3_thin_applies first).
extern "C" int dcvic_conv3x3_thin_applies(int Cin, int Cout) {
    return ((Cout >= 1 && Cout <= 4 && Cin >= 8 && (Cin % 8) == 0) || (Cin >= 1 && Cin <= 4 && Cout >= 1)) ? 1 : 0;
}

extern "C" int dcvic_conv3x3_thin_f32(const float* w, int Cin, int Cout, const dcvic_conv_io* io, void* stream) {
    DCVIC_CHECK_ARG(w && io && io->out, "conv3x3_thin: null pointer");
    DCVIC_CHECK_ARG(dcvic_conv3x3_thin_applies(Cin, Cout), "conv3x3_thin: Cin %d / Cout %d is not a thin layer", Cin, Cout);
    DCVIC_CHECK_ARG(io->n_src == 1 && io->src[0].ptr && io->src[0].C == Cin, "conv3x3_thin: one source with Cin channels");
    DCVIC_CHECK_ARG(io->N > 0 && io->H > 0 && io->W > 0, "conv3x3_thin: bad sizes");
    DCVIC_CHECK_ARG(io->Hout == io->H && io->Wout == io->W && io->Hfull == io->H && io->Wfull == io->W && io->osy == 1 && io->osx == 1 &&
                    io->ooy == 0 && io->oox == 0, "conv3x3_thin: stride-1 pad-1 geometry only");
    DCVIC_CHECK_ARG(!io->aff_scale && !io->aff_shift && !io->init, "conv3x3_thin: affine / init epilogues are not supported");
    const long long HW = (long long)io->H * io->W;
    DCVIC_CHECK_ARG(io->src[0].batch_stride >= (long long)Cin * HW && io->out_batch_stride >= (long long)Cout * HW, "conv3x3_thin: batch stride too small");
    DCVIC_CHECK_ARG(!io->res || io->res_batch_stride >= (long long)Cout * HW, "conv3x3_thin: residual batch stride too small");
    ThinArgs A;
    A.x = io->src[0].ptr; A.x_bs = io->src[0].batch_stride; A.w = w; A.bias = io->bias; A.res = io->res; A.res_bs = io->res_batch_stride;
    A.out = io->out; A.out_bs = io->out_batch_stride; A.N = io->N; A.H = io->H; A.W = io->W; A.Cin = Cin; A.Cout = Cout; A.act = io->act;
    A.vec = (io->W % 4 == 0) && (reinterpret_cast<uintptr_t>(io->out) % 16 == 0) && (io->out_batch_stride % 4 == 0) &&
            (!io->res || ((reinterpret_cast<uintptr_t>(io->res) % 16 == 0) && (io->res_batch_stride % 4 == 0)));
    hipStream_t st = (hipStream_t)stream;
    if (Cout <= 4 && Cin >= 8) {
        A.tiles_x = dcvic_cdiv(io->W, TH_COLS); A.tiles_y = dcvic_cdiv(io->H, TH_ROWS);
        const long long blocks = (long long)io->N * A.tiles_x * A.tiles_y;
        DCVIC_CHECK_ARG(blocks < (1ll << 31), "conv3x3_thin: grid too large");
        switch (Cout) {
            case 1: thin_cout_kernel<1><<<(int)blocks, 256, 0, st>>>(A); break;
            case 2: thin_cout_kernel<2><<<(int)blocks, 256, 0, st>>>(A); break;
            case 3: thin_cout_kernel<3><<<(int)blocks, 256, 0, st>>>(A); break;
            default: thin_cout_kernel<4><<<(int)blocks, 256, 0, st>>>(A); break;
        }
    } else {
        A.tiles_x = dcvic_cdiv(io->W, TH_COLS); A.tiles_y = dcvic_cdiv(io->H, TH_ROWS);
        const long long blocks = (long long)io->N * A.tiles_x * A.tiles_y;
        DCVIC_CHECK_ARG(blocks < (1ll << 31), "conv3x3_thin: grid too large");
        switch (Cin) {
            case 1: thin_cin_kernel<1><<<(int)blocks, 256, 0, st>>>(A); break;
            case 2: thin_cin_kernel<2><<<(int)blocks, 256, 0, st>>>(A); break;
            case 3: thin_cin_kernel<3><<<(int)blocks, 256, 0, st>>>(A); break;
            default: thin_cin_kernel<4><<<(int)blocks, 256, 0, st>>>(A); break;
        }
    }
    DCVIC_CHECK_LAUNCH("conv3x3_thin");
    return DCVIC_OK;
}
