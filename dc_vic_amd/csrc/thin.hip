// thin.hip -- Conv2d(k3, s1, p1) layers with almost no channels on one side: HBM-bound VALU kernels, gfx950.
//
// Two layers of the path are "thin": the VQGAN encoder's conv_in (3 -> 128 at full resolution, ldm model.py:388-392) and the VQGAN
// decoder's conv_out (128 -> 3, model.py:553-557).  Their contraction is 27 / 1 152 terms per output -- no GEMM to speak of -- but
// they read or write a full-resolution 128-channel map (1.07 GB for 32 images of 256 x 256).  On the MFMA kernels they ran with 3 of
// 32 tile rows / columns occupied: 13 and 8.5 TFLOP/s, i.e. 1.0 - 1.7 ms per launch where the bytes need 0.25 ms.  Here they are plain
// fp32 fmaf chains on the vector ALU IN THE SAME ORDER as conv.hip's MFMA chains (chunk ascending, then (4-channel half, tap, channel)
// for the 3x3 / stride-1 family with Cin % 8 == 0, (tap, channel) otherwise; an fp32 MFMA is an exactly ordered fmaf chain on this
// chip), so the results are BIT-IDENTICAL to the kernels they replace: nothing upstream or downstream of them changes by a single bit.
//   * thin_cout_kernel<CO>: Cout = CO <= 4.  Workgroup = 8 x 64 output pixels, 256 threads x 2 pixels; per 8-channel chunk the
//     (10 x 66)-pixel input patch is staged in LDS (coalesced row reads, zero padding applied by the loader), every thread runs
//     8 x 9 x CO fmaf per pixel with the weights as scalar operands (uniform loads through the scalar cache).
//   * thin_cin_kernel<CI>: Cin = CI <= 4.  One pixel per thread: its CI x 9 input values live in registers, then one fmaf chain and
//     one coalesced 256-byte store per output channel.
#include "common.h"

struct ThinArgs {
    const float* x; long long x_bs;
    const float* w;            // [Cout][Cin][3][3], unpacked
    const float* bias;
    const float* res; long long res_bs;
    float* out; long long out_bs;
    int N, H, W, Cin, Cout, act;
    int tiles_x, tiles_y;
};

#define TH_ROWS 8
#define TH_COLS 64
#define TH_PW 68            // LDS row stride (66 used)
#define TH_PLANE (10 * TH_PW)

template <int CO>
__global__ __launch_bounds__(256) void thin_cout_kernel(const ThinArgs A) {
    __shared__ float Xs[8 * TH_PLANE];
    __shared__ float Wsm[72 * 4];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tile_x = b % A.tiles_x; b /= A.tiles_x;
    const int tile_y = b % A.tiles_y; b /= A.tiles_y;
    const int n = b;
    const int oy0 = tile_y * TH_ROWS, ox0 = tile_x * TH_COLS;
    const int tx = tid & 63, ty = tid >> 6;                       // pixels (ty, tx) and (ty + 4, tx) of the tile
    const long long HW = (long long)A.H * A.W;
    const float* xn = A.x + (long long)n * A.x_bs;
    float acc[2][CO];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[p][co] = 0.f;
    const int n_chunks = A.Cin / 8;
    const int wave = tid >> 6, lane = tid & 63;
    for (int ch = 0; ch < n_chunks; ++ch) {
        // stage the chunk's patch: rows oy0 - 1 .. oy0 + 8, columns ox0 - 1 .. ox0 + 64 of 8 channels (zero outside the image).
        // 80 patch rows of 66 floats: wave w takes rows w, w + 4, ...: one coalesced 64-lane load + a 2-lane tail per row
        // (all 20 + 20 loads of a wave are issued before the first LDS store: one memory round trip per chunk, not twenty)
        float pv[20], pt[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            const int rr = wave + 4 * i;
            const int c = rr / 10, py = rr - c * 10;
            const int iy = oy0 - 1 + py;
            const float* xr = xn + (long long)(ch * 8 + c) * HW + (long long)iy * A.W;
            const bool row_in = iy >= 0 && iy < A.H;
            const int ix = ox0 - 1 + lane;
            pv[i] = (row_in && ix >= 0 && ix < A.W) ? xr[ix] : 0.f;
            pt[i] = (lane < 2 && row_in && ix + 64 < A.W) ? xr[ix + 64] : 0.f;
        }
        float wreg[(72 * CO + 255) / 256];
#pragma unroll
        for (int j = 0; j < (72 * CO + 255) / 256; ++j) {
            const int e = tid + 256 * j;
            const int co = e % CO, ct = e / CO, c = ct / 9, tap = ct - 9 * c;
            wreg[j] = e < 72 * CO ? A.w[((long long)co * A.Cin + ch * 8 + c) * 9 + tap] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            const int rr = wave + 4 * i;
            const int c = rr / 10, py = rr - c * 10;
            Xs[c * TH_PLANE + py * TH_PW + lane] = pv[i];
            if (lane < 2) Xs[c * TH_PLANE + py * TH_PW + 64 + lane] = pt[i];
        }
        // the chunk's weights [channel 8][tap 9][CO] (read back as LDS broadcasts: scalar loads would share lgkmcnt with the patch reads)
#pragma unroll
        for (int j = 0; j < (72 * CO + 255) / 256; ++j)
            if (tid + 256 * j < 72 * CO) Wsm[tid + 256 * j] = wreg[j];
        __syncthreads();
        // conv.hip's order inside a chunk for this layer family: 4-channel half, tap, channel
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int c = half * 4 + c4;
                    const float x0 = Xs[c * TH_PLANE + (ty + ky) * TH_PW + tx + kx];
                    const float x1 = Xs[c * TH_PLANE + (ty + 4 + ky) * TH_PW + tx + kx];
#pragma unroll
                    for (int co = 0; co < CO; ++co) {
                        const float wv = Wsm[(c * 9 + tap) * CO + co];
                        acc[0][co] = __builtin_fmaf(wv, x0, acc[0][co]);
                        acc[1][co] = __builtin_fmaf(wv, x1, acc[1][co]);
                    }
                }
            }
        }
        __syncthreads();
    }
    const int ox = ox0 + tx;
    if (ox >= A.W) return;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int oy = oy0 + ty + 4 * p;
        if (oy >= A.H) continue;
        const long long pix = (long long)oy * A.W + ox;
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float v = acc[p][co];
            if (A.bias) v += A.bias[co];                          // ("+ 0" would turn -0 into +0: keep the no-bias path exact)
            v = dcvic_act(v, A.act);
            if (A.res) v += A.res[(long long)n * A.res_bs + (long long)co * HW + pix];
            A.out[(long long)n * A.out_bs + (long long)co * HW + pix] = v;
        }
    }
}

template <int CI>
__global__ __launch_bounds__(256) void thin_cin_kernel(const ThinArgs A) {
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tile_x = b % A.tiles_x; b /= A.tiles_x;
    const int tile_y = b % A.tiles_y; b /= A.tiles_y;
    const int n = b;
    const int ox = tile_x * 64 + (tid & 63), oy = tile_y * 4 + (tid >> 6);
    if (ox >= A.W || oy >= A.H) return;
    const long long HW = (long long)A.H * A.W;
    const float* xn = A.x + (long long)n * A.x_bs;
    float xv[9][CI];                                              // [tap][channel]
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int iy = oy - 1 + tap / 3, ix = ox - 1 + tap % 3;
        const bool in = iy >= 0 && iy < A.H && ix >= 0 && ix < A.W;
#pragma unroll
        for (int c = 0; c < CI; ++c) xv[tap][c] = in ? xn[(long long)c * HW + (long long)iy * A.W + ix] : 0.f;
    }
    const long long pix = (long long)oy * A.W + ox;
    float* op = A.out + (long long)n * A.out_bs + pix;
    const float* rp = A.res ? A.res + (long long)n * A.res_bs + pix : nullptr;
#pragma unroll 4
    for (int co = 0; co < A.Cout; ++co) {
        const float* wc = A.w + (long long)co * CI * 9;           // uniform: scalar loads
        float acc = 0.f;
        // conv.hip's order for a layer whose Cin is not a multiple of 8: tap, then channel
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int c = 0; c < CI; ++c) acc = __builtin_fmaf(wc[c * 9 + tap], xv[tap][c], acc);
        if (A.bias) acc += A.bias[co];
        acc = dcvic_act(acc, A.act);
        if (rp) acc += rp[(long long)co * HW];
        op[(long long)co * HW] = acc;
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
        A.tiles_x = dcvic_cdiv(io->W, 64); A.tiles_y = dcvic_cdiv(io->H, 4);
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
