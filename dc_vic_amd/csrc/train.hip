// train.hip -- backward / loss / optimizer kernels of the training step (reference:
// src/trainer/dual_cond_gan_distortion_vq_code_trainer.py:135-300, src/losses/*, torch autograd of the layers in a14-a17).
// Forward kernels are the inference ones; data-gradients of convolutions reuse them too (the adjoint of a convolution is a
// convolution with transposed / flipped weights, of a stride-2 transposed convolution a stride-2 convolution).  New here:
//   conv_wgrad      dW = sum_pixels dY (x) X   -- fp32 MFMA, split over pixel slabs, slab partials reduced in a FIXED order
//   groupnorm / layernorm backward (+ fused swish), activation and gate backwards, per-channel reductions,
//   column-softmax backward (attention), Swin window-attention backward, MSE / BCE-with-logits / cross-entropy with
//   gradients, Adam, deterministic sum of squares (gradient clipping).
// Everything is deterministic: no atomics, reductions in a layer-defined order (DDP ranks and reruns agree bit for bit).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ double wsum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wsum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// block-wide fp64 sum, fixed order (waves ascending); red: >= 16 doubles
__device__ __forceinline__ double bsum_d(double v, double* red) {
    v = wsum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// ------------------------------------------------------------------------------------------------ conv weight gradient
// dW[m][c][ky][kx] = sum_{n, oy, ox} G[n][m][oy][ox] * X[n][c][oy*s + ky - pt][ox*s + kx - pl]
// (Conv2d: G = dY, X = the layer input, dW in the Conv2d layout [Cout][Cin][KH][KW].  ConvTranspose2d(stride s): G = the
//  layer INPUT, X = dY, and the result is the ConvTranspose2d layout [Cin][Cout][KH][KW].)
// GEMM view per kernel row ky: M = 128 rows m, N = 32 columns c (x KW taps), K = pixels.  A = G[m][pix], B = X[c][pix+tap]:
// both operands have the contraction index contiguous in memory, so the LDS images are [row][pix] with ODD row strides and
// the MFMA 32x32x2 operand reads (lane -> row, lane half -> pixel parity) are conflict-free.
// Workgroup = (m block of 128, c block of 32, ky, pixel slab); 4 waves, wave w owns rows 32 w .. 32 w + 31 and KW accumulators.
// Partials [slab][m][c][ky][kx] are reduced by wgrad_reduce_kernel in ascending slab order.
__device__ __attribute__((aligned(16))) float dcvic_train_zero[4];   // zero-initialised: source of out-of-range elements

struct WgradArgs {
    const float* G; long long g_bs; int M, Hg, Wg;       // "gradient-side" map: N x M x Hg x Wg (pixel index of the sum)
    const float* X; long long x_bs; int Cx, Hx, Wx;      // "input-side" map:    N x Cx x Hx x Wx
    float* part;                                         // [slabs][M][Cx][KH][KW]
    int N, KH, KW, stride, pt, pl;
    int rows_per_slab, slabs_per_img;
    int mblocks, cblocks;
};

// KHT = 1: the kernel row ky is a grid dimension (any KH <= 5, stride <= 4).  KHT = KH (3x3 / stride 1, the bulk of the trained
// layers): one workgroup accumulates all KH x KW taps -- the G chunk is loaded once for nine taps instead of once per kernel row, and a
// chunk carries 144 MFMAs per wave between two barriers instead of 48 (the first build spent 25 k cycles per chunk on 9 k of MFMAs).
template <int KWT, int KHT>
__global__ __launch_bounds__(256, KHT > 1 ? 2 : 1) void conv_wgrad_kernel(const WgradArgs a) {   // (KHT > 1: 144 accumulator + <= 112 other registers, two workgroups per CU)
    constexpr int PX = 32;                 // pixels per chunk (16 MFMA k-steps)
    constexpr int AS = 33;                 // A image row stride
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    int b = blockIdx.x;
    const int mb = b % a.mblocks; b /= a.mblocks;
    const int cb = b % a.cblocks; b /= a.cblocks;
    int ky = 0;
    if (KHT == 1) { ky = b % a.KH; b /= a.KH; }             // first kernel row of this workgroup
    const int slab = b;
    const int n = slab / a.slabs_per_img, srow0 = (slab % a.slabs_per_img) * a.rows_per_slab;
    const int xw = (PX - 1) * a.stride + a.KW;             // input pixels per chunk row
    const int XS = xw | 1;                                  // odd
    // double-buffered LDS images; the next chunk's global loads are in flight (registers) while this chunk's MFMAs run
    const int XSZ = KHT * 32 * XS;
    float* As0 = sm;                                       // 2 x [128][AS]
    float* Xs0 = sm + 2 * 128 * AS;                        // 2 x [KHT][32][XS]
    f32x16 acc[KHT * KWT];
#pragma unroll
    for (int t = 0; t < KHT * KWT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* Gn = a.G + (long long)n * a.g_bs;
    const float* Xn = a.X + (long long)n * a.x_bs;
    const int m0 = mb * 128, c0 = cb * 32;
    const int rend = min(srow0 + a.rows_per_slab, a.Hg);
    const int cpr = (a.Wg + PX - 1) / PX;                  // chunks per row
    const int nchunks = max(rend - srow0, 0) * cpr;
    constexpr int NA = 128 * PX / 256;                     // 16 A elements per thread = four 16-byte groups
    constexpr int NA4 = NA / 4;
    constexpr int NXMAX = KHT > 1 ? (KHT * 32 * (31 + 5) + 255) / 256 : (32 * (31 * 4 + 5) + 255) / 256;   // KHT > 1: stride 1 only; else stride <= 4, KW <= 5
    const int nx = (KHT * 32 * xw + 255) / 256;
    // Everything about an element that does not depend on the chunk is derived ONCE (row / column of the A group, channel and
    // column of the X element, LDS offsets): per chunk a load costs one add and one select.  (The first build re-derived
    // (e / xw, e % xw, 64-bit row offsets) per element and chunk: ~1 300 address instructions per 48 MFMAs -- the vector ALU, which
    // an fp32 MFMA shares its lanes with, was busier than the matrix pipe.)  Out-of-range elements read a zero word (select on the
    // ADDRESS: the loads stay independent of each other).
    const bool a_vec = (a.Wg % 4 == 0) && (a.g_bs % 4 == 0) && (reinterpret_cast<uintptr_t>(a.G) % 16 == 0);
    const int a_p4 = 4 * (tid & 7);                        // group u: row (tid >> 3) + 32 u, pixels a_p4 .. a_p4 + 3 of the chunk
    const float* const a_base = Gn + (long long)(m0 + (tid >> 3)) * a.Hg * a.Wg + a_p4;
    const long long a_step = 32ll * a.Hg * a.Wg;
    int x_off[NXMAX], x_pk[NXMAX];                          // x_pk = LDS offset << 8 | kernel row << 6 | column
#pragma unroll
    for (int u = 0; u < NXMAX; ++u) {
        const int e = tid + u * 256;
        x_off[u] = -1; x_pk[u] = 0;
        if (u < nx && e < KHT * 32 * xw) {
            const int kyi = e / (32 * xw), e2 = e - kyi * (32 * xw);
            const int c = e2 / xw, q = e2 - c * xw;
            x_pk[u] = (((kyi * 32 + c) * XS + q) << 8) | (kyi << 6) | q;     // (q <= 35 for KHT > 1; KHT = 1: kyi = 0 and q < 256)
            if (c0 + c < a.Cx) x_off[u] = (c0 + c) * a.Hx * a.Wx + q;
        }
    }
    float ra[NA], rx[NXMAX];
    auto fetch = [&](int ch) {
        const int oy = srow0 + ch / cpr, ox0 = (ch % cpr) * PX;
        const int iy0 = oy * a.stride + ky - a.pt;         // input row of kernel row ky (+ x_ky[u] for the others)
        const int g_add = oy * a.Wg + ox0;
        if (a_vec) {
#pragma unroll
            for (int u = 0; u < NA4; ++u) {
                const bool ok = m0 + (tid >> 3) + 32 * u < a.M && ox0 + a_p4 < a.Wg;   // (Wg % 4 == 0: a group is inside or outside the row)
                const float4 v = *reinterpret_cast<const float4*>(ok ? a_base + u * a_step + g_add : dcvic_train_zero);
                ra[4 * u] = v.x; ra[4 * u + 1] = v.y; ra[4 * u + 2] = v.z; ra[4 * u + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < NA4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = m0 + (tid >> 3) + 32 * u < a.M && ox0 + a_p4 + j < a.Wg;
                    ra[4 * u + j] = *(ok ? a_base + u * a_step + g_add + j : dcvic_train_zero);
                }
        }
        const int ix0 = ox0 * a.stride - a.pl;
#pragma unroll
        for (int u = 0; u < NXMAX; ++u) {
            if (u < nx) {
                const int ix = ix0 + (KHT > 1 ? (x_pk[u] & 63) : (x_pk[u] & 255)), iy = iy0 + (KHT > 1 ? ((x_pk[u] >> 6) & 3) : 0);
                const bool ok = x_off[u] >= 0 && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx;
                rx[u] = *(ok ? Xn + (x_off[u] + iy * a.Wx + ix0) : dcvic_train_zero);
            }
        }
    };
    auto stage = [&](int buf) {
        float* As = As0 + buf * 128 * AS;
        float* Xs = Xs0 + buf * XSZ;
#pragma unroll
        for (int u = 0; u < NA4; ++u) {
            float* d = As + ((tid >> 3) + 32 * u) * AS + a_p4;
            d[0] = ra[4 * u]; d[1] = ra[4 * u + 1]; d[2] = ra[4 * u + 2]; d[3] = ra[4 * u + 3];
        }
#pragma unroll
        for (int u = 0; u < NXMAX; ++u) {
            const int e = tid + u * 256;
            if (u < nx && e < KHT * 32 * xw) Xs[x_pk[u] >> 8] = rx[u];
        }
    };
    if (nchunks > 0) { fetch(0); stage(0); }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) fetch(ch + 1);
        const float* ap = As0 + buf * 128 * AS + (wave * 32 + lr) * AS + lh;
        const float* xp = Xs0 + buf * XSZ + lr * XS + lh * a.stride;
        // (fully unrolled: with a real inner loop hipcc drains the next chunk's global loads -- `s_waitcnt vmcnt(0)` -- in the loop's
        // preheader, i.e. BEFORE the MFMAs they are meant to hide behind)
#pragma unroll
        for (int s = 0; s < PX / 2; ++s) {
            const float av = ap[2 * s];
#pragma unroll
            for (int kyi = 0; kyi < KHT; ++kyi)
#pragma unroll
                for (int t = 0; t < KWT; ++t) {
                    const float bv = xp[kyi * 32 * XS + 2 * s * a.stride + t];
                    acc[kyi * KWT + t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[kyi * KWT + t], 0, 0, 0);
                }
        }
        if (ch + 1 < nchunks) stage(buf ^ 1);              // everyone finished reading buf ^ 1 before the previous barrier
        __syncthreads();
    }
    // partial[slab][m][c][ky][kx]; accumulator: col = lane%32 -> c, row = (r&3) + 8 (r>>2) + 4 (lane>>5) -> m
    float* P = a.part + (long long)slab * a.M * a.Cx * a.KH * a.KW;
    const int c = c0 + lr;
    if (c < a.Cx) {
#pragma unroll
        for (int kyi = 0; kyi < KHT; ++kyi)
#pragma unroll
            for (int t = 0; t < KWT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (m < a.M && t < a.KW) P[(((long long)m * a.Cx + c) * a.KH + ky + kyi) * a.KW + t] = acc[kyi * KWT + t][r];
                }
    }
}

// out[i] (+)= sum_s part[s][i], s ascending
__global__ void slab_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long long len, int slabs, int accumulate) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    float s = accumulate ? out[i] : 0.f;
    for (int k = 0; k < slabs; ++k) s += part[(long long)k * len + i];
    out[i] = s;
}

extern "C" long long dcvic_conv_wgrad_workspace_floats(int N, int M, int Cx, int KH, int KW, int Hg, int* slabs_out) {
    // slabs: whole images split into row groups so that the launch has >= ~512 workgroups
    const int mblocks = dcvic_cdiv(M, 128), cblocks = dcvic_cdiv(Cx, 32);
    const long long base = (long long)mblocks * cblocks * KH * N;
    int per_img = 1;
    while (base * per_img < 512 && per_img < Hg) per_img *= 2;   // (measured: 1 536 is slower on every trained shape -- shorter workgroups)
    per_img = per_img > Hg ? Hg : per_img;
    const int rows = dcvic_cdiv(Hg, per_img);
    per_img = dcvic_cdiv(Hg, rows);
    if (slabs_out) *slabs_out = N * per_img;
    return (long long)N * per_img * M * Cx * KH * KW;
}

extern "C" int dcvic_conv_wgrad_f32(const float* G, long long g_bs, int M, int Hg, int Wg, const float* X, long long x_bs, int Cx, int Hx,
                                    int Wx, int N, int KH, int KW, int stride, int pt, int pl, float* dW, int accumulate, float* workspace,
                                    void* stream) {
    DCVIC_CHECK_ARG(G && X && dW && workspace, "conv_wgrad: null pointer");
    DCVIC_CHECK_ARG(KW >= 1 && KW <= 5 && KH >= 1 && KH <= 5 && stride >= 1 && stride <= 4, "conv_wgrad: kernel %dx%d stride %d unsupported", KH, KW, stride);
    int slabs = 0;
    dcvic_conv_wgrad_workspace_floats(N, M, Cx, KH, KW, Hg, &slabs);
    WgradArgs a;
    a.G = G; a.g_bs = g_bs; a.M = M; a.Hg = Hg; a.Wg = Wg;
    a.X = X; a.x_bs = x_bs; a.Cx = Cx; a.Hx = Hx; a.Wx = Wx;
    a.part = workspace; a.N = N; a.KH = KH; a.KW = KW; a.stride = stride; a.pt = pt; a.pl = pl;
    a.slabs_per_img = slabs / N; a.rows_per_slab = dcvic_cdiv(Hg, a.slabs_per_img);
    a.mblocks = dcvic_cdiv(M, 128); a.cblocks = dcvic_cdiv(Cx, 32);
    // KHT = KH (all kernel rows of a 3x3 / stride-1 layer in one workgroup: G loaded once for nine taps, 144 MFMAs per barrier) is
    // built but measured SLOWER on every trained shape (0.58 / 1.79 / 3.42 ms against 0.53 / 1.77 / 3.13: 144 accumulator registers
    // leave one workgroup per CU or spills) -- opt-in for experiments only
    static const bool fuse_env = getenv("DCVIC_WGRAD_FUSED") && getenv("DCVIC_WGRAD_FUSED")[0] == '1';
    const bool rows_fused = KH == 3 && KW == 3 && stride == 1 && fuse_env;
    const long long blocks = (long long)a.mblocks * a.cblocks * (rows_fused ? 1 : KH) * slabs;
    DCVIC_CHECK_ARG(blocks < (1ll << 31), "conv_wgrad: grid too large");
    const int xw = 31 * stride + KW;
    const size_t lds = (size_t)2 * (128 * 33 + (rows_fused ? 3 : 1) * 32 * (xw | 1)) * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
    if (rows_fused) {
        static std::atomic<unsigned> attr_mask{0};
        if (DcvicAttrOnce once_{attr_mask})
            hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_kernel<3, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        conv_wgrad_kernel<3, 3><<<(unsigned)blocks, 256, lds, st>>>(a);
    } else switch (KW) {
        case 1: conv_wgrad_kernel<1, 1><<<(unsigned)blocks, 256, lds, st>>>(a); break;
        case 2: conv_wgrad_kernel<2, 1><<<(unsigned)blocks, 256, lds, st>>>(a); break;
        case 3: conv_wgrad_kernel<3, 1><<<(unsigned)blocks, 256, lds, st>>>(a); break;
        case 4: conv_wgrad_kernel<4, 1><<<(unsigned)blocks, 256, lds, st>>>(a); break;
        default: conv_wgrad_kernel<5, 1><<<(unsigned)blocks, 256, lds, st>>>(a); break;
    }
    DCVIC_CHECK_LAUNCH("conv_wgrad");
    const long long len = (long long)M * Cx * KH * KW;
    slab_reduce_kernel<<<dcvic_cdiv(len, 256), 256, 0, st>>>(workspace, dW, len, slabs, accumulate);
    DCVIC_CHECK_LAUNCH("conv_wgrad_reduce");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ per-channel reductions
// out[n][c] = sum_p a[n][c][p] * (b ? b[n][c][p] : 1)   (bias gradients, beta-FT scale / shift gradients); fp64 inside
__global__ __launch_bounds__(256) void chan_reduce_kernel(const float* __restrict__ a, long long a_bs, const float* __restrict__ b,
                                                          long long b_bs, float* __restrict__ out, int C, int HW) {
    __shared__ double red[16];
    const int n = blockIdx.x / C, c = blockIdx.x % C;
    const float* ap = a + (long long)n * a_bs + (long long)c * HW;
    const float* bp = b ? b + (long long)n * b_bs + (long long)c * HW : nullptr;
    double s = 0.0;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) s += bp ? (double)ap[i] * bp[i] : (double)ap[i];
    const double t = bsum_d(s, red);
    if (threadIdx.x == 0) out[(long long)n * C + c] = (float)t;
}
extern "C" int dcvic_chan_reduce_f32(const float* a, long long a_bs, const float* b, long long b_bs, float* out, int N, int C, int HW, void* stream) {
    DCVIC_CHECK_ARG(a && out && N > 0 && C > 0 && HW > 0, "chan_reduce: bad argument");
    chan_reduce_kernel<<<N * C, 256, 0, (hipStream_t)stream>>>(a, a_bs, b, b_bs, out, C, HW);
    DCVIC_CHECK_LAUNCH("chan_reduce");
    return DCVIC_OK;
}
// out[j] (+)= sum_i in[i][j], i ascending (sum over the batch of per-image partials)
extern "C" int dcvic_sum_rows_f32(const float* in, float* out, int rows, long long len, int accumulate, void* stream) {
    DCVIC_CHECK_ARG(in && out && rows > 0 && len > 0, "sum_rows: bad argument");
    slab_reduce_kernel<<<dcvic_cdiv(len, 256), 256, 0, (hipStream_t)stream>>>(in, out, len, rows, accumulate);
    DCVIC_CHECK_LAUNCH("sum_rows");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ elementwise backward forms
// op  0: d = g * act'(ref)      act in {relu, lrelu02, sigmoid, half_tanh: ref = OUTPUT y; swish, gelu: ref = INPUT x}
// op  1: d = g * a                                   (product rule pieces)
// op  2: d = g * sigmoid(a)                           NLAM: dt
// op  3: d = g * a * s(1-s), s = sigmoid(b)           NLAM: d(att)
// op  4: d = g * (1 + w * a)                          SFT: d(dec) given scale a
// op  5: d = w * g * a                                SFT: d(scale) given dec a
// op  6: d = w * g                                    SFT: d(shift) / plain scaling
// op  7: d = g * (1 + s[n][c])                        beta-FT: dx   (a = per-channel vector [N or 1][C], a_bs = C or 0)
// op  8: d = 2 * w * (a - b)                          MSE gradient (w = weight / numel)
// op  9: d = w * (sigmoid(a) - t)                     BCE-with-logits gradient (t = `act` as 0 / 1)
// op 10: d = a + b                                    gradient accumulation
// op 11: d = a * w
__global__ __launch_bounds__(256) void ew_bwd_kernel(int op, float* __restrict__ d, const float* __restrict__ g, const float* __restrict__ a,
                                                     const float* __restrict__ b, long long len, float w, int act, int C, int HW, long long vec_bs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    float r;
    switch (op) {
        case 0: {
            const float v = a[i];
            float dv;
            switch (act) {
                case DCVIC_ACT_RELU: dv = v > 0.f ? 1.f : 0.f; break;
                case DCVIC_ACT_LRELU02: dv = v > 0.f ? 1.f : 0.2f; break;
                case DCVIC_ACT_SIGMOID: dv = v * (1.f - v); break;
                case DCVIC_ACT_HALF_TANH: dv = 0.5f * (1.f - 4.f * v * v); break;            // y = tanh/2 -> dy/dx = (1 - tanh^2)/2
                case DCVIC_ACT_SWISH: { const float s = 1.f / (1.f + expf(-v)); dv = s * (1.f + v * (1.f - s)); break; }
                case DCVIC_ACT_GELU: { const float cdf = 0.5f * (1.f + erff(v * 0.70710678118654752440f));
                                       dv = cdf + v * 0.39894228040143267794f * expf(-0.5f * v * v); break; }
                default: dv = 1.f;
            }
            r = g[i] * dv; break;
        }
        case 1: r = g[i] * a[i]; break;
        case 2: r = g[i] / (1.f + expf(-a[i])); break;
        case 3: { const float s = 1.f / (1.f + expf(-b[i])); r = g[i] * a[i] * s * (1.f - s); break; }
        case 4: r = g[i] * (1.f + w * a[i]); break;
        case 5: r = w * g[i] * a[i]; break;
        case 6: r = w * g[i]; break;
        case 7: { const long long n = i / ((long long)C * HW); const int c = (int)((i / HW) % C); r = g[i] * (1.f + a[n * vec_bs + c]); break; }
        case 8: r = 2.f * w * (a[i] - b[i]); break;
        case 9: r = w * (1.f / (1.f + expf(-a[i])) - (float)act); break;
        case 10: r = a[i] + b[i]; break;
        default: r = a[i] * w;
    }
    d[i] = r;
}
extern "C" int dcvic_ew_bwd_f32(int op, float* d, const float* g, const float* a, const float* b, long long len, float w, int act, int C, int HW,
                                long long vec_bs, void* stream) {
    DCVIC_CHECK_ARG(d && len > 0 && op >= 0 && op <= 11, "ew_bwd: bad argument");
    ew_bwd_kernel<<<dcvic_cdiv(len, 256), 256, 0, (hipStream_t)stream>>>(op, d, g, a, b, len, w, act, C, HW, vec_bs);
    DCVIC_CHECK_LAUNCH("ew_bwd");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ GroupNorm backward
// y = act(xh * gamma + beta), xh = (x - mean) * rstd over the group.  One workgroup per (n, group):
//   dh = dy * act'(h);  dx = rstd * (dh*gamma - (S1 + xh * S2) / L),  S1 = sum dh*gamma, S2 = sum dh*gamma*xh
//   per-image partials dgamma[n][c] = sum_p dh * xh, dbeta[n][c] = sum_p dh   (summed over n by dcvic_sum_rows_f32)
// Three passes over the group (x | x, dy | x, dy -> dx) as 16-byte accesses with four loads in flight per thread: at batch 8 there are
// only N x 32 = 256 workgroups, one per CU, so the bytes in flight per thread decide the bandwidth (scalar loads: 1.4 TB/s).  The
// per-channel sums are wave-reduced into LDS as they finish and combined behind ONE barrier (two block reductions per channel before).
__global__ __launch_bounds__(512) void groupnorm_bwd_kernel(const float* __restrict__ x, long long x_bs, const float* __restrict__ dy,
                                                            long long dy_bs, float* __restrict__ dx, long long dx_bs,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ dgam, float* __restrict__ dbet, int C, int HW, int groups,
                                                            float eps, int act) {
    __shared__ double red[16];
    __shared__ double cred[8][64][2];                              // [wave][channel of the group][dh | dh * xh]
    const int n = blockIdx.x / groups, g = blockIdx.x % groups;
    const int cg = C / groups;
    const long long len = (long long)cg * HW;
    const float* xp = x + (long long)n * x_bs + (long long)g * cg * HW;
    const float* gp = dy + (long long)n * dy_bs + (long long)g * cg * HW;
    float* dp = dx + (long long)n * dx_bs + (long long)g * cg * HW;
    const bool vec = (HW % 4 == 0) && ((reinterpret_cast<uintptr_t>(xp) | reinterpret_cast<uintptr_t>(gp) | reinterpret_cast<uintptr_t>(dp)) % 16 == 0);
    const int B = blockDim.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = B >> 6;
    double s = 0.0, q = 0.0;
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        const long long n4 = len / 4;
        auto acc = [&](const float4 v) {
            s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            q += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
        };
        long long i = threadIdx.x;
        for (; i + 3LL * B < n4; i += 4LL * B) {
            const float4 v0 = x4[i], v1 = x4[i + B], v2 = x4[i + 2LL * B], v3 = x4[i + 3LL * B];
            acc(v0); acc(v1); acc(v2); acc(v3);
        }
        for (; i < n4; i += B) acc(x4[i]);
    } else {
        for (long long i = threadIdx.x; i < len; i += B) { const double v = xp[i]; s += v; q += v * v; }
    }
    const double S = bsum_d(s, red), Q = bsum_d(q, red);
    const double mean_d = S / (double)len;
    const float mean = (float)mean_d;
    const float var = (float)fmax(Q / (double)len - mean_d * mean_d, 0.0);
    const float rstd = 1.0f / sqrtf(var + eps);
    const bool swish = act == DCVIC_ACT_SWISH;
    auto dh1 = [&](float xv, float d, float ga, float be, float& xh) {
        xh = (xv - mean) * rstd;
        if (swish) { const float h = xh * ga + be; const float sg = 1.f / (1.f + expf(-h)); d *= sg * (1.f + h * (1.f - sg)); }
        return d;
    };
    // ---- pass 2: per channel sum dh, sum dh * xh
    for (int cc = 0; cc < cg; ++cc) {
        const int c = g * cg + cc;
        const float ga = gamma[c], be = beta[c];
        double a1 = 0.0, a2 = 0.0;
        auto acc1 = [&](float xv, float dv) { float xh; const float d = dh1(xv, dv, ga, be, xh); a1 += (double)d; a2 += (double)d * xh; };
        if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(xp + (long long)cc * HW);
            const float4* g4 = reinterpret_cast<const float4*>(gp + (long long)cc * HW);
            const int hw4 = HW / 4;
            auto acc4 = [&](const float4 xv, const float4 dv) { acc1(xv.x, dv.x); acc1(xv.y, dv.y); acc1(xv.z, dv.z); acc1(xv.w, dv.w); };
            int i = threadIdx.x;
            for (; i + B < hw4; i += 2 * B) {
                const float4 xa = x4[i], xb = x4[i + B], da = g4[i], db = g4[i + B];
                acc4(xa, da); acc4(xb, db);
            }
            for (; i < hw4; i += B) acc4(x4[i], g4[i]);
        } else {
            for (int i = threadIdx.x; i < HW; i += B) acc1(xp[(long long)cc * HW + i], gp[(long long)cc * HW + i]);
        }
        a1 = wsum_d(a1); a2 = wsum_d(a2);
        if (lane == 0) { cred[wv][cc][0] = a1; cred[wv][cc][1] = a2; }
    }
    __syncthreads();
    double s1 = 0.0, s2 = 0.0;
    for (int cc = 0; cc < cg; ++cc) {                              // (every thread: the same fixed order, no second barrier)
        double A1 = 0.0, A2 = 0.0;
        for (int w = 0; w < nw; ++w) { A1 += cred[w][cc][0]; A2 += cred[w][cc][1]; }
        const int c = g * cg + cc;
        if (threadIdx.x == 0) { dbet[(long long)n * C + c] = (float)A1; dgam[(long long)n * C + c] = (float)A2; }
        s1 += A1 * gamma[c]; s2 += A2 * gamma[c];
    }
    const float m1 = (float)(s1 / (double)len), m2 = (float)(s2 / (double)len);
    // ---- pass 3: dx
    auto dx1 = [&](float xv, float dv, float ga, float be) {
        float xh;
        const float d = dh1(xv, dv, ga, be, xh);
        return rstd * (d * ga - (m1 + xh * m2));
    };
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(xp);
        const float4* g4 = reinterpret_cast<const float4*>(gp);
        float4* d4 = reinterpret_cast<float4*>(dp);
        const int hw4 = HW / 4;
        const long long n4 = len / 4;
        int cc = (int)threadIdx.x / hw4, j = (int)threadIdx.x - cc * hw4;
        const int sc = B / hw4, sj = B - sc * hw4;
        auto put = [&](long long i, const float4 xv, const float4 dv, int ch) {
            const float ga = gamma[g * cg + ch], be = beta[g * cg + ch];
            d4[i] = make_float4(dx1(xv.x, dv.x, ga, be), dx1(xv.y, dv.y, ga, be), dx1(xv.z, dv.z, ga, be), dx1(xv.w, dv.w, ga, be));
        };
        auto step = [&]() { cc += sc; j += sj; if (j >= hw4) { j -= hw4; ++cc; } };
        long long i = threadIdx.x;
        for (; i + B < n4; i += 2LL * B) {
            const float4 xa = x4[i], xb = x4[i + B], da = g4[i], db = g4[i + B];
            const int c0 = cc; step(); const int c1 = cc; step();
            put(i, xa, da, c0); put(i + B, xb, db, c1);
        }
        for (; i < n4; i += B) { put(i, x4[i], g4[i], cc); step(); }
    } else {
        for (long long i = threadIdx.x; i < len; i += B) {
            const int ch = (int)(i / HW);
            dp[i] = dx1(xp[i], gp[i], gamma[g * cg + ch], beta[g * cg + ch]);
        }
    }
}
extern "C" int dcvic_groupnorm_bwd_f32(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dx, long long dx_bs,
                                       const float* gamma, const float* beta, float* dgamma_part, float* dbeta_part, int N, int C, int HW,
                                       int groups, float eps, int act, void* stream) {
    DCVIC_CHECK_ARG(x && dy && dx && gamma && beta && dgamma_part && dbeta_part, "groupnorm_bwd: null pointer");
    DCVIC_CHECK_ARG(C % groups == 0 && (act == DCVIC_ACT_NONE || act == DCVIC_ACT_SWISH), "groupnorm_bwd: C=%d groups=%d act=%d", C, groups, act);
    DCVIC_CHECK_ARG(C / groups <= 64, "groupnorm_bwd: %d channels per group (at most 64)", C / groups);
    groupnorm_bwd_kernel<<<N * groups, 512, 0, (hipStream_t)stream>>>(x, x_bs, dy, dy_bs, dx, dx_bs, gamma, beta, dgamma_part, dbeta_part, C, HW,
                                                                      groups, eps, act);
    DCVIC_CHECK_LAUNCH("groupnorm_bwd");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ channel LayerNorm backward
// y[c] = xh[c]*gamma[c] + beta[c] per pixel over C.  Thread per pixel (coalesced across the wave); dgamma / dbeta partials per
// workgroup [blocks][2][C] (reduced by dcvic_sum_rows_f32).
__global__ __launch_bounds__(256) void layernorm_c_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                              const float* __restrict__ gamma, float* __restrict__ part, int C, int HW,
                                                              int total, float eps) {
    extern __shared__ float sred[];        // [4 waves][2][C]
    const int gp = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = gp < total;
    const int n = ok ? gp / HW : 0, p = ok ? gp % HW : 0;
    const float* xp = x + (long long)n * C * HW + p;
    const float* gy = dy + (long long)n * C * HW + p;
    float mean = 0.f, rstd = 0.f, m1 = 0.f, m2 = 0.f;
    if (ok) {
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += xp[(long long)c * HW];
        mean = s / C;
        float v = 0.f;
        for (int c = 0; c < C; ++c) { const float d = xp[(long long)c * HW] - mean; v += d * d; }
        rstd = 1.f / sqrtf(v / C + eps);
        float a1 = 0.f, a2 = 0.f;
        for (int c = 0; c < C; ++c) {
            const float xh = (xp[(long long)c * HW] - mean) * rstd, d = gy[(long long)c * HW] * gamma[c];
            a1 += d; a2 += d * xh;
        }
        m1 = a1 / C; m2 = a2 / C;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c = 0; c < C; ++c) {
        float xh = 0.f, d = 0.f;
        if (ok) {
            xh = (xp[(long long)c * HW] - mean) * rstd; d = gy[(long long)c * HW];
            dx[(long long)n * C * HW + (long long)c * HW + p] = rstd * (d * gamma[c] - (m1 + xh * m2));
        }
        const float sg = wsum_f(d * xh), sb = wsum_f(d);
        if (lane == 0) { sred[(w * 2 + 0) * C + c] = sg; sred[(w * 2 + 1) * C + c] = sb; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float t = 0.f;
        for (int ww = 0; ww < 4; ++ww) t += sred[ww * 2 * C + i];
        part[(long long)blockIdx.x * 2 * C + i] = t;
    }
}
extern "C" int dcvic_layernorm_c_bwd_blocks(int N, int HW) { return dcvic_cdiv((long long)N * HW, 256); }
extern "C" int dcvic_layernorm_c_bwd_f32(const float* x, const float* dy, float* dx, const float* gamma, float* part, int N, int C, int HW,
                                         float eps, void* stream) {
    DCVIC_CHECK_ARG(x && dy && dx && gamma && part && C <= 1024, "layernorm_c_bwd: bad argument");
    const int blocks = dcvic_cdiv((long long)N * HW, 256);
    layernorm_c_bwd_kernel<<<blocks, 256, (size_t)8 * C * sizeof(float), (hipStream_t)stream>>>(x, dy, dx, gamma, part, C, HW, N * HW, eps);
    DCVIC_CHECK_LAUNCH("layernorm_c_bwd");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ column softmax backward
// P, dP: [N][C][Pn] with the softmax over C (the attention score matrix stored [key][query]).  dS = scale * P * (dP - sum_c P dP)
__global__ __launch_bounds__(256) void softmax_c_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dP, float* __restrict__ dS,
                                                            int C, int Pn, float scale) {
    const int n = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Pn) return;
    const long long base = (long long)n * C * Pn + p;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) dot += P[base + (long long)c * Pn] * dP[base + (long long)c * Pn];
    for (int c = 0; c < C; ++c) { const long long i = base + (long long)c * Pn; dS[i] = scale * P[i] * (dP[i] - dot); }
}
extern "C" int dcvic_softmax_c_bwd_f32(const float* P, const float* dP, float* dS, int N, int C, int Pn, float scale, void* stream) {
    DCVIC_CHECK_ARG(P && dP && dS && N > 0 && N <= 65535, "softmax_c_bwd: bad argument");
    dim3 grid(dcvic_cdiv(Pn, 256), N);
    softmax_c_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(P, dP, dS, C, Pn, scale);
    DCVIC_CHECK_LAUNCH("softmax_c_bwd");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ Swin window attention backward
// Forward (swin.hip / swinir_layers.py:118-148): per (image, window, head): S = (q*scale) k^T + bias[rel] (+ mask), P = softmax_j S,
// o = P v, on the cyclically shifted NCHW qkv map.  One workgroup (64 threads: thread = query token i) per (n, window, head).
// Outputs: dqkv [N][3C][H][W] (written at the un-shifted positions) and the per-(n, window) score gradients dSp[nw][head][64][64]
// from which dcvic_swin_bias_grad_f32 accumulates the relative-position table gradient in a fixed order.
__global__ __launch_bounds__(64) void swin_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout, float* __restrict__ dqkv,
                                                           const float* __restrict__ table, float* __restrict__ dSp, int C, int H, int W,
                                                           int heads, int ws, int shift) {
    constexpr int MAXD = 16;
    __shared__ float Ks[64 * MAXD], Vs[64 * MAXD], Qs[64 * MAXD], dOs[64 * MAXD], dSs[64 * 65];
    const int hd = C / heads, T = ws * ws;
    const int nWx = W / ws, nWy = H / ws;
    int b = blockIdx.x;
    const int head = b % heads; b /= heads;
    const int win = b % (nWx * nWy); const int n = b / (nWx * nWy);
    const int wy = win / nWx, wx = win % nWx;
    const int i = threadIdx.x;                 // token in window
    const int ty = i / ws, tx = i % ws;
    // shifted-frame coordinates -> source coordinates (torch.roll(x, -shift))
    const int sy = wy * ws + ty, sx = wx * ws + tx;
    const int yy = (sy + shift) % H, xx = (sx + shift) % W;
    const long long HWl = (long long)H * W;
    const long long pix = (long long)yy * W + xx;
    const float* base = qkv + (long long)n * 3 * C * HWl;
    const float scale = rsqrtf((float)hd);
    for (int d = 0; d < hd; ++d) {
        Qs[i * MAXD + d] = base[(long long)(head * hd + d) * HWl + pix];
        Ks[i * MAXD + d] = base[(long long)(C + head * hd + d) * HWl + pix];
        Vs[i * MAXD + d] = base[(long long)(2 * C + head * hd + d) * HWl + pix];
        dOs[i * MAXD + d] = dout[(long long)n * C * HWl + (long long)(head * hd + d) * HWl + pix];
    }
    __syncthreads();
    // region label of a shifted-frame position (swinir_layers.py:216-237)
    auto label = [&](int y, int x) {
        const int ry = y < H - ws ? 0 : (y < H - shift ? 1 : 2), rx = x < W - ws ? 0 : (x < W - shift ? 1 : 2);
        return ry * 3 + rx;
    };
    const int li = shift > 0 ? label(sy, sx) : 0;
    float Srow[64];
    float mx = -INFINITY;
    for (int j = 0; j < T; ++j) {
        float s = 0.f;
        for (int d = 0; d < hd; ++d) s = fmaf(Qs[i * MAXD + d] * scale, Ks[j * MAXD + d], s);
        const int jy = j / ws, jx = j % ws;
        const int rel = (ty - jy + ws - 1) * (2 * ws - 1) + (tx - jx + ws - 1);
        s += table[rel * heads + head];
        if (shift > 0 && label(wy * ws + jy, wx * ws + jx) != li) s += -100.f;
        Srow[j] = s; mx = fmaxf(mx, s);
    }
    float den = 0.f;
    for (int j = 0; j < T; ++j) { Srow[j] = expf(Srow[j] - mx); den += Srow[j]; }
    const float inv = 1.f / den;
    // dP[j] = dO_i . v_j ; dS = P (dP - sum P dP)
    float dPr[64];
    float dot = 0.f;
    for (int j = 0; j < T; ++j) {
        float a = 0.f;
        for (int d = 0; d < hd; ++d) a = fmaf(dOs[i * MAXD + d], Vs[j * MAXD + d], a);
        Srow[j] *= inv; dPr[j] = a; dot += Srow[j] * a;
    }
    float dq[MAXD];
    for (int d = 0; d < hd; ++d) dq[d] = 0.f;
    float* dSout = dSp + (((long long)(n * nWx * nWy + win)) * heads + head) * T * T;
    for (int j = 0; j < T; ++j) {
        const float ds = Srow[j] * (dPr[j] - dot);
        dSs[i * 65 + j] = ds;
        dSout[i * T + j] = ds;
        for (int d = 0; d < hd; ++d) dq[d] = fmaf(ds, Ks[j * MAXD + d], dq[d]);
    }
    // P is needed by the dV accumulation of other threads: keep it in Qs' place after q has been consumed -> use a second pass
    __syncthreads();
    float* dbase = dqkv + (long long)n * 3 * C * HWl;
    for (int d = 0; d < hd; ++d) dbase[(long long)(head * hd + d) * HWl + pix] = dq[d] * scale;
    // dK_i = sum_t dS[t][i] * q_t * scale ; dV_i = sum_t P[t][i] * dO_t  (thread i now acts as key / value token i)
    float dk[MAXD], dv[MAXD];
    for (int d = 0; d < hd; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    // P[t][i] recomputed from dS is not possible: stage P through LDS column by column
    __shared__ float Ps[64 * 65];
    for (int j = 0; j < T; ++j) Ps[i * 65 + j] = Srow[j];
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const float ds = dSs[t * 65 + i], pp = Ps[t * 65 + i];
        for (int d = 0; d < hd; ++d) {
            dk[d] = fmaf(ds, Qs[t * MAXD + d] * scale, dk[d]);
            dv[d] = fmaf(pp, dOs[t * MAXD + d], dv[d]);
        }
    }
    for (int d = 0; d < hd; ++d) {
        dbase[(long long)(C + head * hd + d) * HWl + pix] = dk[d];
        dbase[(long long)(2 * C + head * hd + d) * HWl + pix] = dv[d];
    }
}
// dtable[rel][head] (+)= sum over all (n, window) and over the (i, j) pairs of that relative offset.  One wave per table entry:
// lane l sums windows l, l + 64, ... (pairs ascending inside a window), then a fixed xor-shuffle tree -> deterministic.
__global__ __launch_bounds__(64) void swin_bias_grad_kernel(const float* __restrict__ dSp, float* __restrict__ dtable, int nwin_total, int heads, int ws, int accumulate) {
    const int T = ws * ws;
    const int id = blockIdx.x;
    const int rel = id / heads, head = id % heads;
    const int dy = rel / (2 * ws - 1) - (ws - 1), dx = rel % (2 * ws - 1) - (ws - 1);
    float s = 0.f;
    for (int w = threadIdx.x; w < nwin_total; w += 64) {
        const float* D = dSp + ((long long)w * heads + head) * T * T;
        for (int i = 0; i < T; ++i) {
            const int jy = i / ws - dy, jx = i % ws - dx;
            if (jy >= 0 && jy < ws && jx >= 0 && jx < ws) s += D[i * T + jy * ws + jx];
        }
    }
    s = wsum_f(s);
    if (threadIdx.x == 0) dtable[id] = (accumulate ? dtable[id] : 0.f) + s;
}
extern "C" int dcvic_swin_attn_bwd_f32(const float* qkv, const float* dout, float* dqkv, const float* table, float* dtable, float* dS_workspace,
                                       int N, int C, int H, int W, int heads, int ws, int shift, int accumulate, void* stream) {
    DCVIC_CHECK_ARG(qkv && dout && dqkv && table && dtable && dS_workspace, "swin_attn_bwd: null pointer");
    DCVIC_CHECK_ARG(ws == 8 && C % heads == 0 && C / heads <= 16 && H % ws == 0 && W % ws == 0, "swin_attn_bwd: ws=%d C=%d heads=%d", ws, C, heads);
    const int nwin = N * (H / ws) * (W / ws);
    swin_attn_bwd_kernel<<<nwin * heads, 64, 0, (hipStream_t)stream>>>(qkv, dout, dqkv, table, dS_workspace, C, H, W, heads, ws, shift);
    DCVIC_CHECK_LAUNCH("swin_attn_bwd");
    const int R = (2 * ws - 1) * (2 * ws - 1);
    swin_bias_grad_kernel<<<R * heads, 64, 0, (hipStream_t)stream>>>(dS_workspace, dtable, nwin, heads, ws, accumulate);
    DCVIC_CHECK_LAUNCH("swin_bias_grad");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ losses
// deterministic two-stage sums: stage 1 per-workgroup fp64 partials, stage 2 one workgroup in index order
__global__ __launch_bounds__(256) void loss_partial_kernel(int kind, const float* __restrict__ a, const float* __restrict__ b, long long len,
                                                           int t, double* __restrict__ part) {
    __shared__ double red[16];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (long long)gridDim.x * blockDim.x) {
        if (kind == 0) { const double d = (double)a[i] - (double)b[i]; s += d * d; }                 // squared error
        else if (kind == 1) { const float x = a[i]; s += (double)(fmaxf(x, 0.f) - x * (float)t + log1pf(expf(-fabsf(x)))); }   // BCE-with-logits
        else if (kind == 2) { const double v = a[i]; s += v * v; }                                  // sum of squares
        else s += (double)a[i];                                                                     // plain sum
    }
    const double T = bsum_d(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = T;
}
__global__ void loss_final_kernel(const double* __restrict__ part, int n, double scale, float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += part[i];
        out[0] = (float)(s * scale);
    }
}
// out[0] = scale * sum_i f(a[i], b[i] | t); kind 0: (a-b)^2, 1: BCE-with-logits(a, target t), 2: a^2, 3: a.  workspace: 1024 doubles
extern "C" int dcvic_reduce_loss_f32(int kind, const float* a, const float* b, long long len, int target, double scale, float* out,
                                     double* workspace, void* stream) {
    DCVIC_CHECK_ARG(a && out && workspace && len > 0 && kind >= 0 && kind <= 3, "reduce_loss: bad argument");
    const int blocks = (int)min((long long)1024, (long long)dcvic_cdiv(len, 256));
    loss_partial_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(kind, a, b, len, target, workspace);
    loss_final_kernel<<<1, 64, 0, (hipStream_t)stream>>>(workspace, blocks, scale, out);
    DCVIC_CHECK_LAUNCH("reduce_loss");
    return DCVIC_OK;
}
// cross entropy over the channel axis of logits [N][C][HW] against int64 targets [N][HW]:
//   nll[n][p] = logsumexp_c - logit[target];  dlogits = w * (softmax - onehot)   (w = weight / (N*HW))
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target, float* __restrict__ nll,
                                                 float* __restrict__ dlogits, int C, int HW, float w) {
    const int n = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* lp = logits + (long long)n * C * HW + p;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, lp[(long long)c * HW]);
    float den = 0.f;
    for (int c = 0; c < C; ++c) den += expf(lp[(long long)c * HW] - mx);
    const int t = (int)target[(long long)n * HW + p];
    nll[(long long)n * HW + p] = logf(den) + mx - lp[(long long)t * HW];
    if (dlogits) {
        float* dp = dlogits + (long long)n * C * HW + p;
        const float inv = 1.f / den;
        for (int c = 0; c < C; ++c) dp[(long long)c * HW] = w * (expf(lp[(long long)c * HW] - mx) * inv - (c == t ? 1.f : 0.f));
    }
}
extern "C" int dcvic_cross_entropy_f32(const float* logits, const int64_t* target, float* nll, float* dlogits, int N, int C, int HW, float w,
                                       void* stream) {
    DCVIC_CHECK_ARG(logits && target && nll && N > 0 && N <= 65535, "cross_entropy: bad argument");
    dim3 grid(dcvic_cdiv(HW, 256), N);
    ce_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(logits, target, nll, dlogits, C, HW, w);
    DCVIC_CHECK_LAUNCH("cross_entropy");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ optimizer
// torch.optim.Adam (no amsgrad, weight_decay 0): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).   gscale = gradient-clipping factor applied to g first.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   long long len, float lr, float b1, float b2, float eps, float bc1, float bc2s,
                                                   const float* __restrict__ gscale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const float gs = gscale ? gscale[0] : 1.f;
    const float gi = g[i] * gs;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
}
extern "C" int dcvic_adam_step_f32(float* p, const float* g, float* m, float* v, long long len, float lr, float beta1, float beta2, float eps,
                                   int step, const float* gscale, void* stream) {
    DCVIC_CHECK_ARG(p && g && m && v && len > 0 && step >= 1, "adam: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
    adam_kernel<<<dcvic_cdiv(len, 256), 256, 0, (hipStream_t)stream>>>(p, g, m, v, len, lr, beta1, beta2, eps, bc1, bc2s, gscale);
    DCVIC_CHECK_LAUNCH("adam");
    return DCVIC_OK;
}
// clip_grad_norm_: gscale[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))
__global__ void clip_scale_kernel(const float* __restrict__ sumsq, float max_norm, float* __restrict__ gscale) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { const float c = max_norm / (sqrtf(sumsq[0]) + 1e-6f); gscale[0] = c < 1.f ? c : 1.f; }
}
extern "C" int dcvic_clip_scale_f32(const float* sumsq, float max_norm, float* gscale, void* stream) {
    DCVIC_CHECK_ARG(sumsq && gscale, "clip_scale: null pointer");
    clip_scale_kernel<<<1, 64, 0, (hipStream_t)stream>>>(sumsq, max_norm, gscale);
    DCVIC_CHECK_LAUNCH("clip_scale");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ resampling
// nearest x2 upsample (model.py:53-57 F.interpolate) and its adjoint (sum of each 2x2 block)
__global__ __launch_bounds__(256) void resample2_kernel(int down, const float* __restrict__ in, float* __restrict__ out, long long planes, int H, int W) {
    // H, W: LOW resolution
    const long long total = down ? planes * H * W : planes * 4 * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (down) {
        const long long pl = i / ((long long)H * W); const int r = (int)(i % ((long long)H * W)); const int y = r / W, x = r % W;
        const float* ip = in + pl * 4 * H * W + (long long)(2 * y) * 2 * W + 2 * x;
        out[i] = (ip[0] + ip[1]) + (ip[2 * W] + ip[2 * W + 1]);
    } else {
        const long long pl = i / ((long long)4 * H * W); const int r = (int)(i % ((long long)4 * H * W)); const int y = r / (2 * W), x = r % (2 * W);
        out[i] = in[pl * H * W + (long long)(y >> 1) * W + (x >> 1)];
    }
}
extern "C" int dcvic_resample2_f32(int down, const float* in, float* out, long long planes, int Hlow, int Wlow, void* stream) {
    DCVIC_CHECK_ARG(in && out && planes > 0, "resample2: bad argument");
    const long long total = down ? planes * Hlow * Wlow : planes * 4 * Hlow * Wlow;
    resample2_kernel<<<dcvic_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(down, in, out, planes, Hlow, Wlow);
    DCVIC_CHECK_LAUNCH("resample2");
    return DCVIC_OK;
}

// ------------------------------------------------------------------------------------------------ LPIPS pieces
// (src/losses/perceptual_loss.py:10-30 -> lpips.LPIPS(net='alex'): AlexNet features, channel-normalise, squared difference,
//  learned non-negative 1x1 heads, spatial mean, sum over the 5 taps.  The package and its weights are not in the reference
//  tree: restated from the published architecture, parity unpinned.)
// space-to-depth with zero padding: in [P planes][H][W] -> out [P][r*r][Ho][Wo], out[p][dy*r+dx][y][x] = in[p][y*r+dy-pad][x*r+dx-pad];
// inverse = 1 is the adjoint (depth-to-space + crop).  The 11x11 / stride-4 / pad-2 stem of AlexNet becomes a 3x3 / stride-1
// convolution over 48 channels this way (kernel zero-padded to 12x12).
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ in, float* __restrict__ out, long long planes, int H, int W,
                                                  int r, int pad, int Ho, int Wo, int inverse) {
    if (!inverse) {
        const long long total = planes * r * r * Ho * Wo;
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= total) return;
        const int x = (int)(i % Wo); long long t = i / Wo;
        const int y = (int)(t % Ho); t /= Ho;
        const int d = (int)(t % (r * r)); const long long p = t / (r * r);
        const int iy = y * r + d / r - pad, ix = x * r + d % r - pad;
        out[i] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? in[(p * H + iy) * W + ix] : 0.f;
    } else {          // `in` is the depth tensor [P][r*r][Ho][Wo], `out` the image-side gradient [P][H][W]
        const long long total = planes * H * W;
        const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= total) return;
        const int ix = (int)(i % W); long long t = i / W;
        const int iy = (int)(t % H); const long long p = t / H;
        const int yy = iy + pad, xx = ix + pad;
        const int y = yy / r, x = xx / r, d = (yy % r) * r + xx % r;
        out[i] = (y < Ho && x < Wo) ? in[((p * r * r + d) * Ho + y) * Wo + x] : 0.f;
    }
}
extern "C" int dcvic_s2d_f32(const float* in, float* out, long long planes, int H, int W, int r, int pad, int inverse, void* stream) {
    DCVIC_CHECK_ARG(in && out && planes > 0 && r >= 1, "s2d: bad argument");
    const int Ho = (H + 2 * pad) / r, Wo = (W + 2 * pad) / r;
    const long long total = inverse ? planes * H * W : planes * r * r * Ho * Wo;
    s2d_kernel<<<dcvic_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in, out, planes, H, W, r, pad, Ho, Wo, inverse);
    DCVIC_CHECK_LAUNCH("s2d");
    return DCVIC_OK;
}
// MaxPool2d(3, stride 2), floor mode.  fwd: out + argmax position (0..8, first maximum); bwd (gather form, deterministic):
// dx[i] = sum over the <= 4 windows that contain i of dy[window] if that window's argmax is i.
__global__ __launch_bounds__(256) void maxpool3s2_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ am,
                                                         long long planes, int H, int W, int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * Ho * Wo) return;
    const int ox = (int)(i % Wo); long long t = i / Wo;
    const int oy = (int)(t % Ho); const long long p = t / Ho;
    const float* xp = x + p * H * W + (long long)(2 * oy) * W + 2 * ox;
    float best = xp[0]; int bi = 0;
#pragma unroll
    for (int k = 1; k < 9; ++k) { const float v = xp[(k / 3) * W + k % 3]; if (v > best) { best = v; bi = k; } }
    y[i] = best; am[i] = (unsigned char)bi;
}
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ am, float* __restrict__ dx,
                                                             long long planes, int H, int W, int Ho, int Wo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= planes * H * W) return;
    const int ix = (int)(i % W); long long t = i / W;
    const int iy = (int)(t % H); const long long p = t / H;
    float s = 0.f;
    for (int oy = max(0, (iy - 1) / 2); oy <= min(Ho - 1, iy / 2); ++oy)
        for (int ox = max(0, (ix - 1) / 2); ox <= min(Wo - 1, ix / 2); ++ox) {
            const int ky = iy - 2 * oy, kx = ix - 2 * ox;
            if (ky >= 0 && ky < 3 && kx >= 0 && kx < 3 && am[(p * Ho + oy) * Wo + ox] == ky * 3 + kx) s += dy[(p * Ho + oy) * Wo + ox];
        }
    dx[i] = s;
}
extern "C" int dcvic_maxpool3s2_f32(const float* x, float* y, unsigned char* argmax, const float* dy, float* dx, long long planes, int H, int W,
                                    void* stream) {
    DCVIC_CHECK_ARG(argmax && planes > 0 && H >= 3 && W >= 3, "maxpool: bad argument");
    const int Ho = (H - 3) / 2 + 1, Wo = (W - 3) / 2 + 1;
    if (dx) {
        DCVIC_CHECK_ARG(dy, "maxpool bwd: null dy");
        maxpool3s2_bwd_kernel<<<dcvic_cdiv(planes * H * W, 256), 256, 0, (hipStream_t)stream>>>(dy, argmax, dx, planes, H, W, Ho, Wo);
    } else {
        DCVIC_CHECK_ARG(x && y, "maxpool fwd: null pointer");
        maxpool3s2_kernel<<<dcvic_cdiv(planes * Ho * Wo, 256), 256, 0, (hipStream_t)stream>>>(x, y, argmax, planes, H, W, Ho, Wo);
    }
    DCVIC_CHECK_LAUNCH("maxpool3s2");
    return DCVIC_OK;
}
// One LPIPS tap: u = f / (||f||_c + 1e-10) per pixel for both feature maps, val[n] = (1/HW) sum_p sum_c w[c] (u0 - u1)^2  (per-pixel
// values in pix[n][p], reduced by dcvic_reduce_loss_f32 kind 3), and -- if df1 -- the gradient of (gscale * val) w.r.t. f1.
__global__ __launch_bounds__(256) void lpips_tap_kernel(const float* __restrict__ f0, const float* __restrict__ f1, const float* __restrict__ w,
                                                        float* __restrict__ pix, float* __restrict__ df1, int C, int HW, float gscale) {
    const int n = blockIdx.y, p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    const float* a = f0 + (long long)n * C * HW + p;
    const float* b = f1 + (long long)n * C * HW + p;
    float s0 = 0.f, s1 = 0.f;
    for (int c = 0; c < C; ++c) { const float x = a[(long long)c * HW], y = b[(long long)c * HW]; s0 += x * x; s1 += y * y; }
    const float r1 = sqrtf(s1), n0 = sqrtf(s0) + 1e-10f, n1 = r1 + 1e-10f;
    float val = 0.f, dot = 0.f;
    for (int c = 0; c < C; ++c) {
        const float d = a[(long long)c * HW] / n0 - b[(long long)c * HW] / n1;
        val += w[c] * d * d;
        dot += (-2.f * w[c] * d) * b[(long long)c * HW];          // sum_c a_c f1_c with a_c = dval/du1_c
    }
    pix[(long long)n * HW + p] = val;
    if (df1) {
        const float g = gscale / (float)HW;
        const float k2 = r1 > 0.f ? dot / (r1 * n1 * n1) : 0.f;
        float* o = df1 + (long long)n * C * HW + p;
        for (int c = 0; c < C; ++c) {
            const float y = b[(long long)c * HW];
            const float ac = -2.f * w[c] * (a[(long long)c * HW] / n0 - y / n1);
            o[(long long)c * HW] = g * (ac / n1 - y * k2);
        }
    }
}
extern "C" int dcvic_lpips_tap_f32(const float* f0, const float* f1, const float* w, float* pix, float* df1, int N, int C, int HW, float gscale,
                                   void* stream) {
    DCVIC_CHECK_ARG(f0 && f1 && w && pix && N > 0 && N <= 65535, "lpips_tap: bad argument");
    dim3 grid(dcvic_cdiv(HW, 256), N);
    lpips_tap_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(f0, f1, w, pix, df1, C, HW, gscale);
    DCVIC_CHECK_LAUNCH("lpips_tap");
    return DCVIC_OK;
}
