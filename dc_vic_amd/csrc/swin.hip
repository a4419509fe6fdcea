// swin.hip -- Swin window attention (W-MSA / SW-MSA) on NCHW maps.
// Replaces WindowAttention.forward + window_partition/reverse + torch.roll of SwinTransformerBlock.forward
// (src/models/layer/swinir_layers.py:36-65, 118-148, 249-277).  The cyclic shift, the window partition
// and their inverses are pure index arithmetic here -- no rolled or partitioned copy is materialised.
// One wave per (image, window, head): lane i owns query token i of the ws*ws (= 64) window tokens;
// K and V of the window sit in LDS; scores live in registers.
#include "common.h"

#define SWIN_HD 16
#define SWIN_TOK 64

__global__ __launch_bounds__(64) void swin_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                       const float* __restrict__ bias_table, int C, int H, int W, int heads,
                                                       int shift) {
    constexpr int ws = 8;
    __shared__ float Ks[SWIN_TOK][SWIN_HD + 1];
    __shared__ float Vs[SWIN_TOK][SWIN_HD + 1];
    __shared__ int ids[SWIN_TOK];
    const int lane = threadIdx.x;
    const int nWx = W / ws, nWy = H / ws;
    int b = blockIdx.x;
    const int head = b % heads; b /= heads;
    const int wx = b % nWx; b /= nWx;
    const int wy = b % nWy; b /= nWy;
    const int n = b;
    const int ly = lane >> 3, lx = lane & 7;
    // coordinates in the shifted frame, then in the original frame (roll by -shift)
    const int sy = wy * ws + ly, sx = wx * ws + lx;
    int oy = sy + shift, ox = sx + shift;
    if (oy >= H) oy -= H;
    if (ox >= W) ox -= W;
    const long long HW = (long long)H * W;
    const long long pix = (long long)oy * W + ox;
    const float* base = qkv + (long long)n * 3 * C * HW + pix;
    const float scale = 0.25f;  // head_dim ** -0.5 with head_dim 16 (swinir_layers.py:88-89)
    float q[SWIN_HD];
#pragma unroll
    for (int d = 0; d < SWIN_HD; ++d) {
        q[d] = base[(long long)(head * SWIN_HD + d) * HW] * scale;
        Ks[lane][d] = base[(long long)(C + head * SWIN_HD + d) * HW];
        Vs[lane][d] = base[(long long)(2 * C + head * SWIN_HD + d) * HW];
    }
    // region id of the shifted-window mask (swinir_layers.py:216-237)
    int rid = 0;
    if (shift > 0) {
        const int ry = sy < H - ws ? 0 : (sy < H - shift ? 1 : 2);
        const int rx = sx < W - ws ? 0 : (sx < W - shift ? 1 : 2);
        rid = ry * 3 + rx;
    }
    ids[lane] = rid;
    __syncthreads();
    float s[SWIN_TOK];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < SWIN_TOK; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < SWIN_HD; ++d) a = fmaf(q[d], Ks[j][d], a);
        const int jy = j >> 3, jx = j & 7;
        const int ridx = (ly - jy + ws - 1) * (2 * ws - 1) + (lx - jx + ws - 1);
        a += bias_table[ridx * heads + head];
        if (shift > 0) a += (ids[j] != rid) ? -100.0f : 0.0f;
        s[j] = a;
        m = fmaxf(m, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < SWIN_TOK; ++j) { s[j] = expf(s[j] - m); sum += s[j]; }
    const float inv = 1.0f / sum;
    float o[SWIN_HD];
#pragma unroll
    for (int d = 0; d < SWIN_HD; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < SWIN_TOK; ++j) {
        const float p = s[j] * inv;
#pragma unroll
        for (int d = 0; d < SWIN_HD; ++d) o[d] = fmaf(p, Vs[j][d], o[d]);
    }
    float* op = out + (long long)n * C * HW + pix;
#pragma unroll
    for (int d = 0; d < SWIN_HD; ++d) op[(long long)(head * SWIN_HD + d) * HW] = o[d];
}

extern "C" int dcvic_swin_attn_f32(const float* qkv, float* out, const float* bias_table, int N, int C, int H, int W,
                                   int heads, int ws, int shift, void* stream) {
    DCVIC_CHECK_ARG(qkv && out && bias_table && N > 0, "swin_attn: bad argument");
    DCVIC_CHECK_ARG(ws == 8 && C == heads * SWIN_HD, "swin_attn: only window 8 / head_dim 16 (got ws=%d C=%d heads=%d)", ws, C, heads);
    DCVIC_CHECK_ARG(H % ws == 0 && W % ws == 0, "swin_attn: %dx%d not a multiple of the window", H, W);
    DCVIC_CHECK_ARG(shift >= 0 && shift < ws, "swin_attn: shift %d", shift);
    const long long blocks = (long long)N * (H / ws) * (W / ws) * heads;
    DCVIC_CHECK_ARG(blocks < (1ll << 31), "swin_attn: grid too large");
    swin_attn_kernel<<<(unsigned)blocks, 64, 0, (hipStream_t)stream>>>(qkv, out, bias_table, C, H, W, heads, shift);
    DCVIC_CHECK_LAUNCH("swin_attn");
    return DCVIC_OK;
}
