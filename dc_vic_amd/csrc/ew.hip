// ew.hip -- elementwise combiners, per-channel affine (beta-FT), strided plane copies with reflect
// padding, crop+clamp(+uint8).  All HBM-bound; planes are dense so float4 is used when aligned.
#include "common.h"
#include <algorithm>

// y = f(a, b, c) over [N][C][HW] views with independent batch strides.
__global__ __launch_bounds__(256) void ew_kernel(int op, float* __restrict__ y, long long y_bs, const float* __restrict__ a,
                                                 long long a_bs, const float* __restrict__ b, long long b_bs,
                                                 const float* __restrict__ c, long long c_bs, long long CHW, float w, int act) {
    const int n = blockIdx.y;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < CHW; i += stride) {
        const float av = a[n * a_bs + i];
        float r;
        switch (op) {
            case 0: r = av + b[n * b_bs + i]; break;
            case 1: r = av + b[n * b_bs + i] * (1.f / (1.f + expf(-c[n * c_bs + i]))); break;
            case 2: r = av + w * (av * b[n * b_bs + i] + c[n * c_bs + i]); break;
            default: r = dcvic_act(av, act); break;
        }
        y[n * y_bs + i] = r;
    }
}

extern "C" int dcvic_ew_f32(int op, float* y, long long y_bs, const float* a, long long a_bs, const float* b, long long b_bs,
                            const float* c, long long c_bs, int N, int C, int HW, float w, int act, void* stream) {
    DCVIC_CHECK_ARG(y && a && N > 0 && C > 0 && HW > 0, "ew: bad argument");
    DCVIC_CHECK_ARG(op == 0 || op == 1 || op == 2 || op == 4, "ew: op %d", op);
    DCVIC_CHECK_ARG(op == 4 || b, "ew: op %d needs b", op);
    DCVIC_CHECK_ARG(!(op == 1 || op == 2) || c, "ew: op %d needs c", op);
    const long long CHW = (long long)C * HW;
    dim3 grid((unsigned)min((long long)2048, (CHW + 255) / 256), N);
    ew_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(op, y, y_bs, a, a_bs, b, b_bs, c, c_bs, CHW, w, act);
    DCVIC_CHECK_LAUNCH("ew");
    return DCVIC_OK;
}

// y = x * (1 + scale[n][c]) + shift[n][c] (+ add)      BetaScaleShiftModule.forward
__global__ __launch_bounds__(256) void chan_affine_kernel(float* __restrict__ y, long long y_bs, const float* __restrict__ x,
                                                          long long x_bs, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, long long aff_bs,
                                                          const float* __restrict__ add, long long add_bs, int HW) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float s = 1.f + scale[n * aff_bs + c], t = shift[n * aff_bs + c];
    const float* xp = x + n * x_bs + (long long)c * HW;
    float* yp = y + n * y_bs + (long long)c * HW;
    const float* ap = add ? add + n * add_bs + (long long)c * HW : nullptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float v = xp[i] * s + t;
        if (ap) v += ap[i];
        yp[i] = v;
    }
}

extern "C" int dcvic_chan_affine_f32(float* y, long long y_bs, const float* x, long long x_bs, const float* scale,
                                     const float* shift, long long aff_bs, const float* add, long long add_bs, int N, int C,
                                     int HW, void* stream) {
    DCVIC_CHECK_ARG(y && x && scale && shift && N > 0 && C > 0 && HW > 0, "chan_affine: bad argument");
    DCVIC_CHECK_ARG(C <= 65535 && N <= 65535, "chan_affine: grid too large");
    dim3 grid((unsigned)min(64, dcvic_cdiv(HW, 256)), C, N);
    chan_affine_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(y, y_bs, x, x_bs, scale, shift, aff_bs, add, add_bs, HW);
    DCVIC_CHECK_LAUNCH("chan_affine");
    return DCVIC_OK;
}

// dst[n][c][0:copyH][0:copyW] = src[n][c][ry][rx]; with reflect != 0 the source index reflects at the
// bottom/right edge (torch 'reflect': index srcH-2-(i-srcH) for i >= srcH), base_model.py:156-163.
__global__ __launch_bounds__(256) void copy_planes_kernel(float* __restrict__ dst, long long dst_bs, int dstH, int dstW,
                                                          const float* __restrict__ src, long long src_bs, int srcH, int srcW,
                                                          int C, int copyH, int copyW, int reflect) {
    const int n = blockIdx.z, c = blockIdx.y;
    const int total = copyH * copyW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int yy = i / copyW, xx = i - yy * copyW;
        int sy = yy, sx = xx;
        if (reflect) {
            if (sy >= srcH) sy = 2 * srcH - 2 - sy;
            if (sx >= srcW) sx = 2 * srcW - 2 - sx;
        }
        dst[n * dst_bs + ((long long)c * dstH + yy) * dstW + xx] = src[n * src_bs + ((long long)c * srcH + sy) * srcW + sx];
    }
}

extern "C" int dcvic_copy_planes_f32(float* dst, long long dst_bs, int dstH, int dstW, const float* src, long long src_bs,
                                     int srcH, int srcW, int N, int C, int copyH, int copyW, int reflect, void* stream) {
    DCVIC_CHECK_ARG(dst && src && N > 0 && C > 0, "copy_planes: bad argument");
    DCVIC_CHECK_ARG(copyH <= dstH && copyW <= dstW, "copy_planes: copy region exceeds destination");
    if (reflect) {
        DCVIC_CHECK_ARG(copyH <= 2 * srcH - 1 && copyW <= 2 * srcW - 1, "copy_planes: reflect pad larger than the image");
    } else {
        DCVIC_CHECK_ARG(copyH <= srcH && copyW <= srcW, "copy_planes: copy region exceeds source");
    }
    DCVIC_CHECK_ARG(C <= 65535 && N <= 65535, "copy_planes: grid too large");
    dim3 grid((unsigned)min(64, dcvic_cdiv((long long)copyH * copyW, 256)), C, N);
    copy_planes_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dst, dst_bs, dstH, dstW, src, src_bs, srcH, srcW, C, copyH, copyW, reflect);
    DCVIC_CHECK_LAUNCH("copy_planes");
    return DCVIC_OK;
}

// crop top-left + clamp(-1,1); optional uint8 HWC RGB with truncation ((x+1)/2*255 -> uint8).
__global__ __launch_bounds__(256) void crop_clamp_kernel(const float* __restrict__ x, long long x_bs, int H, int W,
                                                         float* __restrict__ y, uint8_t* __restrict__ y8, int C, int outH,
                                                         int outW) {
    const int n = blockIdx.y;
    const long long total = (long long)C * outH * outW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % outW);
        const int yy = (int)((i / outW) % outH);
        const int c = (int)(i / ((long long)outW * outH));
        float v = x[n * x_bs + ((long long)c * H + yy) * W + xx];
        v = fminf(fmaxf(v, -1.f), 1.f);
        if (y) y[n * total + i] = v;
        if (y8) {
            const float f = (v + 1.f) / 2.f * 255.f;   // img_utils.py:64-78 then astype(uint8)
            y8[((long long)n * outH * outW + (long long)yy * outW + xx) * C + c] = (uint8_t)(int)f;
        }
    }
}

extern "C" int dcvic_crop_clamp_f32(const float* x, long long x_bs, int H, int W, float* y, uint8_t* y_u8, int N, int C,
                                    int outH, int outW, void* stream) {
    DCVIC_CHECK_ARG(x && (y || y_u8) && N > 0 && C > 0, "crop_clamp: bad argument");
    DCVIC_CHECK_ARG(outH <= H && outW <= W && outH > 0 && outW > 0, "crop_clamp: crop %dx%d exceeds %dx%d", outH, outW, H, W);
    const long long total = (long long)C * outH * outW;
    dim3 grid((unsigned)min((long long)1024, (total + 255) / 256), N);
    crop_clamp_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, x_bs, H, W, y, y_u8, C, outH, outW);
    DCVIC_CHECK_LAUNCH("crop_clamp");
    return DCVIC_OK;
}

// general strided window copy
__global__ __launch_bounds__(256) void copy_window_kernel(float* __restrict__ dst, long long dst_bs, long long dst_cs,
                                                          long long dst_rs, const float* __restrict__ src, long long src_bs,
                                                          long long src_cs, long long src_rs, int h, int w) {
    const int n = blockIdx.z, c = blockIdx.y;
    const int total = h * w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int yy = i / w, xx = i - yy * w;
        dst[n * dst_bs + c * dst_cs + yy * dst_rs + xx] = src[n * src_bs + c * src_cs + yy * src_rs + xx];
    }
}

extern "C" int dcvic_copy_window_f32(float* dst, long long dst_bs, long long dst_cs, long long dst_rs, const float* src,
                                     long long src_bs, long long src_cs, long long src_rs, int N, int C, int h, int w,
                                     void* stream) {
    DCVIC_CHECK_ARG(dst && src && N > 0 && C > 0 && h > 0 && w > 0, "copy_window: bad argument");
    DCVIC_CHECK_ARG(dst_rs >= w && src_rs >= w, "copy_window: row stride smaller than the window");
    DCVIC_CHECK_ARG(C <= 65535 && N <= 65535, "copy_window: grid too large");
    dim3 grid((unsigned)min(64, dcvic_cdiv((long long)h * w, 256)), C, N);
    copy_window_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(dst, dst_bs, dst_cs, dst_rs, src, src_bs, src_cs, src_rs, h, w);
    DCVIC_CHECK_LAUNCH("copy_window");
    return DCVIC_OK;
}

// out[n] = max |x| over one image.  A maximum does not depend on the order it is taken in, so an image is split over several workgroups
// whose results meet in one atomicMax on the bit pattern (|x| >= 0: IEEE order = unsigned order): deterministic.  (One workgroup per
// image took 1.9 ms on the 2 M latents of a 1280x2048 image.)  `out` must hold zeros on entry (the launcher clears it).
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long long x_bs, float* __restrict__ out,
                                                     long long CHW, int parts) {
    __shared__ float red[4];
    const int n = blockIdx.x / parts, part = blockIdx.x % parts;
    float m = 0.f;
    for (long long i = (long long)part * blockDim.x + threadIdx.x; i < CHW; i += (long long)parts * blockDim.x) m = fmaxf(m, fabsf(x[n * x_bs + i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(out + n), __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

extern "C" int dcvic_absmax_f32(const float* x, long long x_bs, float* out, int N, long long CHW, void* stream) {
    DCVIC_CHECK_ARG(x && out && N > 0 && CHW > 0, "absmax: bad argument");
    const int parts = (int)std::min<long long>(256, std::max<long long>(1, CHW / 16384));
    if (hipMemsetAsync(out, 0, (size_t)N * sizeof(float), (hipStream_t)stream) != hipSuccess) { dcvic_set_error("absmax: memset failed"); return DCVIC_EINVAL; }
    absmax_kernel<<<N * parts, 256, 0, (hipStream_t)stream>>>(x, x_bs, out, CHW, parts);
    DCVIC_CHECK_LAUNCH("absmax");
    return DCVIC_OK;
}
