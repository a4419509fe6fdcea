// Shared between conv.hip (generic implicit-GEMM kernel) and conv3x3.hip (DMA double-buffered 3x3 kernel).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define KC 8
#define MAXSLOT 6
#define NTHREADS 256
#define NXCD 8

struct ConvKArgs {
    int Cin, Cout, T, stride;
    int N, H, W, Hout, Wout, Hfull, Wfull, osy, osx, ooy, oox;
    const float* src[DCVIC_MAX_SRC];
    int srcC[DCVIC_MAX_SRC];
    long long src_bs[DCVIC_MAX_SRC];
    float* out;
    long long out_bs;
    const float* bias;
    int act;
    const float* res;
    long long res_bs;
    const float* affs;
    const float* afft;
    long long aff_bs;
    const float* wp;
    const float* init;        // accumulators start from init[n][co][pix] (same geometry as out) instead of 0
    long long init_bs;
    int TX, dy0, dx0, dstep;  // taps form a grid: t = iy*TX + ix, dy = dy0 + iy*dstep, dx = dx0 + ix*dstep
    int nslots;               // ceil(plane / NTHREADS)
    int TWlog, tiles_x, tiles_y;
    int PH, PW, plane, dy_min, dx_min;
    int TG, CPS, n_chunks, n_cotiles;
    int nblocks;
    int halves;               // 2: 3x3/stride-1 family order (4-channel half, tap, channel); 1: (tap, channel)
    float* gn_part;           // wino44 only: per (image, output channel, pixel tile) partial (sum, sum of squares) of the stored output, or null
};


#include <type_traits>
template <int I, int N, class F>
__device__ __forceinline__ void dcvic_static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); dcvic_static_for<I + 1, N>(f); }
}

// Epilogue of one accumulator group: NV values of one lane that share the output pixel `pix` and differ in the output
// channel co = co_of(r).  bias -> activation -> (+ residual) -> (beta-FT affine) -> store, in that order (the reference's
// layer order).  The optional inputs are fetched in BATCHES of NV independent loads under one uniform branch each: with
// the test inside the element loop hipcc branches around every load and waits vmcnt(0) per element -- measured on the
// 3x3 DMA kernel as 148 k cycles (10 % of a workgroup's life) of serialised ~2 us round trips.  Channels past Cout are
// clamped for the loads and masked for the store.
template <int NV, int B, bool RES, bool AFF, class AccT, class CoF>
__device__ __forceinline__ void dcvic_conv_epilogue(const ConvKArgs& K, int n, const AccT& accv, CoF co_of, long long pix, long long HWo) {
    // B = loads in flight per optional input: 8 where the register budget allows, 4 for the 128-accumulator tiles.
    // RES / AFF are resolved by the caller OUTSIDE its tile loops (dcvic_epilogue_dispatch), so that all optional inputs
    // of a batch are requested before the first one is used.
    static_assert(NV % B == 0, "group size");
    const int cmax = K.Cout - 1;
    float* const op = K.out + (long long)n * K.out_bs + pix;
    const float* const rp = RES ? K.res + (long long)n * K.res_bs + pix : nullptr;
    const long long ab = (long long)n * K.aff_bs;
#pragma unroll
    for (int r0 = 0; r0 < NV; r0 += B) {
        float v[B], bv[B], rv[B], sv[B], tv[B];
        if (K.bias) {
#pragma unroll
            for (int r = 0; r < B; ++r) bv[r] = K.bias[min(co_of(r0 + r), cmax)];
        } else {
#pragma unroll
            for (int r = 0; r < B; ++r) bv[r] = 0.f;
        }
        if constexpr (RES) {
#pragma unroll
            for (int r = 0; r < B; ++r) rv[r] = rp[(long long)min(co_of(r0 + r), cmax) * HWo];
        }
        if constexpr (AFF) {
#pragma unroll
            for (int r = 0; r < B; ++r) { const int c = min(co_of(r0 + r), cmax); sv[r] = K.affs[ab + c]; tv[r] = K.afft[ab + c]; }
        }
#pragma unroll
        for (int r = 0; r < B; ++r) {
            v[r] = accv[r0 + r];
            if (K.bias) v[r] += bv[r];                         // "+ 0" would turn -0 into +0: keep the no-bias path exact
            v[r] = dcvic_act(v[r], K.act);
            if constexpr (RES) v[r] += rv[r];
            if constexpr (AFF) v[r] = v[r] * (1.f + sv[r]) + tv[r];
        }
#pragma unroll
        for (int r = 0; r < B; ++r) {
            const int co = co_of(r0 + r);
            if (co <= cmax) op[(long long)co * HWo] = v[r];
        }
    }
}

// calls f(std::bool_constant<RES>, std::bool_constant<AFF>) for the launch's (residual?, affine?) combination
template <class F>
__device__ __forceinline__ void dcvic_epilogue_dispatch(const ConvKArgs& K, F&& f) {
    if (K.res) {
        if (K.affs) f(std::true_type{}, std::true_type{}); else f(std::true_type{}, std::false_type{});
    } else {
        if (K.affs) f(std::false_type{}, std::true_type{}); else f(std::false_type{}, std::false_type{});
    }
}

// defined in conv3x3.hip: returns DCVIC_OK after launching, or 1 if the layer is not eligible
int dcvic_try_conv3x3_dma(const ConvKArgs& K, int n_src, bool upsample, int cls, hipStream_t st, int* variant_out);

// defined in conv_async.hip: DMA double-buffered twin of conv_mfma_kernel<...,false> for small grids; same return convention
int dcvic_try_conv_async(const ConvKArgs& K, int cls, int P, hipStream_t st);

// defined in conv1x1.hip: DMA-pipelined GEMM for 1x1 / stride-1 layers with at least min_blocks workgroups; same return convention
int dcvic_try_conv1x1_dma(const ConvKArgs& K, int n_src, bool upsample, int cls, int min_blocks, hipStream_t st);

// defined in conv_async16.hip: the async twin on v_mfma_f32_16x16x4_f32 for the small tile variants; same return convention
int dcvic_try_conv_async16(const ConvKArgs& K, int cls, int P, hipStream_t st);
