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
};


// defined in conv3x3.hip: returns DCVIC_OK after launching, or 1 if the layer is not eligible
int dcvic_try_conv3x3_dma(const ConvKArgs& K, int n_src, bool upsample, int cls, hipStream_t st, int* variant_out);

// defined in conv_async.hip: DMA double-buffered twin of conv_mfma_kernel<...,false> for small grids; same return convention
int dcvic_try_conv_async(const ConvKArgs& K, int cls, int P, hipStream_t st);

// defined in conv1x1.hip: DMA-pipelined GEMM for 1x1 / stride-1 layers with at least min_blocks workgroups; same return convention
int dcvic_try_conv1x1_dma(const ConvKArgs& K, int n_src, bool upsample, int cls, int min_blocks, hipStream_t st);

// defined in conv_async16.hip: the async twin on v_mfma_f32_16x16x4_f32 for the small tile variants; same return convention
int dcvic_try_conv_async16(const ConvKArgs& K, int cls, int P, hipStream_t st);
