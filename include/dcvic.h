/* dcvic.h -- C ABI of libdcvic_hip.so, the MI355X (gfx950) kernels behind the DC-VIC
 * compress / decompress path.
 *
 * The reference (iwa-shi/DC_VIC) has no FFI: its hot path is torch.nn modules selected through a
 * Python registry (src/utils/registry.py:73-92).  Each entry point below therefore names the
 * reference *operator* it replaces (file:line relative to the reference tree); INTEGRATION.md
 * shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions:
 *  - plain C types only; every pointer is a DEVICE pointer into caller-owned memory unless the
 *    name ends in _host; nothing is allocated or freed inside the library;
 *  - all tensors are fp32, NCHW, channel planes dense (H*W contiguous); a "view" may have a
 *    batch stride larger than C*H*W, which is how channel-concatenated buffers are expressed;
 *  - every launch goes to the hipStream_t passed in (as void*), no hidden synchronisation;
 *  - return 0 on success, a negative DCVIC_E* code otherwise; dcvic_last_error() gives a
 *    thread-local message;
 *  - kernels are deterministic and batch-invariant: the reduction order of every output element
 *    depends only on the layer description, never on N, the launch grid or timing (no atomics,
 *    no split-K) -- encoder and decoder must derive identical entropy parameters.
 */
#ifndef DCVIC_H
#define DCVIC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCVIC_OK 0
#define DCVIC_EINVAL (-1)   /* bad argument / unsupported shape */
#define DCVIC_ELAUNCH (-2)  /* HIP launch or runtime error */
#define DCVIC_ENOSPACE (-3) /* output buffer too small */
#define DCVIC_ECORRUPT (-4) /* bitstream does not decode */

#define DCVIC_MAX_TAPS 25
#define DCVIC_MAX_SRC 3

/* epilogue / elementwise activations */
enum {
    DCVIC_ACT_NONE = 0,
    DCVIC_ACT_RELU = 1,
    DCVIC_ACT_LRELU02 = 2,  /* LeakyReLU(0.2)  codeformer_layers.py:52-59 */
    DCVIC_ACT_SWISH = 3,    /* x*sigmoid(x)    ldm model.py:33-35, nn.SiLU */
    DCVIC_ACT_GELU = 4,     /* exact erf GELU  swinir_layers.py:17-33 */
    DCVIC_ACT_SIGMOID = 5,
    DCVIC_ACT_HALF_TANH = 6 /* 0.5*tanh(x)     minnen20_charm_context_model.py:107 */
};

const char* dcvic_last_error(void);
int dcvic_version(void);
/* Number of compute units / XCDs the library sized its grids for (queried once). */
int dcvic_device_info(int* n_cu, int* lds_bytes);

/* ------------------------------------------------------------------------------------------
 * Convolution family.  Replaces every torch.nn.Conv2d / ConvTranspose2d on the path:
 *   ldm/modules/diffusionmodules/model.py:42-79 (Upsample nearest x2 + conv, Downsample pad
 *   (0,1,0,1) + stride-2 conv), :82-141 ResnetBlock convs, :150-177 AttnBlock 1x1 convs;
 *   src/models/subnet/autoencoder/elic_autoencoder.py:21-28,42-52; src/models/layer/elic_layers.py:18-24;
 *   src/models/layer/cheng_nlam.py:32-37; src/models/subnet/hyperprior/minnen20_hyperprior.py:15-17,47-49;
 *   src/models/subnet/context_model/minnen20_charm_context_model.py:22-28;
 *   src/models/layer/codeformer_layers.py:26-31,49-59; src/models/layer/swinir_layers.py:22-24,110-112
 *   (nn.Linear == 1x1 conv on the NCHW token map); src/models/subnet/vq_estimator/swin_vq_estimator.py:38-68.
 *
 * out[n][co][oy*osy+ooy][ox*osx+oox] = epilogue( bias[co] +
 *        sum_{ci, t} Wt[t][ci][co] * IN[n][ci][oy*stride + dy[t]][ox*stride + dx[t]] )
 * with IN = the channel concatenation of up to 3 source views (zero outside the image), or its
 * nearest x2 upsampling when `upsample` is set.  A transposed convolution is issued as one
 * launch per output phase with osy=osx=2 (see dcvic_convT_phase_desc).
 * Reduction order per output element (a function of the LAYER only, never of N, sizes or the tile variant):
 * input-channel chunks of 8 ascending; inside a chunk taps t ascending; inside a tap the 8 channels ascending --
 * except the 3x3/stride-1/pad-1 family with Cin % 8 == 0, which runs (4-channel half, tap, channel) inside a chunk
 * (the pipeline-stage order of the LDS-DMA kernel); one fp32 FMA per term (MFMA 32x32x2 f32).
 */
typedef struct {
    int Cin, Cout;
    int T;                         /* number of taps */
    int8_t tap_ky[DCVIC_MAX_TAPS]; /* index into the source kernel [KH][KW] */
    int8_t tap_kx[DCVIC_MAX_TAPS];
    int8_t tap_dy[DCVIC_MAX_TAPS]; /* input offset of the tap */
    int8_t tap_dx[DCVIC_MAX_TAPS];
    int KH, KW;                    /* source kernel dims (for packing) */
    int stride;                    /* 1 or 2 (input step per output pixel) */
    int upsample;                  /* 0/1: virtual nearest x2 of the input */
    int transposed_weight;         /* 0: w[Cout][Cin][KH][KW]; 1: w[Cin][Cout][KH][KW] */
    int cfg;                       /* tile configuration id, filled by dcvic_conv_desc_finalize */
} dcvic_conv_desc;

typedef struct {
    const float* ptr;
    int C;
    long long batch_stride; /* elements */
} dcvic_src;

typedef struct {
    int N, H, W;       /* input planes (all sources) */
    int Hout, Wout;    /* output positions computed per phase */
    int Hfull, Wfull;  /* full output plane */
    int osy, osx, ooy, oox;
    int n_src;
    dcvic_src src[DCVIC_MAX_SRC];
    float* out;
    long long out_batch_stride;
    const float* bias;        /* [Cout] or NULL */
    int act;
    const float* res;         /* added after act, same plane geometry as out, or NULL */
    long long res_batch_stride;
    const float* aff_scale;   /* v = v*(1+scale[n][co]) + shift[n][co] after res, or NULL */
    const float* aff_shift;
    long long aff_batch_stride; /* 0 = same vector for every image */
    const float* init;        /* accumulators start from init[n][co][pix] (out geometry) instead of 0, or NULL:
                                 lets a reduction over input channels be split across launches bit-exactly */
    long long init_batch_stride;
} dcvic_conv_io;

/* Standard k x k convolution (pad_t/pad_l zeros; bottom/right padding is implied by Hout/Wout). */
int dcvic_conv_desc_init(dcvic_conv_desc* d, int Cin, int Cout, int KH, int KW, int stride,
                         int pad_t, int pad_l, int upsample);
/* One output phase (py,px in {0,1}) of ConvTranspose2d(k=5,s=2,p=2,output_padding=1), or the
 * whole ConvTranspose2d(k=3,s=1,p=1) when k==3 (py=px=0). */
int dcvic_convT_phase_desc(dcvic_conv_desc* d, int Cin, int Cout, int k, int py, int px);
/* Output-channel tile class (0: 128, 1: 64, 2: 32, 3: 96 channels per workgroup) that fills the chip best
 * for a launch of N x Hout x Wout outputs; write it to desc.cfg BEFORE packing/launching (packs are per class).
 * Every class produces bit-identical results (same reduction order). */
int dcvic_conv_select_class(const dcvic_conv_desc* d, int N, int Hout, int Wout);
/* Which kernel the calling thread's last dcvic_conv2d_f32 launched (profiling aid):
 * 9000 = conv3x3_dma_kernel, 8000 (8500 for the 16x16x4 build) + class*100 + pixels-per-tile/32 = conv_mfma_async(16)_kernel, 7000 + class = conv1x1_dma_kernel,
 * otherwise class*1000 + pixels-per-tile (+1 for the upsample loader). */
int dcvic_conv_last_variant(void);
/* Scheduling switches (A/B runs, tests): use_dma 0|1, use_async 0|1|2 (2 = async twin without its 16x16x4 build), async_fill = the async twin is used when a launch
 * has <= async_fill x CUs workgroups; -1 keeps a value.  Never changes results: every kernel computes the same
 * reduction order.  Debug / test aid, state PER CALLING THREAD (like dcvic_conv_last_variant); defaults come from
 * DCVIC_CONV_DMA / DCVIC_CONV_ASYNC / DCVIC_CONV_ASYNC16 / DCVIC_CONV_ASYNC_FILL, read once per thread.  Returns DCVIC_OK. */
int dcvic_conv_set_tuning(int use_dma, int use_async, int async_fill);
size_t dcvic_conv_packed_bytes(const dcvic_conv_desc* d);
int dcvic_conv_pack_f32(const dcvic_conv_desc* d, const float* w, float* packed, void* stream);
int dcvic_conv2d_f32(const dcvic_conv_desc* d, const float* packed, const dcvic_conv_io* io, void* stream);

/* Conv2d(k3, s1, p1) as Winograd F(2x2, 3x3) on fp32 MFMA (csrc/wino.hip): 4/9 of the multiplies of the direct sum,
 * re-associated -- results differ from dcvic_conv2d_f32 at the 1e-6 relative level, so it replaces ONLY the layers
 * after the last integer decision of the path: the frozen VQGAN decoder and the SFT fusion blocks
 *   ldm/modules/diffusionmodules/model.py:82-141 (ResnetBlock convs), :462-568 (Decoder);
 *   src/models/layer/codeformer_layers.py:20-67; src/models/subnet/vq_fusion_module.py:78-126.
 * Same io contract as dcvic_conv2d_f32 restricted to: Hout = Hfull = H, Wout = Wfull = W (W % 4 == 0), no scatter, every
 * source a multiple of 8 channels and 16-byte aligned, epilogue bias -> act -> (+res) (no affine, no init), out / res 16-byte aligned.
 * Weights: w[Cout][Cin][3][3] -> G g G^T in fp64, rounded once, packed per (64-channel tile, 8-channel chunk) as the
 * kernel's LDS image.  Deterministic and batch-invariant (fixed tile grid, ordered fmaf chains). */
size_t dcvic_wino_packed_bytes(int Cin, int Cout);
int dcvic_wino_pack_f32(const float* w, float* packed, int Cin, int Cout, void* stream);
int dcvic_conv3x3_wino_f32(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, void* stream);
/* Nearest-x2 upsample + Conv2d(k3, s1, p1) (ldm Upsample, model.py:42-57) as a structured Winograd: an output tile aligned to
 * the 2x2 upsample blocks sees input rows [a, b, b, c], whose transform has a zero third row / column -- 9 of the 16 positions,
 * 2.25 multiplies per output (the four 2x2 sub-pixel phases of dcvic_conv2d_f32 execute 4).  io: H x W = the LOW-resolution
 * input, Hout = Hfull = 2H, Wout = Wfull = 2W, W % 4 == 0; otherwise the contract of dcvic_conv3x3_wino_f32. */
size_t dcvic_wino_ups_packed_bytes(int Cin, int Cout);
int dcvic_wino_ups_pack_f32(const float* w, float* packed, int Cin, int Cout, void* stream);
int dcvic_conv3x3_wino_ups_f32(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, void* stream);
/* Conv2d(k3, s1, p1) as Winograd F(4x4, 3x3) on fp32 MFMA (csrc/wino44.hip): 36 multiplies per 4x4 outputs and input channel, 2.25
 * per output (F(2x2, 3x3): 4; the direct sum: 9), interpolation points 0, +-3/4, +-3/2, inf (all transform constants dyadic).  Its
 * rounding error is ~3x that of dcvic_conv3x3_wino_f32 (1.5e-6 rms relative per layer), so callers use it ONLY for the layers after
 * the path's last integer decision -- the frozen VQGAN decoder and the SFT fusion blocks (same reference operators as
 * dcvic_conv3x3_wino_f32: ldm/modules/diffusionmodules/model.py:82-141, 462-568; codeformer_layers.py:20-67;
 * vq_fusion_module.py:78-126) -- where it moves the reconstruction at the 1e-5 level (contract: 1e-3) and cannot touch indices or
 * bitstreams.  io contract of dcvic_conv3x3_wino_f32 (every source a multiple of 8 channels), activation none / ReLU / LeakyReLU(0.2).
 * Weights: G g G^T in fp64, rounded once, packed per (64-channel tile, 8-channel chunk) in the order the waves load them into registers.
 * Deterministic and batch-invariant. */
size_t dcvic_wino44_packed_bytes(int Cin, int Cout);
int dcvic_wino44_pack_f32(const float* w, float* packed, int Cin, int Cout, void* stream);
int dcvic_conv3x3_wino44_f32(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, void* stream);
/* The same convolution, additionally writing the GroupNorm statistics of its output: gn_part[N][Cout][n_pt][2] = per (image, output
 * channel, 16 x 32-pixel tile) the sum and the sum of squares of the values stored (n_pt = dcvic_wino44_stats_tiles(H, W)), so that the
 * GroupNorm that follows (ldm Normalize, model.py:38-39, inside ResnetBlock model.py:117-133) need not read the map for its statistics:
 * dcvic_groupnorm_part_f32.  Fixed summation order, no atomics. */
int dcvic_wino44_stats_tiles(int H, int W);
int dcvic_conv3x3_wino44_stats_f32(int Cin, int Cout, const float* packed, const dcvic_conv_io* io, float* gn_part, void* stream);
/* Conv2d(k3, s1, p1) with Cout <= 4 (Cin % 8 == 0) or Cin <= 4 (csrc/thin.hip): the VQGAN decoder's conv_out (128 -> 3,
 * ldm/modules/diffusionmodules/model.py:553-557) and the VQGAN encoder's conv_in (3 -> 128, model.py:388-392).  HBM-bound fp32
 * fmaf chains on the vector ALU in exactly the reduction order of dcvic_conv2d_f32, hence BIT-IDENTICAL to it; unpacked weights
 * w[Cout][Cin][3][3]; one source; epilogue bias -> act -> (+res).  dcvic_conv3x3_thin_applies: 1 if the layer qualifies. */
int dcvic_conv3x3_thin_applies(int Cin, int Cout);
int dcvic_conv3x3_thin_f32(const float* w, int Cin, int Cout, const dcvic_conv_io* io, void* stream);

/* ------------------------------------------------------------------------------------------
 * Batched strided GEMM  C[b][m][n] = alpha * sum_k A[b][m][k] * B[b][k][n]
 * Replaces torch.bmm in ldm AttnBlock (model.py:186-196).  Element strides; k ascending.
 */
typedef struct {
    int batch, M, N, K;
    const float* A; long long a_bs, a_ms, a_ks;
    const float* B; long long b_bs, b_ks, b_ns;
    float* C; long long c_bs, c_ms; /* C n-stride is 1 */
    float alpha;
} dcvic_gemm_args;
int dcvic_bgemm_f32(const dcvic_gemm_args* a, void* stream);

/* Fused single-head attention of the ldm AttnBlock (model.py:186-196: bmm, * c^-0.5, softmax(dim=2), bmm) on NCHW
 * planes: q, k, v [N][C][HW] (three channel ranges of one tensor are fine: common batch stride in_bs, plane rows
 * dense), out [N][C][HW].  out[c][i] = sum_j v[c][j] * softmax_j(scale * sum_c q[c][i] k[c][j]).  Online softmax over
 * 64-key tiles, scores never written to HBM; fp32 MFMA fmaf chains in a fixed order (batch- and grid-invariant).
 * HW % 64 == 0; C in {128, 256, 512}.  force_nw: 0 = choose, 2 / 4 = waves per workgroup (bit-identical; tests). */
int dcvic_attn_fused_f32(const float* q, const float* k, const float* v, long long in_bs, float* out, long long out_bs,
                         int N, int C, int HW, float scale, int force_nw, void* stream);

/* ------------------------------------------------------------------------------------------
 * Normalisation / softmax / elementwise
 */
/* GroupNorm(groups, eps, affine) + optional activation.  ldm model.py:38-39 (+ :33-35 swish),
 * femasr_layers.py:20-21, codeformer_layers.py:14-15.  Two-pass fp32 statistics per (n, group). */
int dcvic_groupnorm_f32(const float* x, long long x_bs, float* y, long long y_bs, const float* gamma,
                        const float* beta, int N, int C, int HW, int groups, float eps, int act, void* stream);
/* The same GroupNorm with the statistics pass replaced by the producer's partial sums (dcvic_conv3x3_wino44_stats_f32): part[N][C][n_pt][2],
 * added per (n, group) in fp64 in a fixed order -- one read + one write of the map instead of two reads + one write. */
int dcvic_groupnorm_part_f32(const float* x, long long x_bs, float* y, long long y_bs, const float* gamma,
                             const float* beta, int N, int C, int HW, int groups, float eps, int act,
                             const float* part, int n_pt, void* stream);
/* LayerNorm over the channel axis of an NCHW map (== nn.LayerNorm(C) on the [B, HW, C] token view,
 * swinir_layers.py:190,199), eps 1e-5. */
int dcvic_layernorm_c_f32(const float* x, float* y, const float* gamma, const float* beta, int N, int C,
                          int HW, float eps, void* stream);
/* Softmax over the channel axis of [N][C][P] (P contiguous).  Used for AttnBlock with the score
 * matrix stored [key][query] (model.py:188). */
int dcvic_softmax_c_f32(float* x, int N, int C, int P, void* stream);
/* Swin window attention (swinir_layers.py:118-148 + window_partition/roll 36-65, 249-277) on an NCHW
 * qkv map [N][3*C][H][W] (q,k,v thirds; head h owns channels h*hd..): output [N][C][H][W].
 * rel-pos bias table [(2ws-1)^2][heads]; shift>0 applies the cyclic shift and the -100 mask. */
int dcvic_swin_attn_f32(const float* qkv, float* out, const float* bias_table, int N, int C, int H, int W,
                        int heads, int ws, int shift, void* stream);
/* y = a + act(b) * c-style combiners used by the path:
 *   op 0: y = a + b                                   residuals
 *   op 1: y = a + b * sigmoid(c)                      ChengNLAM cheng_nlam.py:23-27
 *   op 2: y = a + w * (a * b + c)                     FuseSftBlock codeformer_layers.py:65-66
 *   op 3: y = a * (1 + s[n][ch]) + t[n][ch] (+ a2)    BetaScaleShiftModule elic_dual_beta_ft_autoencoder.py:45
 *   op 4: y = act(a)                                                                                     */
int dcvic_ew_f32(int op, float* y, long long y_bs, const float* a, long long a_bs, const float* b, long long b_bs,
                 const float* c, long long c_bs, int N, int C, int HW, float w, int act, void* stream);
int dcvic_chan_affine_f32(float* y, long long y_bs, const float* x, long long x_bs, const float* scale,
                          const float* shift, long long aff_bs, const float* add, long long add_bs,
                          int N, int C, int HW, void* stream);
/* Strided plane copy (concat materialisation, crops, reflect pad right/bottom base_model.py:156-163). */
int dcvic_copy_planes_f32(float* dst, long long dst_bs, int dstH, int dstW, const float* src, long long src_bs,
                          int srcH, int srcW, int N, int C, int copyH, int copyW, int reflect, void* stream);
/* General strided window copy dst[n][c][y][x] = src[n][c][y][x] for y<h, x<w (element strides; used
 * to cut / stitch the 512-px tiles of hyperprior_vic_model.py:190-246, 413-473). */
int dcvic_copy_window_f32(float* dst, long long dst_bs, long long dst_cs, long long dst_rs, const float* src,
                          long long src_bs, long long src_cs, long long src_rs, int N, int C, int h, int w, void* stream);
/* out[n] = max |x[n][...]| over C*HW elements (header field max_sample, codec_utils.py:18). */
int dcvic_absmax_f32(const float* x, long long x_bs, float* out, int N, long long CHW, void* stream);
/* Crop top-left + clamp(-1,1) (base_model.py:45-57) and optional truncating uint8 HWC RGB
 * ((x+1)/2*255 -> uint8, img_utils.py:19-44). */
int dcvic_crop_clamp_f32(const float* x, long long x_bs, int H, int W, float* y, uint8_t* y_u8, int N, int C,
                         int outH, int outW, void* stream);

/* ------------------------------------------------------------------------------------------
 * VQ nearest-codeword search.  Replaces VectorQuantizer2.forward, taming/modules/vqvae/quantize.py:271-312
 * (distance 280-282 in the expanded form sum(z^2)+sum(e^2)-2 z.e, argmin 284 first minimum,
 * gather 285, straight-through 298) plus F.one_hot (hyperprior_vic_model.py:268-271).
 * z [N][D][HW] NCHW; codebook [n_e][D]; idx int64 [N][HW]; zq [N][D][HW] (may be NULL);
 * feat [N][D+n_e][HW] = cat[zq, onehot] (may be NULL).
 */
int dcvic_vq_argmin_f32(const float* z, const float* codebook, int64_t* idx, float* zq, float* feat,
                        int N, int D, int HW, int n_e, void* stream);
/* argmax over channels (hyperprior_dc_vic_model.py:430, first maximum) + embedding gather
 * (hyperprior_vic_model.py:165-168) + post_quant_conv 1x1 (ldm/models/autoencoder.py:43). */
int dcvic_argmax_lut_f32(const float* logits, int64_t* idx, float* latent, const float* codebook,
                         const float* pq_w, const float* pq_b, int N, int n_e, int D, int HW, void* stream);

/* ------------------------------------------------------------------------------------------
 * Rate estimation + symbolisation.
 * GaussianConditional (CompressAI 1.2.4) as used by ste_gaussian_conditional.py:16-23,
 * minnen20_charm_context_model.py:96,148,164,199:
 *   sym = rint(y - mu); y_hat = sym + mu; p = max(Phi((.5-|sym|)/s) - Phi((-.5-|sym|)/s), 1e-9), s = max(sigma, .11)
 *   index = 63 - #{table[:-1] >= s}.  decode mode (y == NULL): y_hat = sym_in + mu.
 * bits[n] += -log2(p) summed per image in a fixed order (per-block fp64 partials, then ascending).
 */
int dcvic_gaussian_rate_f32(const float* y, long long y_bs, const int32_t* sym_in, const float* mu, const float* sigma,
                            long long ms_bs, const float* scale_table, int n_scales, float* y_hat, long long yh_bs,
                            int32_t* sym_out, int32_t* index_out, long long si_bs, float* lik_out, float* bits_out,
                            double* partial_ws /* N * dcvic_rate_blocks(C*HW) doubles, needed with bits_out */,
                            int N, int C, int HW, void* stream);
/* bits_out[n] = -sum(ln x[n][...]) / ln 2 over C*HW likelihoods (likelihood_to_bit, hyperprior_vic_model.py:80-82);
 * partial_ws: N * dcvic_rate_blocks(CHW) doubles. */
int dcvic_neglog2_sum_f32(const float* x, long long x_bs, float* bits_out, double* partial_ws, int N, long long CHW, void* stream);
/* Workgroups per image of the rate kernels (a function of C*HW only, so sums are batch-invariant). */
int dcvic_rate_blocks(long long CHW);
/* EntropyBottleneck eval forward (entropy_bottleneck.py:19-28 -> CompressAI, App-B):
 * z_hat = rint(z - med) + med; p = max(|sig(s*up) - sig(s*lo)|, 1e-9); symbols = rint(z - med).
 * decode mode (z == NULL): z_hat = float(sym_in) + med (EntropyBottleneck.decompress dequantisation). */
int dcvic_eb_rate_f32(const float* z, const int32_t* sym_in, const float* matrices, const float* biases,
                      const float* factors, const float* medians, float* z_hat, int32_t* sym_out, float* lik_out,
                      float* bits_out, int N, int C, int HW, void* stream);

/* ------------------------------------------------------------------------------------------
 * Host entropy coder (CPU, C++).  Replaces compressai.ans RansEncoder.encode_with_indexes /
 * RansDecoder.set_stream / decode_stream / decode_with_indexes and _CXX.pmf_to_quantized_cdf
 * (call sites hyperprior_charm_dc_vic_model.py:68,84; minnen20_charm_context_model.py:165,179-202;
 * hyperprior_dc_vic_model.py:66-68).  All pointers here are HOST pointers.
 */
int dcvic_pmf_to_quantized_cdf_host(const float* pmf, int n, int32_t* cdf_out /* n+1 */);
typedef struct dcvic_cdf_tables dcvic_cdf_tables;
dcvic_cdf_tables* dcvic_tables_create_host(const int32_t* cdfs, int n_cdf, int stride, const int32_t* sizes,
                                           const int32_t* offsets);
void dcvic_tables_destroy_host(dcvic_cdf_tables*);
/* Encode `n_streams` independent streams (one per image) in parallel on `threads` host threads.
 * symbols/indexes: [n_streams][n_sym] int32.  out: n_streams slots of out_cap bytes; out_len[i] set. */
int dcvic_rans_encode_batch_host(const dcvic_cdf_tables*, const int32_t* symbols, const int32_t* indexes,
                                 int n_streams, long long n_sym, uint8_t* out, long long out_cap,
                                 long long* out_len, int threads);
typedef struct dcvic_rans_decoder dcvic_rans_decoder;
dcvic_rans_decoder* dcvic_rans_decoder_create_host(const uint8_t* stream, long long nbytes);
void dcvic_rans_decoder_destroy_host(dcvic_rans_decoder*);
/* Batched incremental decode: decoders[i] consumes indexes[i][0..n_sym) -> symbols[i][..]. */
int dcvic_rans_decode_batch_host(const dcvic_cdf_tables*, dcvic_rans_decoder* const* decoders,
                                 const int32_t* indexes, int n_streams, long long n_sym, int32_t* symbols,
                                 int threads);

/* ==========================================================================================
 * Training step.  Replaces torch autograd / torch.optim on the reference's
 * src/trainer/dual_cond_gan_distortion_vq_code_trainer.py:135-300 path.  Data gradients of convolutions are
 * dcvic_conv2d_f32 launches on transposed / flipped weights; these are the remaining backward pieces.  All reductions
 * run in a fixed order (no atomics): ranks and reruns agree bit for bit.
 */
/* Weight gradient dW[m][c][ky][kx] = sum_{n,oy,ox} G[n][m][oy][ox] * X[n][c][oy*s+ky-pt][ox*s+kx-pl]  (fp32 MFMA, split over
 * pixel slabs, partials reduced in ascending slab order).  Conv2d: G = dY, X = input -> [Cout][Cin][KH][KW];
 * ConvTranspose2d(stride s): G = input, X = dY -> [Cin][Cout][KH][KW].  workspace: dcvic_conv_wgrad_workspace_floats(). */
long long dcvic_conv_wgrad_workspace_floats(int N, int M, int Cx, int KH, int KW, int Hg, int* slabs_out);
int dcvic_conv_wgrad_f32(const float* G, long long g_bs, int M, int Hg, int Wg, const float* X, long long x_bs, int Cx,
                         int Hx, int Wx, int N, int KH, int KW, int stride, int pt, int pl, float* dW, int accumulate,
                         float* workspace, void* stream);
/* out[n][c] = sum_p a[n][c][p] * (b ? b[n][c][p] : 1)  (bias / beta-FT vector gradients), fp64 accumulation. */
int dcvic_chan_reduce_f32(const float* a, long long a_bs, const float* b, long long b_bs, float* out, int N, int C, int HW, void* stream);
/* out[j] (+)= sum_i in[i][j], i ascending. */
int dcvic_sum_rows_f32(const float* in, float* out, int rows, long long len, int accumulate, void* stream);
/* Elementwise backward forms (op table in csrc/train.hip): activation derivatives, NLAM / SFT / beta-FT gates, MSE and
 * BCE-with-logits gradients, accumulation, scaling. */
int dcvic_ew_bwd_f32(int op, float* d, const float* g, const float* a, const float* b, long long len, float w, int act, int C, int HW,
                     long long vec_bs, void* stream);
/* GroupNorm(+swish) backward: dx and per-image partials dgamma_part / dbeta_part [N][C]. */
int dcvic_groupnorm_bwd_f32(const float* x, long long x_bs, const float* dy, long long dy_bs, float* dx, long long dx_bs,
                            const float* gamma, const float* beta, float* dgamma_part, float* dbeta_part, int N, int C, int HW,
                            int groups, float eps, int act, void* stream);
/* Channel LayerNorm backward: dx and per-workgroup partials part[blocks][2][C] (dgamma | dbeta). */
int dcvic_layernorm_c_bwd_blocks(int N, int HW);
int dcvic_layernorm_c_bwd_f32(const float* x, const float* dy, float* dx, const float* gamma, float* part, int N, int C, int HW,
                              float eps, void* stream);
/* Backward of the column softmax of dcvic_softmax_c_f32: dS = scale * P * (dP - sum_c P dP). */
int dcvic_softmax_c_bwd_f32(const float* P, const float* dP, float* dS, int N, int C, int Pn, float scale, void* stream);
/* Backward of dcvic_swin_attn_f32: dqkv and the relative-position table gradient (dS_workspace: N*windows*heads*64*64 floats). */
int dcvic_swin_attn_bwd_f32(const float* qkv, const float* dout, float* dqkv, const float* table, float* dtable, float* dS_workspace,
                            int N, int C, int H, int W, int heads, int ws, int shift, int accumulate, void* stream);
/* out[0] = scale * sum f: kind 0 (a-b)^2 (src/losses/distortion_loss.py), 1 BCE-with-logits(a, target) (gan_loss.py), 2 a^2
 * (clip_grad_norm_).  Two-stage fp64 sum in index order.  workspace: 1024 doubles. */
int dcvic_reduce_loss_f32(int kind, const float* a, const float* b, long long len, int target, double scale, float* out,
                          double* workspace, void* stream);
/* Cross entropy over the channel axis (src/losses/cross_entropy_loss.py): per-pixel nll and dlogits = w * (softmax - onehot). */
int dcvic_cross_entropy_f32(const float* logits, const int64_t* target, float* nll, float* dlogits, int N, int C, int HW, float w, void* stream);
/* torch.optim.Adam step on a flat parameter buffer; gscale (device scalar or NULL) multiplies the gradient first. */
int dcvic_adam_step_f32(float* p, const float* g, float* m, float* v, long long len, float lr, float beta1, float beta2, float eps,
                        int step, const float* gscale, void* stream);
/* gscale[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6))   (torch.nn.utils.clip_grad_norm_). */
int dcvic_clip_scale_f32(const float* sumsq, float max_norm, float* gscale, void* stream);
/* Nearest x2 upsample (down = 0) and its adjoint, the 2x2 block sum (down = 1), on `planes` planes of Hlow x Wlow. */
int dcvic_resample2_f32(int down, const float* in, float* out, long long planes, int Hlow, int Wlow, void* stream);

/* LPIPS(alex) pieces (src/losses/perceptual_loss.py:10-30; parity unpinned -- the lpips package is not in the reference tree):
 * zero-padded space-to-depth and its adjoint (the 11x11/s4/p2 stem as a 3x3 conv over 48 channels), MaxPool2d(3, 2) forward
 * (dx == NULL) / backward, and one tap: per-pixel sum_c w[c] (f0/|f0| - f1/|f1|)^2 plus the gradient w.r.t. f1. */
int dcvic_s2d_f32(const float* in, float* out, long long planes, int H, int W, int r, int pad, int inverse, void* stream);
int dcvic_maxpool3s2_f32(const float* x, float* y, unsigned char* argmax, const float* dy, float* dx, long long planes, int H, int W, void* stream);
int dcvic_lpips_tap_f32(const float* f0, const float* f1, const float* w, float* pix, float* df1, int N, int C, int HW, float gscale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DCVIC_H */
