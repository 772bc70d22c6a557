/* climategan_hip.h -- C ABI of libcgan_hip.so: the MI355X (gfx950) implementation of the ClimateGAN
 * generator/discriminator hot path.
 *
 * The reference (cc-ai/climategan) is pure Python on torch.nn: it has no FFI layer.  The drop-in boundary is
 * therefore the Python module API (climategan/generator.py, discriminator.py, blocks.py, norms.py,
 * painter.py); the modules in climategan_amd/ mirror that API and call THIS library through ctypes with raw
 * device pointers (tensor.data_ptr()).  Each entry point below names the reference code it replaces
 * (file:line, relative to the reference repo root).  See INTEGRATION.md for the binding a maintainer adds.
 *
 * Conventions
 *   - All pointers are DEVICE pointers owned by the caller; the library never allocates or frees device
 *     memory and keeps no reference after return.  Work is enqueued asynchronously on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream).
 *   - Activations are NHWC, 16-bit (dtype CGAN_F16 or CGAN_BF16), channel-padded: a tensor with C logical
 *     channels is stored with cgan_cs(C) = round_up(C, 8) channels per pixel and the pad channels are ZERO
 *     (every kernel here preserves that invariant).  Accumulation is fp32.
 *   - Weights arrive as fp32 in the reference's state-dict layout (OIHW) and are re-laid-out ("packed")
 *     into MFMA fragment order by the *_pack_* entry points; packed buffers are opaque.
 *   - Return value: 0 on success, negative cgan_status_t on failure; cgan_last_error() returns a
 *     thread-local message.  No exception or exit() crosses this boundary (the Python wrappers raise
 *     RuntimeError, mirroring the reference's ValueError/NotImplementedError for unsupported options,
 *     climategan/blocks.py:95-96,113-114, climategan/norms.py:156-160).
 */
#ifndef CLIMATEGAN_HIP_H
#define CLIMATEGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGAN_ABI_VERSION 1

typedef enum {
  CGAN_OK = 0,
  CGAN_ERR_BAD_ARG = -1,      /* null pointer, bad shape, unsupported option */
  CGAN_ERR_UNSUPPORTED = -2,  /* valid request this build has no kernel for */
  CGAN_ERR_WORKSPACE = -3,    /* workspace too small */
  CGAN_ERR_HIP = -4           /* HIP runtime error (launch failure ...) */
} cgan_status_t;

typedef enum { CGAN_F16 = 0, CGAN_BF16 = 1, CGAN_F32 = 2 /* cgan_allreduce_bucket only */ } cgan_dtype_t;
typedef enum { CGAN_ACT_NONE = 0, CGAN_ACT_RELU = 1, CGAN_ACT_LRELU = 2, CGAN_ACT_TANH = 3, CGAN_ACT_SIGMOID = 4 } cgan_act_t;
typedef enum { CGAN_PAD_ZERO = 0, CGAN_PAD_REFLECT = 1 } cgan_pad_t;

/* storage channels of a C-channel NHWC tensor */
static inline int cgan_cs(int c) { return (c + 7) & ~7; }

int cgan_version(void);
const char* cgan_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * conv2d (NHWC, implicit GEMM on MFMA 16x16x32)
 * Replaces nn.Conv2d as used by: SPADEResnetBlock conv_0/conv_1/conv_s (climategan/blocks.py:350-353,
 * 372-375,387-392), PainterSpadeDecoder.fc / conv_img (climategan/painter.py:50,111,152,166),
 * Conv2dBlock (climategan/blocks.py:117-139), NLayerDiscriminator / get_fc_discriminator convs
 * (climategan/discriminator.py:100-163,327-349).
 * y = act( conv(x, w) + bias + residual )
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t dtype;             /* cgan_dtype_t */
  int32_t n, h_in, w_in;     /* LOGICAL input extent (after the optional x2 nearest upsample) */
  int32_t c_in, c_out;       /* logical channels */
  int32_t kh, kw, stride, pad, dilation;
  int32_t pad_mode;          /* cgan_pad_t: zero (nn.ZeroPad2d / padding=) or reflect (nn.ReflectionPad2d) */
  int32_t h_out, w_out;      /* must equal floor((h_in + 2 pad - dil (k-1) - 1)/stride) + 1 */
  int32_t in_upsample;       /* 1: x is stored at (h_in/2, w_in/2) and read through nearest x2
                                (InterpolateNearest2d folded into the consumer, climategan/blocks.py:28-43) */
  int32_t act;               /* cgan_act_t applied last */
  float act_slope;           /* LeakyReLU slope (0.2 everywhere in the reference) */
  int32_t has_bias;
  int32_t has_residual;      /* add an NHWC tensor with c_out channels at (h_out, w_out) before act */
  int32_t residual_upsample; /* 1: residual stored at (h_out/2, w_out/2), read through nearest x2 */
} CganConvDesc;

size_t cgan_conv2d_packed_weight_bytes(const CganConvDesc* d);
/* w_oihw: fp32 [c_out][c_in][kh][kw].  sigma: optional device scalar; packed = w / *sigma (spectral norm's
 * w_bar / sigma, climategan/norms.py:112).  bias_out: fp32 [round_up(c_out,16)] zero-padded copy of bias
 * (bias may be NULL -> zeros). */
int cgan_conv2d_pack_weight(const float* w_oihw, const float* bias, const float* sigma, void* packed,
                            float* bias_out, const CganConvDesc* d, void* stream);
int cgan_conv2d_nhwc_fwd(const void* x, const void* packed_w, const float* bias_padded, const void* residual,
                         void* y, const CganConvDesc* d, void* stream);
/* Batched weight packing: one launch for `count` layers (e.g. the 23 spectral-norm convs of the Painter, whose
 * w_bar / sigma changes every forward).  `items_device` is a DEVICE array of CganPackItem; `max_fragments` =
 * max over layers of cgan_conv2d_packed_weight_bytes / 16. */
typedef struct {
  const float* w_oihw;   /* fp32 [c_out][c_in][kh][kw] */
  const float* bias;     /* fp32 [c_out] or NULL */
  const float* sigma;    /* device scalar or NULL */
  void* packed;          /* cgan_conv2d_packed_weight_bytes */
  float* bias_out;       /* fp32 [round_up(round_up(c_out,8),16)] */
  int32_t c_out, c_in, kh, kw;
  int32_t transposed;    /* 0: the forward operator; 1: the stride-1 data-gradient operator of the FORWARD weight
                          * [c_out][c_in][kh][kw] (rows = c_in, K channels = c_out, taps flipped) -- what
                          * cgan_conv2d_pack_weight_dgrad writes for a stride-1 descriptor; bias / bias_out unused */
} CganPackItem;
int cgan_conv2d_pack_weight_batched(const CganPackItem* items_device, int32_t count, int32_t dtype,
                                    int32_t max_fragments, void* stream);

/* Forward convolution that ALSO writes training-mode BatchNorm statistics of its own output from the kernel's epilogue
 * (fp32 accumulators, before the 16-bit store): replaces the read pass of cgan_batchnorm_train_stats over y for
 * nn.Conv2d -> nn.BatchNorm2d pairs in training mode (resnet101_v3.py:30-50, deeplab_v3.py:54-57, blocks.py:129-136).
 * cgan_conv2d_stats_chunk_pixels(d): pixels per statistics chunk of the kernel this descriptor selects, or 0 when that
 * kernel writes none (then use cgan_conv2d_nhwc_fwd + cgan_batchnorm_train_stats) -- no activation, no residual, no folded
 * upsample, n*h_out*w_out a whole number of chunks.  partial: fp32 [n*h_out*w_out / chunk][round_up(c_out,8)][2] =
 * (mean, M2) per chunk and channel, consumed by cgan_batchnorm_train_stats_from_partials. */
int32_t cgan_conv2d_stats_chunk_pixels(const CganConvDesc* d);
int cgan_conv2d_nhwc_fwd_stats(const void* x, const void* packed_w, const float* bias_padded, void* y, float* partial,
                               size_t partial_bytes, const CganConvDesc* d, void* stream);

/* Which kernel the forward (bwd_data = 0) or the data-gradient (bwd_data = 1) entry point runs for a descriptor --
 * the selection is a pure function of the descriptor: CGAN_CONV_KERNEL_GENERAL (gather implicit GEMM, conv_mfma.hip),
 * _LDS3X3 (spatially tiled 3x3, conv3x3_lds.hip) or _GEMM (wide-layer implicit GEMM, conv_gemm.hip); negative = the
 * descriptor is invalid.  Measurement aid: bench.py brackets the launches of one kernel family with events.
 * One run-time input besides the descriptor: the small-grid / long-K layers run as split-K launches of the GEMM kernel only
 * when the launch stream has a workspace bound that holds their partial tiles (cgan_conv2d_bind_workspace, below) -- so the
 * kernel, and with it the fp32 summation order, of such a layer can change with the batch size (partials outgrow the
 * workspace) and between callers that bind a workspace and callers that do not.  Results are bit-identical from run to run
 * for a fixed (descriptor, binding); cgan_conv2d_kernel_kind_on answers for the binding of one stream of the current
 * device, cgan_conv2d_kernel_kind for the largest workspace bound on the current device. */
enum { CGAN_CONV_KERNEL_GENERAL = 0, CGAN_CONV_KERNEL_LDS3X3 = 1, CGAN_CONV_KERNEL_GEMM = 2 };
/* Split-K scratch for the convolution entry points (round 5).  Layers with few output pixels and a long K -- the Painter's
 * 640-channel 3x3 convs at 5x5 .. 20x20 (climategan/painter.py:149-160), the discriminators' 512-channel 4x4 convs at
 * 20x20 / 10x10 (climategan/discriminator.py:130-163) -- cannot fill 256 CUs with (c_out block x pixel block) tiles; with a
 * workspace bound to the launch stream they run as K slices of the LDS-tiled GEMM whose fp32 partial tiles
 * ([slice][pixel][round_up(c_out,8)]) are summed in slice order by a second kernel (deterministic).  One buffer per stream
 * (launches of a stream are ordered; two streams must not share one), caller-owned, 16-byte aligned, registered once per
 * (current device, stream) -- the NULL stream of two devices are two bindings; workspace = NULL, bytes = 0 unbinds.  The
 * table holds 64 bindings; one more replaces the oldest (whose stream falls back to the general kernel: never an error).  Without a binding those layers run on the general kernel: same results up to the fp32
 * summation order, slower.  64 MiB covers every layer of the reference's default model at the benchmark batch sizes. */
int cgan_conv2d_bind_workspace(void* stream, void* workspace, size_t bytes);
int cgan_conv2d_kernel_kind(const CganConvDesc* d, int32_t bwd_data);
int cgan_conv2d_kernel_kind_on(const CganConvDesc* d, int32_t bwd_data, void* stream);
/* Backward of the convolution above (autograd of nn.Conv2d, reached from g_loss.backward() / d_loss.backward(),
 * climategan/trainer.py:1011,1028).  All take the FORWARD descriptor; act / bias / residual fields are ignored (their
 * backward is elementwise and lives in the callers).  bwd_data takes zero padding only: for a reflect-padded conv call it
 * with the pad-0 descriptor of the PADDED extent and fold the result with cgan_reflect_pad_bwd_nhwc; bwd_weight accepts
 * reflect padding directly.  With in_upsample, bwd_weight reads x through
 * the folded upsample and bwd_data returns the gradient at the LOGICAL (h_in, w_in) extent (follow with
 * cgan_sumpool2x2_nhwc to get the gradient of the stored tensor).
 *  - bwd_data:   dx[n][h_in][w_in][cgan_cs(c_in)] = conv_transpose(dy, w)  (rows / columns no window reached are zero);
 *                packed_w_dgrad comes from cgan_conv2d_pack_weight_dgrad (channel-transposed, tap-flipped, / *sigma).
 *  - bwd_weight: dw_oihw[c_out][c_in][kh][kw] += sum_pixels dy * x(shifted)   and   dbias[c_out] += sum_pixels dy
 *                (fp32, ACCUMULATED: zero them first; dbias may be NULL).  The pixel range is split over many
 *                workgroups; with a workspace (cgan_conv2d_bwd_weight_workspace_bytes, fp32 partial tiles) they are summed
 *                by a second kernel, with workspace == NULL every workgroup adds its tile to dw with fp32 atomics
 *                (same result up to summation order; several times slower on small-channel layers: cross-XCD atomics). */
size_t cgan_conv2d_dgrad_packed_weight_bytes(const CganConvDesc* fwd);
int cgan_conv2d_pack_weight_dgrad(const float* w_oihw, const float* sigma, void* packed, const CganConvDesc* fwd,
                                  void* stream);
int cgan_conv2d_nhwc_bwd_data(const void* dy, const void* packed_w_dgrad, void* dx, const CganConvDesc* fwd,
                              void* stream);
/* dx = (data gradient of the conv) + dx_add, for stride-1 'same' convolutions: the gradient that reaches the conv's INPUT
 * tensor through another consumer (the residual branch of a ResNet bottleneck, resnet101_v3.py:30-50) is added in the
 * conv kernel's epilogue instead of by a separate element-wise pass (autograd's gradient accumulation). */
int cgan_conv2d_nhwc_bwd_data_add(const void* dy, const void* packed_w_dgrad, const void* dx_add, void* dx,
                                  const CganConvDesc* fwd, void* stream);
/* dx = conv_transpose(dy, w) * [relu_out > 0]: the data gradient of a conv whose INPUT was the output of a ReLU (``relu_out``,
 * the forward input map itself, [n][h_in][w_in][cgan_cs(c_in)]), with that ReLU's derivative applied in the kernel's epilogue
 * -- the gradient w.r.t. the ReLU's input, without the separate activation-backward pass over the map (SPADE's mlp_shared
 * ReLU, norms.py:163-166; the VGG-19 ReLUs, losses.py:304-334).  Stride-1 'same' convolutions, like _bwd_data_add. */
int cgan_conv2d_nhwc_bwd_data_relu(const void* dy, const void* packed_w_dgrad, const void* relu_out, void* dx,
                                   const CganConvDesc* fwd, void* stream);
/* dx = [relu_out > 0] * (conv_transpose(dy, w) + dx_add): _bwd_data_add and _bwd_data_relu at once (round 6).  The first conv
 * of a ResNet bottleneck reads the previous block's output relu(bn3(.) + skip) (resnet101_v3.py:30-50) and hands it on to its
 * own block's skip branch: everything that flows back into that tensor passes through this call, so the ReLU's derivative is
 * taken here and the previous block's BatchNorm backward receives its gradient already masked (cgan_batchnorm_act_bwd_grouped
 * with act = none: one read of `out` and one write of the masked gradient less per bottleneck).  Same values as _bwd_data_add
 * followed by cgan_act_bwd(relu_out, dx, dx) -- which is what runs where the selected kernel has no fused form. */
int cgan_conv2d_nhwc_bwd_data_add_relu(const void* dy, const void* packed_w_dgrad, const void* dx_add, const void* relu_out,
                                       void* dx, const CganConvDesc* fwd, void* stream);
size_t cgan_conv2d_bwd_weight_workspace_bytes(const CganConvDesc* fwd);
int cgan_conv2d_nhwc_bwd_weight(const void* x, const void* dy, float* dw_oihw, float* dbias, const CganConvDesc* fwd,
                                void* workspace, size_t workspace_bytes,
                                void* stream);

/* ------------------------------------------------------------------------------------------------
 * Instance-norm statistics (biased variance over H*W per (n, c)), nn.InstanceNorm2d(affine=False,
 * track_running_stats=False): climategan/norms.py:151, climategan/discriminator.py:70-73.
 * mean, rstd: fp32 [n][cgan_cs(c)], rstd = 1/sqrt(var + eps).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t dtype;
  int32_t n, hw, c;
  float eps;
} CganNormStatsDesc;
size_t cgan_instnorm_stats_workspace_bytes(const CganNormStatsDesc* d);
int cgan_instnorm_stats(const void* x, float* mean, float* rstd, const CganNormStatsDesc* d, void* workspace,
                        size_t workspace_bytes, void* stream);
/* y = act((x - mean) * rstd) elementwise (instance norm + LeakyReLU of the PatchGAN,
 * climategan/discriminator.py:113-154) */
int cgan_norm_act_apply(const void* x, const float* mean, const float* rstd, void* y, const CganNormStatsDesc* d,
                        int32_t act, float act_slope, void* stream);
/* y = act((x - mean) * rstd + residual): the tail of a ResNet bottleneck, relu(bn3(conv3(.)) + residual)
 * (climategan/deeplab/resnet101_v3.py:43-72), in one pass; residual (same layout as x) may be NULL. */
int cgan_norm_add_act_apply(const void* x, const float* mean, const float* rstd, const void* residual, void* y,
                            const CganNormStatsDesc* d, int32_t act, float act_slope, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused SPADE (climategan/norms.py:146-186):
 *   seg  = nearest_resize(cond, (h, w))                                   norms.py:179
 *   actv = ReLU(conv3x3(seg; w_shared, b_shared))        [hidden 128]     norms.py:163-168,180
 *   y    = act( (x - mean) * rstd * (1 + conv3x3(actv; w_gamma)) + conv3x3(actv; w_beta) )   norms.py:181-184
 * The 128-channel hidden map lives only in LDS.  act = LeakyReLU(0.2) folds SPADEResnetBlock.activation
 * (climategan/blocks.py:372-373,394-395); act = NONE is the shortcut path (blocks.py:387-392).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  int32_t dtype;
  int32_t n, h, w, c;         /* x / y logical shape [n][h][w][c] */
  int32_t x_upsample;         /* 1: x stored at (h/2, w/2), read through nearest x2 */
  int32_t cond_h, cond_w, cond_c; /* cond NHWC [n][cond_h][cond_w][cgan_cond_cs(cond_c)] */
  int32_t hidden;             /* must be 128 (hard-coded in the reference, norms.py:163) */
  int32_t ksize;              /* must be 3 */
  int32_t act;                /* CGAN_ACT_NONE or CGAN_ACT_LRELU */
  float act_slope;
} CganSpadeDesc;
/* storage channels of the conditioning tensor: round_up(cond_c, 4) */
static inline int cgan_cond_cs(int c) { return (c + 3) & ~3; }
size_t cgan_spade_packed_weight_bytes(const CganSpadeDesc* d);
/* fp32 OIHW weights as in the state dict: mlp_shared.0.{weight[hidden][cond_c][3][3],bias},
 * mlp_gamma.{weight[c][hidden][3][3],bias}, mlp_beta.{...} */
int cgan_spade_pack_weights(const float* w_shared, const float* b_shared, const float* w_gamma, const float* b_gamma,
                            const float* w_beta, const float* b_beta, void* packed, const CganSpadeDesc* d,
                            void* stream);
int cgan_spade_fused_fwd(const void* x, const float* mean, const float* rstd, const void* cond, const void* packed,
                         void* y, const CganSpadeDesc* d, void* stream);
/* The same launch in training: gamma_out [n][h][w][cgan_cs(c)] (may be NULL = cgan_spade_fused_fwd) also receives the
 * modulation map gamma = mlp_gamma(actv) (climategan/norms.py:181, bias included, without the "1 +" of norms.py:184), which
 * the backward needs for d(normalized) = dout * (1 + gamma): 2 bytes per element written here instead of a 128 -> c
 * convolution over a re-materialised hidden map there. */
int cgan_spade_fused_fwd_train(const void* x, const float* mean, const float* rstd, const void* cond, const void* packed,
                               void* y, void* gamma_out, const CganSpadeDesc* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Spectral norm power iteration (climategan/norms.py:100-112), run on every forward:
 *   v <- l2n(W^T u); u <- l2n(W v); sigma <- u . (W v)        W = w_bar viewed [rows][cols]
 * u, v are updated IN PLACE (they are nn.Parameters of the wrapped conv in the reference).
 * workspace: fp32, cgan_spectral_norm_workspace_bytes(rows, cols).
 * ------------------------------------------------------------------------------------------------ */
size_t cgan_spectral_norm_workspace_bytes(int32_t rows, int32_t cols);
int cgan_spectral_norm_power_iter(const float* w_bar, float* u, float* v, float* sigma, int32_t rows, int32_t cols,
                                  void* workspace, size_t workspace_bytes, void* stream);
/* Batched: the power iteration of `count` layers in 4 launches (the reference runs one per wrapped conv per forward:
 * 23 in the Painter, norms.py:141-143).  `items_device` is a DEVICE array; each item's workspace holds
 * cgan_spectral_norm_workspace_bytes(rows, cols).  Bit-identical to the single-layer entry point. */
typedef struct {
  const float* w_bar;
  float* u;
  float* v;
  float* sigma;
  float* workspace;
  int32_t rows, cols;
} CganSnItem;
int cgan_spectral_norm_power_iter_batched(const CganSnItem* items_device, int32_t count, int32_t max_rows,
                                          int32_t max_cols, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ExtraAdam (climategan/optim.py:137-291), fused over all parameter tensors.
 *   mode 0 = Extragradient.extrapolation (optim.py:153-172): Adam moment update, [copy = p if save_copy], p += u
 *   mode 1 = Extragradient.step          (optim.py:174-197): Adam moment update, p = copy + u
 * `step` is the Adam step count AFTER this update (state["step"] is incremented by every update(), optim.py:268).
 * items_device: DEVICE array; all tensors fp32, contiguous.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
  float* p;
  const float* g;
  float* m;      /* exp_avg */
  float* v;      /* exp_avg_sq */
  float* copy;   /* parameters saved by the first extrapolation */
  int64_t numel;
} CganAdamItem;
int cgan_extra_adam_multi_tensor(const CganAdamItem* items_device, int32_t count, int64_t max_numel, int32_t mode,
                                 int32_t save_copy, int32_t step, double lr, double beta1, double beta2, double eps,
                                 double weight_decay, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Edge / glue kernels
 * ------------------------------------------------------------------------------------------------ */
/* fp32 NCHW [n][c][h][w] -> 16-bit NHWC [n][h][w][cs] (cs >= c, multiple of 4; pad channels zeroed).
 * If mask != NULL (fp32 [n][1][h][w]) the values are multiplied by (1 - mask): cond = x * (1 - m),
 * climategan/generator.py:294. */
int cgan_nchw_to_nhwc(const float* x, const float* mask, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h,
                      int32_t w, int32_t cs, void* stream);
/* 16-bit NHWC [n][h][w][cs] -> fp32 NCHW [n][c][h][w].  If paste_x/paste_m != NULL:
 * out = paste_x * (1 - m) + y * m   (climategan/generator.py:295-296) */
int cgan_nhwc_to_nchw(const void* y, const float* paste_x, const float* paste_m, float* out, int32_t dtype,
                      int32_t n, int32_t c, int32_t h, int32_t w, int32_t cs, void* stream);
/* nearest resize NHWC -> NHWC, legacy rule src = min(floor(dst * in/out), in-1) (F.interpolate
 * mode="nearest": climategan/painter.py:152, climategan/norms.py:179). cs_in/cs_out storage channels
 * (cs_out >= c; extra channels zeroed). */
int cgan_resize_nearest_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                             int32_t cs_in, int32_t h_out, int32_t w_out, int32_t cs_out, void* stream);
/* AvgPool2d(3, stride 2, padding 1, count_include_pad=False) on NHWC (climategan/discriminator.py:223-225) */
int cgan_avgpool3x3s2_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                           void* stream);

/* MaxPool2d(3, stride 2, padding 1) on NHWC (ResNet stem, climategan/deeplab/resnet101_v3.py:74,179) */
int cgan_maxpool3x3s2_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                           void* stream);
/* F.interpolate(mode="bilinear", align_corners=...) NHWC -> NHWC (climategan/deeplab/deeplab_v3.py:136-138,262-264,
 * climategan/blocks.py:300-302) */
int cgan_resize_bilinear_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                              int32_t h_out, int32_t w_out, int32_t align_corners, void* stream);
/* F.interpolate(mode="bicubic", align_corners=False) NHWC -> NHWC (climategan/depth.py:143-149, the MiDaS-size
 * re-sampling of the depth map when its width differs from the target) */
int cgan_resize_bicubic_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                             int32_t h_out, int32_t w_out, void* stream);
/* backward of the nearest x2 upsample (InterpolateNearest2d, climategan/blocks.py:28-43): x [n][2 h_out][2 w_out][cs]
 * -> y [n][h_out][w_out][cs], each output the sum of its 2x2 block */
int cgan_sumpool2x2_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_out, int32_t w_out,
                         void* stream);
/* backward of nn.ReflectionPad2d(pad) (Conv2dBlock's reflect padding, climategan/blocks.py:66-72): dx_padded
 * [n][h+2 pad][w+2 pad][cs] (the data gradient of the pad-0 convolution over the padded extent) -> dx [n][h][w][cs] */
int cgan_reflect_pad_bwd_nhwc(const void* dx_padded, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w,
                              int32_t pad, void* stream);
/* torch.cat along channels, one call per input: copies the c channels of src (pixel stride cs_src) into channels
 * [c_off, c_off + c) of dst (pixel stride cs_dst); c_off % 8 == 0 (deeplab_v3.py:107,139; blocks.py:311) */
int cgan_copy_channels_nhwc(const void* src, void* dst, int64_t npix, int32_t c, int32_t cs_src, int32_t cs_dst,
                            int32_t c_off, void* stream);
/* elementwise on NHWC storage: op 0: y = a * b (DADA fusion z * z_depth, deeplab_v3.py:253-254); op 1: y = sigmoid(a)
 * (climategan/generator.py:277); op 2: y = a * s, b pointing at ONE device fp32 scalar s (gradient of a loss term
 * times the upstream scalar gradient, e.g. the lambdas of climategan/trainer.py:1369-1380) */
int cgan_eltwise_nhwc(const void* a, const void* b, void* y, int32_t dtype, int32_t op, int64_t numel, void* stream);
/* fold an eval-mode BatchNorm2d into the preceding conv (fp32 weights [c_out][per_out]): w' = w s, b' = (b - mean) s + beta,
 * s = gamma / sqrt(var + eps)  (same algebra as climategan/bn_fusion.py:121-132) */
int cgan_fold_bn(const float* w, const float* bias, const float* gamma, const float* beta, const float* mean,
                 const float* var, float eps, float* w_out, float* b_out, int32_t c_out, int64_t per_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training path: backward of the elementwise / norm layers, losses with their gradients, spectral-norm gradient.
 * Loss entry points ACCUMULATE `weight * sum(...)` into a device fp32 scalar (zero it first; weight carries the
 * 1/N of the mean and any lambda) and write the gradient of that accumulated term.
 * ------------------------------------------------------------------------------------------------ */
/* dx = dy * act'(.) with act' expressed through the activation's OUTPUT `out` (nn.LeakyReLU / ReLU / tanh / sigmoid) */
int cgan_act_bwd(const void* out, const void* dy, void* dx, int32_t dtype, int32_t act, float act_slope,
                 int64_t numel, void* stream);
/* backward of out = act(instance_norm(x)) (climategan/discriminator.py:113-154: InstanceNorm2d + LeakyReLU(0.2)):
 * dx = rstd * (dz - mean_hw(dz) - y * mean_hw(dz * y)), dz = dy * act'(out), y = act^-1(out); act none or LeakyReLU.
 * workspace: cgan_instnorm_act_bwd_workspace_bytes(d) device bytes. */
size_t cgan_instnorm_act_bwd_workspace_bytes(const CganNormStatsDesc* d);
int cgan_instnorm_act_bwd(const void* out, const void* dy, const float* rstd, void* dx, const CganNormStatsDesc* d,
                          int32_t act, float act_slope, void* workspace, size_t workspace_bytes, void* stream);
/* elementwise stage of the SPADE backward (autograd of climategan/norms.py:181-186 and of the block's LeakyReLU):
 * with dz = dy * act'(y), xh = (x - mean) * rstd (x read through the folded upsample when d->x_upsample):
 *   dgb   [n][h][w][cgan_cs(2c)] : logical channels [dz * xh (c) | dz (c)]  = gradients of gamma and beta
 *   xhat  [n][h][w][cgan_cs(c)]  : xh
 *   dxhat [n][h][w][cgan_cs(c)]  : dz * (1 + gamma)   (gamma = the mlp_gamma conv output, bias included)
 * The conv gradients (mlp_gamma / mlp_beta / mlp_shared) and the instance-norm backward (cgan_instnorm_act_bwd on
 * xhat / dxhat) are separate calls. */
int cgan_spade_bwd_prepare(const void* dy, const void* y, const void* x, const float* mean, const float* rstd,
                           const void* gamma, void* dgb, void* xhat, void* dxhat, const CganSpadeDesc* d, void* stream);
/* Fused backward of SPADE's hidden map (round 5; autograd of climategan/norms.py:163-172,181-183 below gamma / beta): from
 * dgb (cgan_spade_bwd_prepare) the gradient of mlp_shared's weight [hidden][cond_c][3][3] and bias [hidden], ADDED to
 * dw_shared / db_shared (db_shared may be NULL), in one kernel per layer: the data gradient of the gamma||beta convolution
 * (packed_dgrad_gb = cgan_conv2d_pack_weight_dgrad of cat[w_gamma, w_beta] with the forward descriptor hidden -> 2c, 3x3,
 * pad 1), the ReLU mask from a hidden tile RE-COMPUTED from the conditioning image (packed_w_shared / bias_shared_padded =
 * cgan_conv2d_pack_weight of mlp_shared), and the contraction with the conditioning image's 3x3 neighbourhood -- neither the
 * hidden map nor its gradient touches HBM.  cond_hw: the conditioning image at the map's (h, w) (cond_h = h, cond_w = w),
 * NHWC with cgan_cs(cond_c) storage channels; cond_c <= 4 (the Painter's x (1 - m)); hidden = 128.  The conditioning image
 * receives no gradient here (callers that need one keep cgan_conv2d_nhwc_bwd_data_relu + the separate weight gradient).
 * workspace: cgan_spade_hidden_bwd_workspace_bytes(d) (per-workgroup partial blocks, summed in workgroup order). */
size_t cgan_spade_hidden_bwd_workspace_bytes(const CganSpadeDesc* d);
int cgan_spade_hidden_bwd(const void* dgb, const void* packed_dgrad_gb, const void* cond_hw, const void* packed_w_shared,
                          const float* bias_shared_padded, float* dw_shared, float* db_shared, void* workspace,
                          size_t workspace_bytes, const CganSpadeDesc* d, void* stream);
/* Training-mode nn.BatchNorm2d (+ ReLU / LeakyReLU) (ResNet-101, ASPP, depth decoder: climategan/deeplab/
 * resnet101_v3.py:30-50, deeplab_v3.py:54-57, depth.py:56-114).  Batch statistics = cgan_instnorm_stats on the tensor
 * viewed as one image of n*h*w pixels; then
 *  bn_train_prepare   folds gamma / beta into the (mean', rstd') pair cgan_norm_act_apply consumes and updates the
 *                     running statistics (momentum, unbiased variance) as nn.BatchNorm2d does; count = n*h*w;
 *                     num_batches_tracked (device int64 scalar, may be NULL) is incremented by one
 *  batchnorm_act_bwd  given x (the BN input), out = act(bn(x) [+ residual]) and dy: dx, and dgamma / dbeta
 *                     (out may be NULL when no residual was fused: act'(.) is then recomputed from x with fold_mean /
 *                     fold_rstd -- the mean_out / rstd_out rows the forward's statistics call wrote, i.e. the forward
 *                     apply's own arithmetic -- and the kernels read one map less per pass; with out given both may be NULL)
 *                     WRITTEN (fp32 [c], no zero fill needed; either may be NULL); dz_out (may be NULL) receives dz = dy * act'(out), the gradient of the
 *                     residual fused by cgan_norm_add_act_apply; workspace cgan_batchnorm_act_bwd_workspace_bytes(c) */
int cgan_bn_train_prepare(const float* batch_mean, const float* batch_rstd, const float* gamma, const float* beta,
                          float eps, float momentum, int64_t count, float* running_mean, float* running_var,
                          float* mean_out, float* rstd_out, int64_t* num_batches_tracked, int32_t c, void* stream);
/* cgan_instnorm_stats + cgan_bn_train_prepare in one call (the prepare step rides in the statistics' finalize kernel):
 * d describes the batch as ONE image (n = 1, hw = n*h*w); workspace cgan_instnorm_stats_workspace_bytes(d). */
int cgan_batchnorm_train_stats(const void* x, const float* gamma, const float* beta, float momentum, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float* batch_mean, float* batch_rstd,
                               float* mean_out, float* rstd_out, const CganNormStatsDesc* d, void* workspace,
                               size_t workspace_bytes, void* stream);
/* The finalize half of cgan_batchnorm_train_stats on the per-chunk (mean, M2) rows a convolution's epilogue wrote
 * (cgan_conv2d_nhwc_fwd_stats): d->n = groups, d->hw = pixels per group (a whole number of chunk_pixels), same outputs,
 * running-statistics updates and step counter as cgan_batchnorm_train_stats.  The partial rows are CONSUMED: lists of 512
 * rows and more per group are shortened in place (256 rows -> 1, fixed merge order) before the per-channel walk. */
int cgan_batchnorm_train_stats_from_partials(float* partial, int32_t chunk_pixels, const float* gamma,
                                             const float* beta, float momentum, float* running_mean, float* running_var,
                                             int64_t* num_batches_tracked, float* batch_mean, float* batch_rstd,
                                             float* mean_out, float* rstd_out, const CganNormStatsDesc* d, void* stream);
size_t cgan_batchnorm_act_bwd_workspace_bytes(int32_t c);   /* per group */
int cgan_batchnorm_act_bwd(const void* x, const void* out, const void* dy, const float* batch_mean,
                           const float* batch_rstd, const float* gamma, const float* fold_mean, const float* fold_rstd,
                           void* dx, float* dgamma, float* dbeta, void* dz_out, int32_t dtype, int64_t npix, int32_t c,
                           int32_t act, float act_slope, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the batch split into `groups` equal slices that are normalised independently (cgan_batchnorm_train_stats
 * with d->n = groups): what the reference does when it passes the real and the simulated domain batch through the
 * Masker in separate forward calls (trainer.py:1200-1254) -- here one launch sequence over the concatenated batch.
 * batch_mean / batch_rstd: [groups][cgan_cs(c)]; dgamma / dbeta: summed over the groups; npix_total = all pixels;
 * workspace: groups * cgan_batchnorm_act_bwd_workspace_bytes(c). */
int cgan_batchnorm_act_bwd_grouped(const void* x, const void* out, const void* dy, const float* batch_mean,
                                   const float* batch_rstd, const float* gamma, const float* fold_mean,
                                   const float* fold_rstd, void* dx, float* dgamma, float* dbeta, void* dz_out,
                                   int32_t dtype, int64_t npix_total, int32_t c, int32_t groups, int32_t act,
                                   float act_slope, void* workspace, size_t workspace_bytes, void* stream);
/* nn.BCEWithLogitsLoss(x, target) pieces against a constant target (GANLoss, climategan/losses.py:50-83; ADVENT
 * D-side BCE, losses.py:461-477) over the c logical channels of x [npix][cgan_cs(c)]:
 * *loss_accum += weight * sum(max(x,0) - x t + log1p(exp(-|x|))), dx = weight * (sigmoid(x) - t); dx may be NULL. */
int cgan_bce_logits_nhwc(const void* x, int32_t dtype, int64_t npix, int32_t c, float target, float weight,
                         float* loss_accum, void* dx, void* stream);
/* HingeLoss.loss (climategan/losses.py:565-579; selected by gen.p.loss == "hinge", losses.py:381-383) over the c logical
 * channels of x [npix][cgan_cs(c)]: for_discriminator: *loss_accum += weight * sum(-min(+-x - 1, 0)) (+ for a real
 * target, - for a fake one), dx = weight * d/dx (half on an exact tie, torch.min's rule); generator side (target must
 * be real, else negative rc like the reference's assert): *loss_accum += weight * sum(-x), dx = -weight. dx may be NULL. */
int cgan_hinge_nhwc(const void* x, int32_t dtype, int64_t npix, int32_t c, int32_t target_is_real,
                    int32_t for_discriminator, float weight, float* loss_accum, void* dx, void* stream);
/* nn.MSELoss against a constant target over the c logical channels of x [npix][cgan_cs(c)]: the LSGAN form of GANLoss
 * (use_lsgan=True, climategan/losses.py:50-52): *loss_accum += weight * sum (x - target)^2, dx = 2 weight (x - target). */
int cgan_mse_const_nhwc(const void* x, int32_t dtype, int64_t npix, int32_t c, float target, float weight, float* loss_accum,
                        void* dx, void* stream);
/* The Painter's optional image-space terms of get_painter_loss (climategan/trainer.py:1289-1315; TVLoss losses.py:142-169,
 * ContextLoss :281-287, ReconstructionLoss :290-296) on the pasted image p = x (1 - m) + fake m (fake NHWC 3 channels stored
 * as 8; x [n,3,h,w], m [n,1,h,w] fp32): loss3[0] += w_tv_h sum_y (q[y+1] - q[y])^2 + w_tv_w sum_x (q[x+1] - q[x])^2 with
 * q = p m; loss3[1] += w_context sum |(p - x)(1 - m)|; loss3[2] += w_reconstruction sum |(p - x) m|; d_fake (may be NULL) =
 * the gradient of their sum w.r.t. fake (gather form: no atomics).  The weights carry lambda and the means' 1 / count. */
int cgan_painter_aux_losses(const void* fake_nhwc, const float* x_nchw, const float* m_nchw, int32_t dtype, int32_t n,
                            int32_t h, int32_t w, float w_tv_h, float w_tv_w, float w_context, float w_reconstruction,
                            float* loss3, void* d_fake, void* stream);
/* Adjoint of cgan_resize_bicubic_nhwc (the DADA depth decoder's 384^2 resize under autograd, climategan/depth.py:143-149):
 * dx [n][h_in][w_in][cgan_cs(c)] from dy [n][h_out][w_out][cgan_cs(c)], gather form. */
int cgan_resize_bicubic_bwd_nhwc(const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                                 int32_t h_out, int32_t w_out, void* stream);
/* nn.L1Loss pieces (FeatMatchLoss, climategan/losses.py:86-103): *loss_accum += weight * sum|a - b|,
 * da = weight * sign(a - b); da may be NULL. */
int cgan_l1_nhwc(const void* a, const void* b, int32_t dtype, int64_t numel, float weight, float* loss_accum, void* da,
                 void* stream);
/* spectral norm backward (autograd of climategan/norms.py:107-112, u and v constants): in place
 * grad_w <- grad_w / sigma - (<grad_w, w_bar> / sigma^2) u v^T; workspace: CGAN_SN_BWD_WORKSPACE_FLOATS device floats
 * (per-block partial dot products, summed in a fixed order: no memset, no atomics, run-to-run identical). */
#define CGAN_SN_BWD_WORKSPACE_FLOATS 1024
int cgan_spectral_norm_bwd(float* grad_w, const float* w_bar, const float* u, const float* v, const float* sigma,
                           int32_t rows, int32_t cols, float* workspace, void* stream);

/* Painter training-step glue (climategan/trainer.py:1256-1387 G side, 1073-1107 D side).
 * heads_fwd: p = fake ? x (1 - m) + fake m : x  (the paste of generator.py:295-296; fake NHWC 3 channels stored as 8,
 *   x / m NCHW fp32), then d_in = [m | p] (torch.cat([m, x], axis=1), trainer.py:1101-1102; 4 channels stored as 8)
 *   and vgg_in = vgg_preprocess(p * m) (tutils.py:416-427: BGR, (t + 1) * 255 * 0.5 - mean), stored as a 16-bit PAIR
 *   per colour -- channels [b_hi g_hi r_hi b_lo g_lo r_lo 0 0], value = hi + lo -- because a single 16-bit store of
 *   magnitudes 100-150 loses +-0.5 (bf16); the first VGG conv runs on the six channels with its weights repeated;
 *   either output may be NULL.  heads_bwd: d_fake = m * (d_d_in[1..3] + 127.5 m d_vgg_in[BGR -> RGB]) (d_vgg_in: the hi channels' gradient). */
int cgan_painter_heads_fwd(const void* fake_nhwc, const float* x_nchw, const float* m_nchw, void* d_in, void* vgg_in,
                           int32_t dtype, int32_t n, int32_t h, int32_t w, void* stream);
int cgan_painter_heads_bwd(const void* d_d_in, const void* d_vgg_in, const float* m_nchw, void* d_fake, int32_t dtype,
                           int32_t n, int32_t h, int32_t w, void* stream);
/* backward of cgan_avgpool3x3s2_nhwc: dy [n][h_out][w_out][cs] -> dx [n][h_in][w_in][cs] */
int cgan_avgpool3x3s2_bwd_nhwc(const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                               int32_t w_in, void* stream);
/* nn.MaxPool2d(2, 2) of the VGG19 feature extractor (climategan/losses.py:304-334) and its backward (dy routed to the
 * first maximum of each window, torch's tie rule) */
int cgan_maxpool2x2_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                         void* stream);
int cgan_maxpool2x2_bwd_nhwc(const void* x, const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                             int32_t w_in, void* stream);
/* the same times [x > 0]: x is the output of a ReLU (VGG-19's conv + ReLU in front of every pool, losses.py:304-334) whose
 * derivative is taken here instead of by cgan_act_bwd over dx -- bit-identical to the two calls (round 6) */
int cgan_maxpool2x2_relu_bwd_nhwc(const void* x, const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                  int32_t w_in, void* stream);

/* Masker-side losses (climategan/losses.py:106-196, 444-524) on the decoders' NHWC maps; same accumulate-into-a-
 * device-scalar convention as above (weight carries 1/N and the lambdas).
 *  softmax / softmax_bwd        torch.softmax(s, dim=1) of the segmentation logits (trainer.py:1433) and its backward
 *  sigmoid_pair(_bwd)           prob = cat[sigmoid(x), 1 - sigmoid(x)] of the mask logits (trainer.py:1533-1534)
 *  softmax_ce                   nn.CrossEntropyLoss(logits, target int64) (losses.py:106-112): sum(lse(x) - x[t])
 *  tv                           TVLoss (losses.py:142-169): weight_h sum (x[y]-x[y-1])^2 + weight_w sum (x[x]-x[x-1])^2
 *  entropy_map(_bwd)            prob_2_entropy (losses.py:453-458), optionally times a 1-channel depth map (DADA)
 *  minent                       MinentLoss v1 / v2 (losses.py:172-196) on a probability map; workspace: one float
 *  bce_logits_map               nn.BCEWithLogitsLoss(x, target map) for the mask (trainer.py:1549-1553)
 *  ground_intersection          GroundIntersectionLoss (losses.py:444-450): sum 1[(g - p) > 0.5] (no gradient)
 *  affine_sum                   sum(a x + b): the WGAN form -mean(y D + (1 - y)(1 - D)) of losses.py:498-499 */
int cgan_softmax_nhwc(const void* x, void* y, int32_t dtype, int64_t npix, int32_t c, void* stream);
int cgan_softmax_bwd_nhwc(const void* y, const void* dy, void* dx, int32_t dtype, int64_t npix, int32_t c, void* stream);
int cgan_sigmoid_pair_nhwc(const void* x, void* y, int32_t dtype, int64_t npix, void* stream);
int cgan_sigmoid_pair_bwd_nhwc(const void* y, const void* dy, void* dx, int32_t dtype, int64_t npix, void* stream);
int cgan_softmax_ce_nhwc(const void* logits, const int64_t* target, int32_t dtype, int64_t npix, int32_t c, float weight,
                         float* loss_accum, void* dlogits, void* stream);
int cgan_tv_nhwc(const void* x, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, float weight_h, float weight_w,
                 float* loss_accum, void* dx, void* stream);
/* The ADVENT discriminators' input from the LOGITS (round 2): prob_2_entropy(softmax(s)) [* depth] (trainer.py:1433,
 * 1455-1456, losses.py:453-458, 517-519; c = 2..16 logit channels) or, with sigmoid_pair, prob_2_entropy(cat[sigmoid(x),
 * 1 - sigmoid(x)]) of the 1-channel mask logits (trainer.py:1533-1534), evaluated in fp32 and written as a 16-bit pair:
 * channels [0, C) = round16(v), [C, 2C) = round16(v - hi) of an NHWC map with 2C channels (C = c, or 2).  The
 * discriminator's first conv runs on it with input-duplicated weights.  _bwd: dlogits from the gradient of the pair map
 * (its first C channels are read: both halves carry the same data gradient). */
int cgan_advent_entropy_pair_nhwc(const void* logits, const void* depth, void* y, int32_t dtype, int64_t npix, int32_t c,
                                  int32_t sigmoid_pair, void* stream);
int cgan_advent_entropy_pair_bwd_nhwc(const void* logits, const void* depth, const void* dy, void* dlogits, int32_t dtype,
                                      int64_t npix, int32_t c, int32_t sigmoid_pair, void* stream);
/* The same pair map from fp32 NCHW PROBABILITIES [n, c, h, w] (+ optional fp32 depth [n, 1, h, w]): the reference's call
 * signature hands ADVENTAdversarialLoss softmax(pred) / cat[p, 1 - p] as NCHW tensors (losses.py:517-519,
 * trainer.py:1470-1476, 1588-1594); prob_2_entropy (losses.py:453-458) is evaluated in fp32 here.  _bwd: fp32 NCHW d(prob). */
int cgan_entropy_pair_from_nchw(const float* prob, const float* depth, void* y, int32_t dtype, int32_t n, int32_t c,
                                int32_t h, int32_t w, void* stream);
int cgan_entropy_pair_from_nchw_bwd(const float* prob, const float* depth, const void* dy, float* dprob, int32_t dtype,
                                    int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
int cgan_entropy_map_nhwc(const void* p, const void* depth, void* y, int32_t dtype, int64_t npix, int32_t c, void* stream);
int cgan_entropy_map_bwd_nhwc(const void* p, const void* depth, const void* dy, void* dp, int32_t dtype, int64_t npix,
                              int32_t c, void* stream);
/* workspace: CGAN_MINENT_WORKSPACE_FLOATS device floats (per-block partial sums of the mean entropy, added in a fixed order) */
#define CGAN_MINENT_WORKSPACE_FLOATS 4096
int cgan_minent_nhwc(const void* p, int32_t dtype, int64_t npix, int32_t c, int32_t version, float lambda_var,
                     float weight, float* loss_accum, void* dp, float* workspace, void* stream);
int cgan_bce_logits_map_nhwc(const void* x, const float* target, int32_t dtype, int64_t npix, float weight,
                             float* loss_accum, void* dx, void* stream);
int cgan_ground_intersection_nhwc(const void* p, const float* ground, int32_t dtype, int64_t npix, float weight,
                                  float* loss_accum, void* stream);
int cgan_affine_sum_nhwc(const void* x, int32_t dtype, int64_t npix, int32_t c, float a, float b, float* loss_accum,
                         void* dx, void* stream);

/* Backward / training-mode pieces of the Masker's graph:
 *  resize_bilinear_bwd   adjoint of cgan_resize_bilinear_nhwc (gather form: every input pixel sums the outputs that read it;
 *                        no workspace is used any more -- workspace_bytes() returns 0, the arguments are ignored)
 *  maxpool3x3s2_bwd      adjoint of cgan_maxpool3x3s2_nhwc (gradient to the first maximum of each window)
 *  add_act               y = act(a + b): the residual add + ReLU of a bottleneck when BatchNorm cannot be folded
 *                        (climategan/deeplab/resnet101_v3.py:46-48); its backward is cgan_act_bwd on y, to both inputs
 *  slice_channels        dst[.., 0..c) = src[.., c_off_src .. c_off_src + c): adjoint of the channel concatenation */
size_t cgan_resize_bilinear_bwd_workspace_bytes(int32_t n, int32_t c, int32_t h_in, int32_t w_in);
int cgan_resize_bilinear_bwd_nhwc(const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                  int32_t w_in, int32_t h_out, int32_t w_out, int32_t align_corners, void* workspace,
                                  size_t workspace_bytes, void* stream);
int cgan_maxpool3x3s2_bwd_nhwc(const void* x, const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                               int32_t w_in, void* stream);
int cgan_add_act_nhwc(const void* a, const void* b, void* y, int32_t dtype, int32_t act, float act_slope, int64_t numel,
                      void* stream);
int cgan_slice_channels_nhwc(const void* src, void* dst, int64_t npix, int32_t c, int32_t cs_src, int32_t c_off_src,
                             void* stream);

/* SIGMLoss (climategan/losses.py:237-278): scale-and-shift-invariant depth loss with a `scales`-level Sobel gradient
 * matching term, prediction = the depth decoder's 1-channel NHWC map [b][h][w][8], target fp32 [b][1][h][w].
 * *loss_accum += weight * loss; dpred (may be NULL) = weight * d loss / d prediction.  Medians are exact order statistics
 * (radix select).  workspace: cgan_sigm_loss_workspace_bytes(b, h, w). */
size_t cgan_sigm_loss_workspace_bytes(int32_t b, int32_t h, int32_t w);
int cgan_sigm_loss_nhwc(const void* pred, const float* target, int32_t dtype, int32_t b, int32_t h, int32_t w,
                        float gmweight, int32_t scales, float weight, float* loss_accum, void* dpred, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Output post-ops of the inference harness (Trainer.infer_all, climategan/trainer.py:311-332)
 * ------------------------------------------------------------------------------------------------ */
/* tutils.normalize (climategan/tutils.py:567-576) per image over (C,H,W), then `(t * 255).astype(uint8)` (truncation)
 * and the NCHW -> NHWC permute (trainer.py:314-326): x_nchw [n,c,h,w] fp32 (is_half 0) or fp16 (is_half 1, every
 * intermediate rounded to fp16 as the reference's half tensors are) -> out_nhwc uint8 [n,h,w,c].
 * workspace: cgan_normalize_u8_workspace_bytes(n) device bytes. */
size_t cgan_normalize_u8_workspace_bytes(int32_t n);
int cgan_normalize_u8_nhwc(const void* x_nchw, int32_t is_half, uint8_t* out_nhwc, int32_t n, int32_t c, int32_t h,
                           int32_t w, void* workspace, size_t workspace_bytes, void* stream);
/* `(m > bin_value).to(m.dtype)` (climategan/trainer.py:1870-1871) into y (same dtype as x; may be NULL) and
 * `((mask > bin_value) * 255).astype(uint8)` (trainer.py:329-332) into y_u8 (may be NULL) */
int cgan_binarize(const void* x, int32_t is_half, void* y, uint8_t* y_u8, float threshold, int64_t numel, void* stream);

/* Smog event (Trainer.compute_smog, climategan/trainer.py:1879-1939, parameters shared/trainer/events.yaml:9-14):
 * irradiance = srgb2lrgb(normalize(x)) (tutils.py:534-538), depth = normalize(1 / normalize(d, 0.3, 1), 0.1, 1)
 * bilinearly resized (align_corners=True) to (h, w), transmission = exp(-beta depth),
 * out = lrgb2srgb(t irradiance + (1 - t) airlight) (1 - alpha) + yellow alpha.
 * x_nchw fp32 [n][3][h][w]; depth_nhwc: the depth decoder's 16-bit map [n][dh][dw][8] (channel 0); yellow_rgb01:
 * HOST pointer to 3 floats in [0, 1]; out_nchw fp32 [n][3][h][w]; workspace cgan_smog_workspace_bytes(n). */
size_t cgan_smog_workspace_bytes(int32_t n);
int cgan_smog_nchw(const float* x_nchw, const void* depth_nhwc, int32_t dtype, float* out_nchw, int32_t n, int32_t h,
                   int32_t w, int32_t dh, int32_t dw, float airlight, float beta, float alpha, const float* yellow_rgb01,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Conditioning image of OmniGenerator.paint_cloudy (climategan/generator.py:299-328): sky = argmax_c(bilinear(s -> (h,w),
 * align_corners=False)) == sky_idx; noised = sky ? weight (perlin - min(perlin)) + (1 - weight) x : x
 * (tutils.mix_noise / rand_perlin_2d, tutils.py:647-694, one noise map per call); cond = noised (1 - m), NHWC with the 3
 * channels stored as 4 (what the Painter consumes).  angles: DEVICE fp32 [(res_y+1)][(res_x+1)] = 2 pi U(0,1) drawn
 * by the caller (the reference draws them with torch.rand); seg_nhwc: the segmentation decoder's 16-bit logits
 * [n][seg_h][seg_w][cgan_cs(seg_c)]; workspace cgan_cloudy_cond_workspace_bytes(h, w). */
size_t cgan_cloudy_cond_workspace_bytes(int32_t h, int32_t w);
int cgan_cloudy_cond_nhwc(const float* x_nchw, const float* m_nchw, const void* seg_nhwc, const float* angles,
                          void* cond_nhwc, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t seg_h, int32_t seg_w,
                          int32_t seg_c, int32_t sky_idx, int32_t res_y, int32_t res_x, float weight, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Eval-mode nn.BatchNorm2d as the per-(n, c) (mean, rstd) pair cgan_norm_act_apply / cgan_spade_fused_fwd consume:
 * rstd = gamma / sqrt(running_var + eps), mean = running_mean - beta / rstd; gamma / beta NULL for affine=False (the
 * "batch" param-free norm of SPADE, climategan/norms.py:152-153; Conv2dBlock's spectral_batch, blocks.py:118-136).
 * mean, rstd: fp32 [n][cgan_cs(c)]. */
int cgan_bn_eval_stats(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, float* mean, float* rstd, int32_t n, int32_t c, void* stream);
/* OmniGenerator.make_m_cond (climategan/generator.py:196-230): cond = cat[normalize(d), softmax(s, dim=1),
 * bilinear(x -> (h, w), align_corners=True)] as NHWC with round_up(1 + seg_c + 3, 4) channels; x_nchw NULL drops the
 * image channels (cond_nc 12).  depth_nhwc [n][h][w][8] (channel 0), seg_nhwc [n][h][w][cgan_cs(seg_c)]. */
size_t cgan_make_m_cond_workspace_bytes(int32_t n);
int cgan_make_m_cond_nhwc(const void* depth_nhwc, const void* seg_nhwc, const float* x_nchw, void* cond_nhwc,
                          int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t seg_c, int32_t x_h, int32_t x_w,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Autograd of make_m_cond with gen.m.spade.detach = false (generator.py:216-221): given the gradient of the
 * conditioning map, ddepth [n][h][w][8] (channel 0; the per-sample min / max of tutils.normalize, tutils.py:567-576, send
 * their gradient to the first arg-min / arg-max pixel as torch's min(1) / max(1) do) and dseg [n][h][w][cgan_cs(seg_c)]
 * (softmax backward).  The image channels carry no gradient (x is data). */
int cgan_make_m_cond_bwd_nhwc(const void* dcond_nhwc, const void* depth_nhwc, const void* seg_nhwc, void* ddepth_nhwc,
                              void* dseg_nhwc, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t seg_c,
                              int32_t with_x, void* stream);
/* Adjoint of cgan_resize_nearest_nhwc (autograd of F.interpolate(segmap, mode="nearest"), norms.py:179): dx[n][h_in][w_in]
 * [cs_in] = sum of the dy[n][h_out][w_out][cs_out] pixels that read it (fp32 accumulation; storage padding zeroed). */
int cgan_resize_nearest_bwd_nhwc(const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                 int32_t w_in, int32_t cs_in, int32_t h_out, int32_t w_out, int32_t cs_out, void* stream);

/* Wildfire event (climategan/fire.py:68-126 add_fire, parameters shared/trainer/events.yaml:1-8): normalize(x, 0, 255),
 * warm, uint8, adjust_contrast(1.5), adjust_brightness(0.73); sky = argmax(seg) == sky_idx (bottom third cleared when
 * crop_bottom), nearest-resized to (h, w), grown by 18 % (increase_sky_mask), blurred by the kernel_size x kernel_size
 * Gaussian (reflect border); paste of the (255, filter_green, 0) filter with `transparency`/255, adjust_brightness(0.8),
 * the two dummy corner pixels.  filter_green: the reference's random.randint(100, 150), drawn by the caller.
 * x_nchw fp32 [n][3][h][w] in [-1, 1]; seg_nhwc 16-bit logits [n][seg_h][seg_w][cgan_cs(seg_c)]; out_nchw fp32 in [0, 255].
 * The torchvision / kornia arithmetic is restated from those libraries' documentation (not in the reference tree). */
size_t cgan_wildfire_workspace_bytes(int32_t n, int32_t h, int32_t w, int32_t seg_h, int32_t seg_w, int32_t kernel_size);
int cgan_wildfire_nchw(const float* x_nchw, const void* seg_nhwc, int32_t dtype, float* out_nchw, int32_t n, int32_t h,
                       int32_t w, int32_t seg_h, int32_t seg_w, int32_t seg_c, int32_t sky_idx, int32_t kernel_size,
                       float kernel_sigma, float transparency, int32_t crop_bottom, float filter_green, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Input pipeline of apply_events (apply_events.py:179-195 to_m1_p1, :211-241 resize_and_crop): uint8 HWC
 * photo -> aspect-preserving resize (smaller side = `to`) -> centre crop `to` x `to` -> uint8 truncation -> [-1, 1] fp32
 * CHW.  The resize is scikit-image 0.18.3 resize(img, size, preserve_range=True, anti_aliasing=True) (a dependency that is
 * NOT in the reference tree): float64 Gaussian pre-filter (sigma = max(0, (scale - 1) / 2) per axis, truncate 4, mirror
 * border) + bilinear warp with pixel-centre alignment, restated from that library's published algorithm.
 *  - cgan_resize_crop_geometry: resized extent (rows, cols) and crop origin (top, left) for an h x w input (host only);
 *  - weights_rows / weights_cols: the normalised 1-D Gaussian taps [2 radius + 1] (fp64, device) the caller computes
 *    as scipy does (exp(-0.5 x^2 / sigma^2) / sum, radius = int(4 sigma + 0.5)); radius 0 = no filter on that axis;
 *  - out_chw: fp32 [c][to][to]; workspace cgan_resize_crop_u8_workspace_bytes(h, w, c). */
int cgan_resize_crop_geometry(int32_t h, int32_t w, int32_t to, int32_t* rows, int32_t* cols, int32_t* top, int32_t* left);
size_t cgan_resize_crop_u8_workspace_bytes(int32_t h, int32_t w, int32_t c);
int cgan_resize_crop_u8(const void* img_hwc_u8, int32_t h, int32_t w, int32_t c, int32_t to, const double* weights_rows,
                        int32_t radius_rows, const double* weights_cols, int32_t radius_cols, float* out_chw,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The keep_ratio branch of apply_events (apply_events.py:494-497, 502): to_m1_p1(resize(img, (rows, cols),
 * anti_aliasing=True)) -- the same resize to an arbitrary extent (rows, cols from utils.to_128), no crop, no uint8
 * truncation (the float image / 255).  out_chw fp32 [c][rows][cols]; same workspace. */
int cgan_resize_u8(const void* img_hwc_u8, int32_t h, int32_t w, int32_t c, int32_t rows, int32_t cols,
                   const double* weights_rows, int32_t radius_rows, const double* weights_cols, int32_t radius_cols,
                   float* out_chw, void* workspace, size_t workspace_bytes, void* stream);

/* Validation metrics (climategan/eval_metrics.py:67-130 accuracy, mIOU; used by Trainer.eval_images, trainer.py:1706-1790):
 * per-class pixel counts of argmax_c(pred) against a label map.  pred: layout 0 = NHWC 16-bit [n][hw][cgan_cs(c)],
 * layout 1 = NCHW fp32 [n][c][hw]; labels fp32 [n][hw] (class ids as floats; anything that is not an integer in [0, c)
 * counts for no class, like the reference's ignore index).  counts u64 [3][c] ACCUMULATED (zero first):
 * [0][k] pixels predicted k, [1][k] pixels labelled k, [2][k] both.  accuracy = sum_k [2][k] / (n hw);
 * IoU_k = [2][k] / ([0][k] + [1][k] - [2][k]).  First maximum wins on ties (np.argmax / torch.argmax). */
int cgan_seg_counts(const void* pred, int32_t layout, int32_t dtype, int32_t n, int64_t hw, int32_t c,
                    const float* labels, unsigned long long* counts, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange on RCCL.  The reference trains in ONE process on one device
 * (trainer.py:674-683: backward, then the optimizer on the local gradients); data parallelism over per-GPU slices needs the
 * average of the ranks' gradients before ExtraAdam's extrapolation / step.  The host mirror's default path runs that
 * all-reduce through torch.distributed (backend "nccl" = RCCL); these entry points are the same collective for a host that
 * owns its communicator (and the reducer's opt-in direct path, CGAN_DDP_DIRECT_RCCL=1).
 *  - librccl is loaded at run time, never linked: cgan_rccl_load(path) (the copy torch ships: one RCCL per process); all
 *    other entry points return CGAN_ERR_BAD_ARG until it has succeeded.
 *  - cgan_comm_unique_id: rank 0 fills 128 bytes (ncclGetUniqueId) and distributes them out of band;
 *    cgan_comm_init_rank: every rank, collectively (ncclCommInitRank); cgan_comm_destroy.
 *  - cgan_allreduce_bucket: IN-PLACE SUM of one flat bucket (count elements of CGAN_F32, CGAN_BF16 or CGAN_F16) over the
 *    communicator's ranks, enqueued on `stream`; the caller divides by the world size.
 * ------------------------------------------------------------------------------------------------ */
int cgan_rccl_load(const char* librccl_path);
int cgan_rccl_loaded(void);
int cgan_comm_unique_id(void* id128);
int cgan_comm_init_rank(void** comm, int32_t nranks, const void* id128, int32_t rank);
int cgan_comm_destroy(void* comm);
int cgan_allreduce_bucket(void* buf, int64_t count, int32_t dtype, void* comm, void* stream);

/* ---- split-precision ("pair16") inference path of the Masker (round 4) -------------------------------------------------
 * The reference's default apply_events run is fp32 (apply_events.py:465-468: --half is opt-in) and binarises an fp32 mask
 * (trainer.py:1866-1871); these entry points carry every activation of the eval-mode Masker (deeplab/resnet101_v3.py:176-187,
 * deeplab_v3.py:244-266, depth.py:128-155, blocks.py:292-313) as SEVERAL 16-bit numbers whose sum is the value, so that the
 * 16-bit MFMA kernels reproduce the fp32 arithmetic: dtype CGAN_F16 = pairs hi + lo (22 bits of mantissa where lo stays a
 * normal number, an absolute floor of 2^-24 below |v| ~ 0.1), dtype CGAN_BF16 = triples hi + mid + lo (24 bits at any
 * magnitude: what the module mirror's G.float() / G.set_compute_dtype("split24") selects; "pair16" = the fp16 pairs).
 * A split map of C channels is an NHWC buffer of NS * cgan_cs(C) channels per pixel that stores every component ONCE: NS = 2
 * blocks (hi | lo) resp. NS = 3 blocks (hi | mid | lo) (round 6; rounds 4-5 stored the NB K-blocks below, i.e. twice the
 * bytes).  A conv multiplies NB = 3 K-blocks (hi | lo | hi) resp. NB = 6 (hi | mid | lo | hi | mid | hi) -- its kernels read
 * K-block b from the storage block holding that component -- against weights expanded to (W_hi | W_hi | W_lo) resp. (W_hi |
 * W_hi | W_hi | W_mid | W_mid | W_lo) along the input channels (cgan_pair_expand_weight, then the ordinary
 * cgan_conv2d_pack_weight with c_in = NB * cgan_cs(C_in)): every cross product above the type's precision floor accumulates
 * in fp32 inside the existing conv kernel.
 * cgan_conv2d_nhwc_fwd_pair: the conv of such a map (d->c_in = NB * cgan_cs(C_in) = the K extent, x3 holds NS * cgan_cs(C_in)
 * channels per pixel; d->c_out = C_out; bias, optional split residual, activation in fp32) stored as a split map again
 * (NS * cgan_cs(C_out) channels).  The rest are the Masker's glue
 * ops between convs, each evaluated in fp32 on the sum of the components: layout edges (fp32 NCHW <-> split; _to_nchw with
 * the sigmoid of generator.py:277; _to_nhwc = one ordinary 16-bit map for the event kernels), nn.MaxPool2d(3, 2, 1),
 * F.interpolate bilinear (both align_corners) / legacy nearest, the DADA product (deeplab_v3.py:253-254), torch.cat on
 * channels. */
int cgan_conv2d_nhwc_fwd_pair(const void* x3, const void* packed_w3, const float* bias_padded, const void* residual3,
                              void* y3, const CganConvDesc* d, void* stream);
int cgan_pair_expand_weight(const float* w_oihw, const float* sigma, float* w3, int32_t dtype, int32_t c_out, int32_t c_in,
                            int32_t kh, int32_t kw, void* stream);
/* Split-precision Painter (round 5; the reference's fp32 apply_events run, apply_events.py:465-468, through SPADE:
 * climategan/norms.py:151,174-186 and painter.py:149-168).  The fused SPADE kernel multiplies a 16-bit hidden map; on split maps
 * the three convolutions run as cgan_conv2d_nhwc_fwd_pair and these two kernels do the rest in fp32 on the sums of the components:
 *  pair_instnorm_stats  mean / rstd [n][round_up(c,8)] fp32 of F.instance_norm (biased variance), two fp64 passes
 *  pair_spade_apply     y = act((x - mean) rstd (1 + gamma) + beta), gamma / beta = the mlp_gamma / mlp_beta conv outputs (bias
 *                       included), x optionally read through the folded x2 nearest upsample; act none, ReLU or LeakyReLU.
 *                       gamma3 == beta3 == NULL: y = act((x - mean) rstd) -- the eval-mode BatchNorm + LeakyReLU behind the SPADE
 *                       mask decoder's spectral-norm projection convs (climategan/masker.py:96-140, blocks.py:96-150)
 *  pair_make_m_cond     the SPADE mask decoder's conditioning map (OmniGenerator.make_m_cond, climategan/generator.py:196-230;
 *                       tutils.normalize :567-576) from the split depth / segmentation maps in the reference's fp32 arithmetic:
 *                       cat[(d - min) / max(d - min) per image, softmax(s, dim 1), bilinear(x, align_corners = True)] as a split
 *                       map of 1 + seg_c (+ 3 with x_nchw) channels; workspace = 2 floats per image */
int cgan_pair_instnorm_stats(const void* x3, float* mean, float* rstd, int32_t dtype, int32_t n, int32_t c, int64_t hw, float eps,
                             void* stream);
int cgan_pair_spade_apply(const void* x3, const float* mean, const float* rstd, const void* gamma3, const void* beta3, void* y3,
                          int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t x_upsample, int32_t act,
                          float act_slope, void* stream);
size_t cgan_pair_make_m_cond_workspace_bytes(int32_t n);
int cgan_pair_make_m_cond(const void* depth3, const void* seg3, const float* x_nchw, void* cond3, int32_t dtype, int32_t n,
                          int32_t h, int32_t w, int32_t seg_c, int32_t x_h, int32_t x_w, void* workspace, size_t workspace_bytes,
                          void* stream);
int cgan_pair_from_nchw(const float* x, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
int cgan_pair_to_nchw(const void* x3, float* y, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, int32_t sigmoid,
                      void* stream);
int cgan_pair_to_nhwc(const void* x3, void* y, int32_t dtype, int64_t npix, int32_t c, void* stream);
int cgan_pair_maxpool3x3s2(const void* x3, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
int cgan_pair_resize_bilinear(const void* x3, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                              int32_t h_out, int32_t w_out, int32_t align_corners, void* stream);
int cgan_pair_resize_nearest(const void* x3, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                             int32_t h_out, int32_t w_out, void* stream);
/* F.interpolate(mode="bicubic", align_corners=False) on a split map in fp32: the depth decoder's resize to the MiDaS size
 * (climategan/depth.py:143-149) when the depth map feeds the SPADE mask decoder's conditioning in the split-precision mode */
int cgan_pair_resize_bicubic(const void* x3, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                             int32_t h_out, int32_t w_out, void* stream);
int cgan_pair_mul(const void* a3, const void* b3, void* y3, int32_t dtype, int64_t npix, int32_t c, void* stream);
int cgan_pair_copy_channels(const void* src3, void* dst3, int32_t dtype, int64_t npix, int32_t c, int32_t c_dst, int32_t c_off,
                            void* stream);

/* libcgan_hip.so exports exactly the entry points declared above: no development knob, no process-global mutable state
 * behind the ABI besides the thread-local error string and the lazily loaded RCCL handle.  The kernel-selection /
 * ablation / timestamp knobs (cgan_debug_set_*) that tools/ and the every-kernel-variant tests use exist only in the
 * DEVELOPMENT build of the same sources (-DCGAN_DEV -> libcgan_hip_dev.so, `make -C climategan_amd/csrc dev`); in the
 * product build they are compile-time constants (csrc/cgan_common.h). */

#ifdef __cplusplus
}
#endif
#endif /* CLIMATEGAN_HIP_H */
