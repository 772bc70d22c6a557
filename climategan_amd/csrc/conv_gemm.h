// Interface of the LDS-tiled implicit-GEMM convolution kernel for wide layers (conv_gemm.hip), used by
// conv_mfma.hip's dispatcher.
#pragma once
#include "cgan_common.h"

struct ConvGemmArgs {
  const uint16_t* x;
  const u32x4* w;       // packed [ctile][ks = tap * (cin_s/32) + cc][lane]
  const float* bias;    // padded to ctiles*16, or null
  const uint16_t* res;  // residual or null
  uint16_t* y;
  int n, h_in, w_in, cin_s;
  int cout, cout_s, ctiles, ksteps;
  int kh, kw, stride, pad, dil, pad_mode;
  int h_out, w_out, npix;
  int act, has_res, res_ups;
  float slope;
  // optional: per-(pixel chunk, channel) (mean, M2) of the fp32 results before bias-free rounding, [chunk][cout_s][2],
  // chunk = conv_gemm_stats_chunk_pixels() consecutive output pixels (training-mode BatchNorm statistics without a
  // second pass over y); null = off
  float* stats;
};

// true for convs whose channel counts make the 128x256 (cout x pixel) LDS tiling worthwhile
bool conv_gemm_applicable(const CganConvDesc* d);
int conv_gemm_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);
// pixels per statistics chunk of the kernel conv_gemm_launch would run for ``a`` (0: that kernel writes no statistics,
// or npix is not a whole number of chunks)
int conv_gemm_stats_chunk_pixels(const ConvGemmArgs& a, int dtype);
// conv_gemm_big.hip: the 256 couts x 256 pixels / K = 64 kernel (eight waves, one workgroup per CU)
bool conv_gemm_big_ok(const ConvGemmArgs& a);
// conv1x1_direct.hip: 1x1 layers with <= 256 couts, activations straight into B-fragment registers, all couts per workgroup
bool conv1x1_allc_ok(const ConvGemmArgs& a);
int conv1x1_allc_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);
int conv_gemm_big_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);
// conv1x1_xres.hip: short-K (cin <= 256) 1x1 layers, activation tile resident in LDS, all couts per workgroup
bool conv1x1_xres_ok(const ConvGemmArgs& a);
int conv1x1_xres_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);
