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
};

// true for convs whose channel counts make the 128x256 (cout x pixel) LDS tiling worthwhile
bool conv_gemm_applicable(const CganConvDesc* d);
int conv_gemm_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);
