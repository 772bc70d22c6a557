// Interface of the LDS-tiled 3x3 convolution kernel (conv3x3_lds.hip), used by conv_mfma.hip's dispatcher.
#pragma once
#include "cgan_common.h"

struct Conv3x3LdsArgs {
  const uint16_t* x;
  const u32x4* w;       // packed [ctile][ks = tap * (cin_p/32) + q][lane] (the 3x3 padded K layout)
  const float* bias;    // padded to ctiles*16, or null
  const uint16_t* res;  // residual or null
  uint16_t* y;
  int n, h, w_;         // output extent
  int hi, wi;           // logical input extent (h + 2 - 2 pad)
  int pad, reflect;     // 0 / 1 / 2 zero padding ('valid' / 'same' / 'full'), or 1 with reflect = 1
  int hx, wx;           // stored input extent (hi/2, wi/2 when in_ups)
  int cin;              // logical input channels (<= 4: the folded-tap kernel)
  int cin_s, cin_p, cout, cout_s, ctiles, ksteps;
  int in_ups, act, has_res, res_ups;
  float slope;
  int shuffle;          // > 0: depth-to-space epilogue of the sub-pixel data gradient (conv_mfma.hip): the 16 output "channels"
                        // are (class a, class b, 4 channels); lane group g = 2 a + b stores its four values (+ four zero pad
                        // channels) to pixel (2 y + a, 2 x + b) of an [n][shuffle_h][shuffle_w][8] map; = shuffle_h
  int shuffle_w;
  int k, stride;        // conv_smallcin_kernel only: square kernel size (<= 7) and stride (1 / 2); 0 elsewhere
};

// true if this conv is a 3x3 / stride 1 / dilation 1 conv (zero pad 0..2, or reflect pad 1) large enough for the tiled kernel
bool conv3x3_lds_applicable(const CganConvDesc* d);
int conv3x3_lds_launch(const Conv3x3LdsArgs& a, int dtype, hipStream_t s);
// first-layer convolutions: 8 storage channels in, <= 64 out, k x k (k != 3) with stride 1 / 2, zero padding: input halo tile in
// LDS, all output channels per workgroup (the PatchGAN 4x4 s2 input conv, the ResNet 7x7 s2 stem)
bool conv_smallcin_applicable(const CganConvDesc* d);
int conv_smallcin_launch(const Conv3x3LdsArgs& a, int dtype, hipStream_t s);
