// Split-precision ("pair16") activation maps for the inference-time Masker (round 4).
//
// The reference's default apply_events run and its binarised flood mask are fp32 (apply_events.py:465-468,
// trainer.py:1866-1871); every conv kernel of this library multiplies 16-bit operands on the MFMA units.  A value v is
// therefore carried as several 16-bit numbers (cgan_common.h, Split<T>: fp16 pairs hi + lo, bf16 triples hi + mid + lo) and
// a map is stored per pixel as NS = NC channel blocks of round_up(C, 8) channels each -- (hi | lo) resp. (hi | mid | lo); a conv
// reads them as the NB K-blocks (hi | lo | hi) resp. (hi | mid | lo | hi | mid | hi) -- multiplied by packed weights (W_hi | W_hi | W_lo) resp. (W_hi | W_hi | W_hi | W_mid |
// W_mid | W_lo) along K: an existing conv kernel then accumulates every cross product above the type's precision floor in
// fp32.  cgan_conv2d_nhwc_fwd_pair (conv_mfma.hip) stores its fp32 result as such a map again; the few glue ops of the Masker
// between convs (max-pool, bilinear / nearest resize, channel concatenation, the DADA product, sigmoid, the layout edges) are
// here, each in fp32 on the sum of the components.
#include "cgan_common.h"

namespace {

inline int grid_for(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 65535 * 4 ? 65535 * 4 : g));
}

// 8 channels (group cg) of pixel `px` of a split map with per-block stride cs: v = sum of the components
template <typename T>
__device__ __forceinline__ void pair_load8(const uint16_t* __restrict__ px, int cs, int cg, float* v) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
  for (int k = Split<T>::NC - 1; k >= 0; --k) {       // smallest component first
    const u32x4 q = *reinterpret_cast<const u32x4*>(px + k * cs + cg * 8);     // blocks 0 .. NC-1 hold components 0 .. NC-1
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a, b;
      unpack2<T>(q[e], a, b);
      v[2 * e] += a;
      v[2 * e + 1] += b;
    }
  }
}
// split 8 fp32 values into the components and store every block
template <typename T>
__device__ __forceinline__ void pair_store8(uint16_t* __restrict__ px, int cs, int cg, const float* v) {
  u32x4 comp[Split<T>::NC];
  float rem[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) rem[e] = v[e];
#pragma unroll
  for (int k = 0; k < Split<T>::NC; ++k) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      comp[k][e] = pack2<T>(rem[2 * e], rem[2 * e + 1]);
      float q0, q1;
      unpack2<T>(comp[k][e], q0, q1);
      rem[2 * e] -= q0;
      rem[2 * e + 1] -= q1;
    }
  }
#pragma unroll
  for (int b = 0; b < Split<T>::NS; ++b) *reinterpret_cast<u32x4*>(px + b * cs + cg * 8) = comp[b];
}

// fp32 NCHW -> pair NHWC (pad channels zero)
template <typename T>
__global__ void pair_from_nchw_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, int c, int hw, int cs, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const long n = pix / hw, p = pix % hw;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = cg * 8 + e;
      v[e] = ch < c ? x[(n * c + ch) * (long)hw + p] : 0.f;
    }
    pair_store8<T>(y + pix * Split<T>::NS * cs, cs, cg, v);
  }
}

// pair NHWC -> fp32 NCHW (op 1: sigmoid first, generator.py:277)
template <typename T>
__global__ void pair_to_nchw_kernel(const uint16_t* __restrict__ x, float* __restrict__ y, int c, int hw, int cs, int op,
                                    long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const long n = pix / hw, p = pix % hw;
    float v[8];
    pair_load8<T>(x + pix * Split<T>::NS * cs, cs, cg, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = cg * 8 + e;
      if (ch < c) y[(n * c + ch) * (long)hw + p] = op == 1 ? 1.f / (1.f + expf(-v[e])) : v[e];
    }
  }
}

// pair NHWC -> ordinary 16-bit NHWC (hi + lo rounded once): what the event kernels (wildfire, smog, flood painter) read
template <typename T>
__global__ void pair_to_nhwc_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int cs, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    float v[8];
    pair_load8<T>(x + pix * Split<T>::NS * cs, cs, cg, v);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
    *reinterpret_cast<u32x4*>(y + pix * cs + cg * 8) = o;
  }
}

// nn.MaxPool2d(3, stride 2, padding 1) (resnet101_v3.py:69)
template <typename T>
__global__ void pair_maxpool3x3s2_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h, int w, int ho, int wo,
                                         int cs, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int ox = (int)(pix % wo);
    const long r = pix / wo;
    const int oy = (int)(r % ho);
    const long n = r / ho;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy - 1 + dy;
      if (iy < 0 || iy >= h) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = 2 * ox - 1 + dx;
        if (ix < 0 || ix >= w) continue;
        float v[8];
        pair_load8<T>(x + ((n * h + iy) * (long)w + ix) * Split<T>::NS * cs, cs, cg, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    pair_store8<T>(y + pix * Split<T>::NS * cs, cs, cg, m);
  }
}

// F.interpolate(mode="bilinear", align_corners=...) in fp32, torch's index rule (as resize_bilinear_kernel, edge.hip)
template <typename T>
__global__ void pair_resize_bilinear_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h_in, int w_in,
                                            int h_out, int w_out, int cs, float sy, float sx, int align, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int ox = (int)(pix % w_out);
    const long r = pix / w_out;
    const int oy = (int)(r % h_out);
    const long n = r / h_out;
    // the products rounded before anything is subtracted from them, as torch's CPU kernels form the source index (a fused
    // multiply-add here moves the interpolation weight by up to half an ulp of the INDEX: 6e-6 of the neighbours' difference
    // at index 60 -- visible at the split maps' precision)
    float py = (float)oy * sy, px = (float)ox * sx;
    asm volatile("" : "+v"(py), "+v"(px));
    const float fy = align ? py : fmaxf(fmaf((float)oy + 0.5f, sy, -0.5f), 0.f);
    const float fx = align ? px : fmaxf(fmaf((float)ox + 0.5f, sx, -0.5f), 0.f);
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < h_in - 1 ? y0 : h_in - 1;
    x0 = x0 < w_in - 1 ? x0 : w_in - 1;
    const int y1 = y0 < h_in - 1 ? y0 + 1 : y0, x1 = x0 < w_in - 1 ? x0 + 1 : x0;
    const float ly = fy - y0, lx = fx - x0;
    const uint16_t* base = x + n * (long)h_in * w_in * Split<T>::NS * cs;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    pair_load8<T>(base + ((long)y0 * w_in + x0) * Split<T>::NS * cs, cs, cg, v00);
    pair_load8<T>(base + ((long)y0 * w_in + x1) * Split<T>::NS * cs, cs, cg, v01);
    pair_load8<T>(base + ((long)y1 * w_in + x0) * Split<T>::NS * cs, cs, cg, v10);
    pair_load8<T>(base + ((long)y1 * w_in + x1) * Split<T>::NS * cs, cs, cg, v11);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = (1.f - ly) * ((1.f - lx) * v00[e] + lx * v01[e]) + ly * ((1.f - lx) * v10[e] + lx * v11[e]);
    pair_store8<T>(y + pix * Split<T>::NS * cs, cs, cg, o);
  }
}

// F.interpolate(mode="bicubic", align_corners=False) in fp32 (climategan/depth.py:144-149; as resize_bicubic_kernel, edge.hip:
// torch's cubic convolution with A = -0.75, source index (dst + 0.5) in/out - 0.5, taps clamped to the border)
__device__ __forceinline__ void pair_cubic_coeffs(float t, float* w) {
  const float A = -0.75f;
  const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
template <typename T>
__global__ void pair_resize_bicubic_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h_in, int w_in,
                                           int h_out, int w_out, int cs, float sy, float sx, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int ox = (int)(pix % w_out);
    const long r = pix / w_out;
    const int oy = (int)(r % h_out);
    const long n = r / h_out;
    const float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int iy = (int)fly, ix = (int)flx;
    float wy[4], wx[4];
    pair_cubic_coeffs(fy - fly, wy);
    pair_cubic_coeffs(fx - flx, wx);
    const uint16_t* base = x + n * (long)h_in * w_in * Split<T>::NS * cs;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int i = 0; i < 4; ++i) {
      int yy = iy - 1 + i;
      yy = yy < 0 ? 0 : (yy > h_in - 1 ? h_in - 1 : yy);
      float row[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) row[e] = 0.f;
      for (int j = 0; j < 4; ++j) {
        int xx = ix - 1 + j;
        xx = xx < 0 ? 0 : (xx > w_in - 1 ? w_in - 1 : xx);
        float v[8];
        pair_load8<T>(base + ((long)yy * w_in + xx) * Split<T>::NS * cs, cs, cg, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) row[e] += wx[j] * v[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += wy[i] * row[e];
    }
    pair_store8<T>(y + pix * Split<T>::NS * cs, cs, cg, acc);
  }
}

// F.interpolate(mode="nearest") (legacy index rule floor(dst * in / out), blocks.py:28-43): a copy of both halves
__global__ void pair_resize_nearest_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h_in, int w_in,
                                           int h_out, int w_out, int cs3, float sy, float sx, long total) {
  const int g_total = cs3 / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int gi = (int)(idx % g_total);
    const long pix = idx / g_total;
    const int ox = (int)(pix % w_out);
    const long r = pix / w_out;
    const int oy = (int)(r % h_out);
    const long n = r / h_out;
    int iy = (int)floorf(oy * sy), ix = (int)floorf(ox * sx);
    iy = iy < h_in - 1 ? iy : h_in - 1;
    ix = ix < w_in - 1 ? ix : w_in - 1;
    *reinterpret_cast<u32x4*>(y + pix * cs3 + gi * 8) =
        *reinterpret_cast<const u32x4*>(x + ((n * h_in + iy) * (long)w_in + ix) * cs3 + gi * 8);
  }
}

// y = a * b (DADA feature fusion, deeplab_v3.py:253-254) in fp32
template <typename T>
__global__ void pair_mul_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ y,
                                int cs, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    float va[8], vb[8];
    pair_load8<T>(a + pix * Split<T>::NS * cs, cs, cg, va);
    pair_load8<T>(b + pix * Split<T>::NS * cs, cs, cg, vb);
#pragma unroll
    for (int e = 0; e < 8; ++e) va[e] *= vb[e];
    pair_store8<T>(y + pix * Split<T>::NS * cs, cs, cg, va);
  }
}

// the c channels of every block of src into channels [c_off, c_off + c) of the same block of dst (torch.cat on channels:
// one call per input; c_off a multiple of 8)
__global__ void pair_copy_channels_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int cs_src, int cs_dst,
                                          int c_off, int nb, long total) {
  const int groups = cs_src / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int gi = (int)(idx % groups);
    const long r = idx / groups;
    const int blk = (int)(r % nb);
    const long pix = r / nb;
    *reinterpret_cast<u32x4*>(dst + pix * nb * cs_dst + blk * cs_dst + c_off + gi * 8) =
        *reinterpret_cast<const u32x4*>(src + pix * nb * cs_src + blk * cs_src + gi * 8);
  }
}

// fp32 OIHW weight [cout][cin][taps] (optionally / sigma) -> fp32 [cout][NB * cs_in][taps]: block b holds component
// wcomp(b) of the weight (w0 = round16(W), w1 = round16(W - w0), ...); zeros on the pad channels.  The ordinary weight pack
// then rounds nothing.
template <typename T>
__global__ void pair_expand_weight_kernel(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ w3,
                                          int cin, int cs_in, int taps, long total) {
  const float sg = sigma ? sigma[0] : 1.f;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(idx % taps);
    const long r = idx / taps;
    const int k = (int)(r % (Split<T>::NB * cs_in));
    const long co = r / (Split<T>::NB * cs_in);
    const int blk = k / cs_in, ci = k - blk * cs_in;
    float v = 0.f;
    if (ci < cin) {
      float rem = __fdiv_rn(w[(co * cin + ci) * taps + tap], sg);
      const int want = Split<T>::wcomp(blk);
#pragma unroll
      for (int q = 0; q < Split<T>::NC; ++q) {
        const float c = f32_of_bits<T>(bits_of<T>(rem));
        if (q == want) v = c;
        rem -= c;
      }
    }
    w3[idx] = v;
  }
}


// Instance-norm statistics of a split map (round 5: the split-precision Painter): one workgroup per (image, 8-channel group),
// two passes over the pixels in fp64 (mean, then the biased variance around it) -- the arithmetic of F.instance_norm in the
// reference's fp32 run (norms.py:151,174) with room to spare; mean / rstd rows fp32 [n][cs]
template <typename T>
__global__ __launch_bounds__(256) void pair_instnorm_stats_kernel(const uint16_t* __restrict__ x, float* __restrict__ mean,
                                                                  float* __restrict__ rstd, int hw, int cs, float eps) {
  __shared__ double red[256][8];
  const int cg = blockIdx.x, n = blockIdx.y;
  const uint16_t* base = x + (size_t)n * hw * Split<T>::NS * cs;
  double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = threadIdx.x; p < hw; p += 256) {
    float v[8];
    pair_load8<T>(base + (size_t)p * Split<T>::NS * cs, cs, cg, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += (double)v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[threadIdx.x + st][e];
    __syncthreads();
  }
  double m[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = red[0][e] / (double)hw;
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0;
  for (int p = threadIdx.x; p < hw; p += 256) {
    float v[8];
    pair_load8<T>(base + (size_t)p * Split<T>::NS * cs, cs, cg, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double d = (double)v[e] - m[e];
      s[e] += d * d;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[threadIdx.x][e] += red[threadIdx.x + st][e];
    __syncthreads();
  }
  if (threadIdx.x < 8) {
    const int e = threadIdx.x;
    mean[(size_t)n * cs + cg * 8 + e] = (float)m[e];
    rstd[(size_t)n * cs + cg * 8 + e] = (float)(1.0 / sqrt(red[0][e] / (double)hw + (double)eps));
  }
}

// SPADE's de-normalisation on split maps: y = act((x - mean) rstd (1 + gamma) + beta) in fp32 on the sums of the components
// (norms.py:181-186 + the block's LeakyReLU); x optionally read through the folded x2 nearest upsample.  gamma == beta ==
// nullptr: y = act((x - mean) rstd), the eval-mode BatchNorm of the SPADE mask decoder's projection convs (masker.py:96-140)
template <typename T>
__global__ void pair_spade_apply_kernel(const uint16_t* __restrict__ x, const float* __restrict__ mean,
                                        const float* __restrict__ rstd, const uint16_t* __restrict__ gamma,
                                        const uint16_t* __restrict__ beta, uint16_t* __restrict__ y, int h, int w, int c, int cs,
                                        int ups, int act, float slope, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int xx = (int)(pix % w);
    const long r = pix / w;
    const int yy = (int)(r % h);
    const long n = r / h;
    const long xpix = ups ? (n * (h >> 1) + (yy >> 1)) * (long)(w >> 1) + (xx >> 1) : pix;
    float xv[8], gv[8], bv[8], o[8];
    pair_load8<T>(x + xpix * Split<T>::NS * cs, cs, cg, xv);
    if (gamma) {
      pair_load8<T>(gamma + pix * Split<T>::NS * cs, cs, cg, gv);
      pair_load8<T>(beta + pix * Split<T>::NS * cs, cs, cg, bv);
    } else {                     // a plain normalisation (+ activation): eval-mode BatchNorm behind a split-precision conv
#pragma unroll
      for (int e = 0; e < 8; ++e) gv[e] = bv[e] = 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = cg * 8 + e;
      float v = 0.f;
      if (ch < c) {
        const float xh = (xv[e] - mean[n * cs + ch]) * rstd[n * cs + ch];
        v = act_apply(xh * (1.f + gv[e]) + bv[e], act, slope);
      }
      o[e] = v;
    }
    pair_store8<T>(y + pix * Split<T>::NS * cs, cs, cg, o);
  }
}

// Conditioning map of the SPADE mask decoder on split maps (OmniGenerator.make_m_cond, generator.py:196-230, in the
// reference's fp32 arithmetic): per-image min / max of the depth map ...
template <typename T>
__global__ __launch_bounds__(256) void pair_minmax_c0_kernel(const uint16_t* __restrict__ d, float* __restrict__ mm, int hw) {
  __shared__ float smin[256], smax[256];
  const int n = blockIdx.x;
  const uint16_t* base = d + (size_t)n * hw * Split<T>::NS * 8;
  float lo = __builtin_inff(), hi = -__builtin_inff();
  for (int p = threadIdx.x; p < hw; p += 256) {
    float v[8];
    pair_load8<T>(base + (size_t)p * Split<T>::NS * 8, 8, 0, v);
    lo = fminf(lo, v[0]);
    hi = fmaxf(hi, v[0]);
  }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + st]);
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    mm[2 * n] = smin[0];
    mm[2 * n + 1] = smax[0];
  }
}

// ... then cat[normalize(d), softmax(s, dim = 1), bilinear(x -> (h, w), align_corners = True)] per pixel, every term in fp32
constexpr int PAIR_COND_MAX_C = 32;
template <typename T>
__global__ __launch_bounds__(256) void pair_make_m_cond_kernel(const uint16_t* __restrict__ d, const uint16_t* __restrict__ seg,
                                                               const float* __restrict__ x, const float* __restrict__ mm,
                                                               uint16_t* __restrict__ cond, int h, int w, int sc, int scs, int xh,
                                                               int xw, int with_x, int ccs, float sy, float sx, long total) {
  const long hw = (long)h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, p = i - n * hw;
    float o[PAIR_COND_MAX_C];
#pragma unroll
    for (int c = 0; c < PAIR_COND_MAX_C; ++c) o[c] = 0.f;
    float v[8];
    pair_load8<T>(d + i * Split<T>::NS * 8, 8, 0, v);
    const float dmin = mm[2 * n], dmax = mm[2 * n + 1];
    o[0] = __fdiv_rn(v[0] - dmin, dmax - dmin);            // tutils.normalize: (t - min) / max(t - min)
    float sv[PAIR_COND_MAX_C];
    float mx = -__builtin_inff();
#pragma unroll
    for (int g = 0; g < PAIR_COND_MAX_C / 8; ++g)
      if (g * 8 < sc) {
        pair_load8<T>(seg + i * Split<T>::NS * scs, scs, g, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sv[g * 8 + e] = v[e];
          if (g * 8 + e < sc) mx = fmaxf(mx, v[e]);
        }
      }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < PAIR_COND_MAX_C; ++c)
      if (c < sc) {
        sv[c] = expf(sv[c] - mx);
        sum += sv[c];
      }
#pragma unroll
    for (int c = 0; c < PAIR_COND_MAX_C - 1; ++c)
      if (c < sc) o[1 + c] = __fdiv_rn(sv[c], sum);
    if (with_x) {
      const int oy = (int)(p / w), ox = (int)(p - (long)oy * w);
      float fy = (float)oy * sy, fx = (float)ox * sx;
      asm volatile("" : "+v"(fy), "+v"(fx));      // the rounded products, as torch forms them: no fma with the subtraction below
      int y0 = (int)fy, x0 = (int)fx;
      y0 = y0 < xh - 1 ? y0 : xh - 1;
      x0 = x0 < xw - 1 ? x0 : xw - 1;
      const int y1 = y0 < xh - 1 ? y0 + 1 : y0, x1 = x0 < xw - 1 ? x0 + 1 : x0;
      const float ly = fy - y0, lx = fx - x0;
      for (int c = 0; c < 3; ++c) {
        const float* xb = x + (n * 3 + c) * (long)xh * xw;
        const float val = (1.f - ly) * ((1.f - lx) * xb[(long)y0 * xw + x0] + lx * xb[(long)y0 * xw + x1]) +
                          ly * ((1.f - lx) * xb[(long)y1 * xw + x0] + lx * xb[(long)y1 * xw + x1]);
#pragma unroll
        for (int k = 0; k < PAIR_COND_MAX_C; ++k)           // (a register array takes compile-time indices)
          if (k == 1 + sc + c) o[k] = val;
      }
    }
#pragma unroll
    for (int g = 0; g < PAIR_COND_MAX_C / 8; ++g)
      if (g * 8 < ccs) pair_store8<T>(cond + i * Split<T>::NS * ccs, ccs, g, o + g * 8);
  }
}

}  // namespace

#define PAIR_DISPATCH(dtype, KERNEL, ...)                                  \
  do {                                                                     \
    if ((dtype) == CGAN_F16) hipLaunchKernelGGL(KERNEL<F16>, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<BF16>, __VA_ARGS__);                    \
  } while (0)
#define PAIR_CHECK_DT(what) CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, what ": bad dtype %d", dtype)

extern "C" int cgan_pair_from_nchw(const float* x, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w,
                                   void* stream) {
  CGAN_REQUIRE(x && y3 && n > 0 && c > 0 && h > 0 && w > 0, "pair_from_nchw: bad arguments");
  PAIR_CHECK_DT("pair_from_nchw");
  const int cs = cgan_cs(c);
  const long total = (long)n * h * w * (cs / 8);
  PAIR_DISPATCH(dtype, pair_from_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y3, c,
                h * w, cs, total);
  CGAN_CHECK_LAUNCH("pair_from_nchw");
  return CGAN_OK;
}

extern "C" int cgan_pair_to_nchw(const void* x3, float* y, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w,
                                 int32_t sigmoid, void* stream) {
  CGAN_REQUIRE(x3 && y && n > 0 && c > 0 && h > 0 && w > 0, "pair_to_nchw: bad arguments");
  PAIR_CHECK_DT("pair_to_nchw");
  const int cs = cgan_cs(c);
  const long total = (long)n * h * w * (cs / 8);
  PAIR_DISPATCH(dtype, pair_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x3, y, c,
                h * w, cs, sigmoid ? 1 : 0, total);
  CGAN_CHECK_LAUNCH("pair_to_nchw");
  return CGAN_OK;
}

extern "C" int cgan_pair_to_nhwc(const void* x3, void* y, int32_t dtype, int64_t npix, int32_t c, void* stream) {
  CGAN_REQUIRE(x3 && y && npix > 0 && c > 0, "pair_to_nhwc: bad arguments");
  PAIR_CHECK_DT("pair_to_nhwc");
  const int cs = cgan_cs(c);
  const long total = (long)npix * (cs / 8);
  PAIR_DISPATCH(dtype, pair_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x3,
                (uint16_t*)y, cs, total);
  CGAN_CHECK_LAUNCH("pair_to_nhwc");
  return CGAN_OK;
}

extern "C" int cgan_pair_maxpool3x3s2(const void* x3, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w,
                                      void* stream) {
  CGAN_REQUIRE(x3 && y3 && n > 0 && c > 0 && h > 0 && w > 0, "pair_maxpool3x3s2: bad arguments");
  PAIR_CHECK_DT("pair_maxpool3x3s2");
  const int cs = cgan_cs(c), ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const long total = (long)n * ho * wo * (cs / 8);
  PAIR_DISPATCH(dtype, pair_maxpool3x3s2_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x3,
                (uint16_t*)y3, h, w, ho, wo, cs, total);
  CGAN_CHECK_LAUNCH("pair_maxpool3x3s2");
  return CGAN_OK;
}

extern "C" int cgan_pair_resize_bilinear(const void* x3, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                         int32_t w_in, int32_t h_out, int32_t w_out, int32_t align_corners, void* stream) {
  CGAN_REQUIRE(x3 && y3 && n > 0 && c > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "pair_resize_bilinear: bad arguments");
  PAIR_CHECK_DT("pair_resize_bilinear");
  const int cs = cgan_cs(c);
  const long total = (long)n * h_out * w_out * (cs / 8);
  float sy, sx;
  if (align_corners) {
    sy = h_out > 1 ? (float)(h_in - 1) / (float)(h_out - 1) : 0.f;
    sx = w_out > 1 ? (float)(w_in - 1) / (float)(w_out - 1) : 0.f;
  } else {
    sy = (float)h_in / (float)h_out;
    sx = (float)w_in / (float)w_out;
  }
  PAIR_DISPATCH(dtype, pair_resize_bilinear_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                (const uint16_t*)x3, (uint16_t*)y3, h_in, w_in, h_out, w_out, cs, sy, sx, align_corners ? 1 : 0, total);
  CGAN_CHECK_LAUNCH("pair_resize_bilinear");
  return CGAN_OK;
}

extern "C" int cgan_pair_resize_nearest(const void* x3, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                        int32_t w_in, int32_t h_out, int32_t w_out, void* stream) {
  CGAN_REQUIRE(x3 && y3 && n > 0 && c > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "pair_resize_nearest: bad arguments");
  PAIR_CHECK_DT("pair_resize_nearest");
  const int cs3 = cgan_split_store_blocks(dtype) * cgan_cs(c);
  const long total = (long)n * h_out * w_out * (cs3 / 8);
  hipLaunchKernelGGL(pair_resize_nearest_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x3,
                     (uint16_t*)y3, h_in, w_in, h_out, w_out, cs3, (float)h_in / (float)h_out, (float)w_in / (float)w_out, total);
  CGAN_CHECK_LAUNCH("pair_resize_nearest");
  return CGAN_OK;
}

extern "C" int cgan_pair_mul(const void* a3, const void* b3, void* y3, int32_t dtype, int64_t npix, int32_t c, void* stream) {
  CGAN_REQUIRE(a3 && b3 && y3 && npix > 0 && c > 0, "pair_mul: bad arguments");
  PAIR_CHECK_DT("pair_mul");
  const int cs = cgan_cs(c);
  const long total = (long)npix * (cs / 8);
  PAIR_DISPATCH(dtype, pair_mul_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a3,
                (const uint16_t*)b3, (uint16_t*)y3, cs, total);
  CGAN_CHECK_LAUNCH("pair_mul");
  return CGAN_OK;
}

extern "C" int cgan_pair_copy_channels(const void* src3, void* dst3, int32_t dtype, int64_t npix, int32_t c, int32_t c_dst,
                                       int32_t c_off, void* stream) {
  CGAN_REQUIRE(src3 && dst3 && npix > 0 && c > 0 && c_dst > 0, "pair_copy_channels: bad arguments");
  PAIR_CHECK_DT("pair_copy_channels");
  CGAN_REQUIRE((c_off % 8) == 0 && c_off + cgan_cs(c) <= cgan_cs(c_dst), "pair_copy_channels: bad channel offset %d", c_off);
  const int nb = cgan_split_store_blocks(dtype);
  const long total = (long)npix * nb * (cgan_cs(c) / 8);
  hipLaunchKernelGGL(pair_copy_channels_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src3,
                     (uint16_t*)dst3, cgan_cs(c), cgan_cs(c_dst), c_off, nb, total);
  CGAN_CHECK_LAUNCH("pair_copy_channels");
  return CGAN_OK;
}

extern "C" int cgan_pair_expand_weight(const float* w_oihw, const float* sigma, float* w3, int32_t dtype, int32_t c_out,
                                       int32_t c_in, int32_t kh, int32_t kw, void* stream) {
  CGAN_REQUIRE(w_oihw && w3 && c_out > 0 && c_in > 0 && kh > 0 && kw > 0, "pair_expand_weight: bad arguments");
  PAIR_CHECK_DT("pair_expand_weight");
  const int cs_in = cgan_cs(c_in), taps = kh * kw;
  const long total = (long)c_out * cgan_split_blocks(dtype) * cs_in * taps;
  PAIR_DISPATCH(dtype, pair_expand_weight_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w_oihw, sigma, w3,
                c_in, cs_in, taps, total);
  CGAN_CHECK_LAUNCH("pair_expand_weight");
  return CGAN_OK;
}

extern "C" int cgan_pair_instnorm_stats(const void* x3, float* mean, float* rstd, int32_t dtype, int32_t n, int32_t c, int64_t hw,
                                        float eps, void* stream) {
  CGAN_REQUIRE(x3 && mean && rstd && n > 0 && c > 0 && hw > 0 && hw < (1L << 31), "pair_instnorm_stats: bad arguments");
  PAIR_CHECK_DT("pair_instnorm_stats");
  const int cs = cgan_cs(c);
  PAIR_DISPATCH(dtype, pair_instnorm_stats_kernel, dim3(cs / 8, n), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x3, mean,
                rstd, (int)hw, cs, eps);
  CGAN_CHECK_LAUNCH("pair_instnorm_stats");
  return CGAN_OK;
}

extern "C" int cgan_pair_spade_apply(const void* x3, const float* mean, const float* rstd, const void* gamma3, const void* beta3,
                                     void* y3, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, int32_t x_upsample,
                                     int32_t act, float act_slope, void* stream) {
  CGAN_REQUIRE(x3 && mean && rstd && y3 && n > 0 && h > 0 && w > 0 && c > 0, "pair_spade_apply: bad arguments");
  CGAN_REQUIRE((gamma3 == nullptr) == (beta3 == nullptr), "pair_spade_apply: gamma and beta go together");
  PAIR_CHECK_DT("pair_spade_apply");
  CGAN_REQUIRE(!x_upsample || ((h % 2) == 0 && (w % 2) == 0), "pair_spade_apply: x_upsample needs even h / w");
  CGAN_REQUIRE(act == CGAN_ACT_NONE || act == CGAN_ACT_LRELU || act == CGAN_ACT_RELU,
               "pair_spade_apply: activation none, ReLU or LeakyReLU");
  const int cs = cgan_cs(c);
  const long total = (long)n * h * w * (cs / 8);
  PAIR_DISPATCH(dtype, pair_spade_apply_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x3, mean,
                rstd, (const uint16_t*)gamma3, (const uint16_t*)beta3, (uint16_t*)y3, h, w, c, cs, x_upsample ? 1 : 0, act, act_slope,
                total);
  CGAN_CHECK_LAUNCH("pair_spade_apply");
  return CGAN_OK;
}

extern "C" size_t cgan_pair_make_m_cond_workspace_bytes(int32_t n) { return n > 0 ? (size_t)n * 2 * sizeof(float) : 0; }

extern "C" int cgan_pair_make_m_cond(const void* depth3, const void* seg3, const float* x_nchw, void* cond3, int32_t dtype,
                                     int32_t n, int32_t h, int32_t w, int32_t seg_c, int32_t x_h, int32_t x_w, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  CGAN_REQUIRE(depth3 && seg3 && cond3 && workspace, "pair_make_m_cond: null pointer");
  PAIR_CHECK_DT("pair_make_m_cond");
  CGAN_REQUIRE(n > 0 && h > 0 && w > 0 && seg_c > 0, "pair_make_m_cond: bad shape");
  CGAN_REQUIRE((long)h * w < (1l << 31), "pair_make_m_cond: map too large");
  CGAN_REQUIRE(!x_nchw || (x_h > 0 && x_w > 0), "pair_make_m_cond: bad x shape");
  const int cond_c = 1 + seg_c + (x_nchw ? 3 : 0);
  CGAN_REQUIRE(cgan_cs(cond_c) <= PAIR_COND_MAX_C, "pair_make_m_cond: at most %d conditioning channels", PAIR_COND_MAX_C);
  CGAN_REQUIRE(workspace_bytes >= cgan_pair_make_m_cond_workspace_bytes(n), "pair_make_m_cond: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const float sy = (x_nchw && h > 1) ? (float)(x_h - 1) / (float)(h - 1) : 0.f;
  const float sx = (x_nchw && w > 1) ? (float)(x_w - 1) / (float)(w - 1) : 0.f;
  const long total = (long)n * h * w;
  PAIR_DISPATCH(dtype, pair_minmax_c0_kernel, dim3(n), dim3(256), 0, s, (const uint16_t*)depth3, (float*)workspace, h * w);
  PAIR_DISPATCH(dtype, pair_make_m_cond_kernel, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)depth3,
                (const uint16_t*)seg3, x_nchw, (const float*)workspace, (uint16_t*)cond3, h, w, seg_c, cgan_cs(seg_c), x_h, x_w,
                x_nchw ? 1 : 0, cgan_cs(cond_c), sy, sx, total);
  CGAN_CHECK_LAUNCH("pair_make_m_cond");
  return CGAN_OK;
}

extern "C" int cgan_pair_resize_bicubic(const void* x3, void* y3, int32_t dtype, int32_t n, int32_t c, int32_t h_in, int32_t w_in,
                                        int32_t h_out, int32_t w_out, void* stream) {
  CGAN_REQUIRE(x3 && y3 && n > 0 && c > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "pair_resize_bicubic: bad arguments");
  PAIR_CHECK_DT("pair_resize_bicubic");
  const int cs = cgan_cs(c);
  const long total = (long)n * h_out * w_out * (cs / 8);
  PAIR_DISPATCH(dtype, pair_resize_bicubic_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x3,
                (uint16_t*)y3, h_in, w_in, h_out, w_out, cs, (float)h_in / (float)h_out, (float)w_in / (float)w_out, total);
  CGAN_CHECK_LAUNCH("pair_resize_bicubic");
  return CGAN_OK;
}
