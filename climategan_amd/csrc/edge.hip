// Edge / glue kernels: NCHW fp32 <-> NHWC 16-bit layout passes (with the paint() masking and paste folded
// in), nearest resize and the PatchGAN's 3x3/s2 average pool.  All HBM-bound, one pass.
#include "cgan_common.h"

namespace {

// ---- NCHW fp32 -> NHWC 16-bit.  One thread per pixel: each channel-plane read is coalesced along w
// (64 consecutive floats per wave) and the thread writes its pixel's cs*2 contiguous bytes.  Only used at the
// API edge on 3-4 channel tensors.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                    uint16_t* __restrict__ y, int c, int hw, int cs, long total_pix) {
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total_pix; pix += (long)gridDim.x * blockDim.x) {
    long n = pix / hw;
    long p = pix - n * hw;
    float keep = mask ? 1.f - mask[n * hw + p] : 1.f;
    const float* src = x + n * (long)c * hw + p;
    uint16_t* dst = y + pix * cs;
    for (int c0 = 0; c0 < cs; c0 += 4) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (c0 + e < c) ? src[(long)(c0 + e) * hw] * keep : 0.f;
      u32x2 o;
      o[0] = pack2<T>(v[0], v[1]);
      o[1] = pack2<T>(v[2], v[3]);
      *reinterpret_cast<u32x2*>(dst + c0) = o;
    }
  }
}

// ---- NHWC 16-bit -> NCHW fp32 (+ optional paste: out = px * (1 - m) + y * m)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const uint16_t* __restrict__ y, const float* __restrict__ px,
                                    const float* __restrict__ pm, float* __restrict__ out, int c, int hw, int cs,
                                    long total_pix) {
  for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total_pix; pix += (long)gridDim.x * blockDim.x) {
    long n = pix / hw;
    long p = pix - n * hw;
    const uint16_t* src = y + pix * cs;
    float m = pm ? pm[n * hw + p] : 1.f;
    for (int c0 = 0; c0 < c; c0 += 4) {
      u32x2 v = *reinterpret_cast<const u32x2*>(src + c0);
      float f[4];
      unpack2<T>(v[0], f[0], f[1]);
      unpack2<T>(v[1], f[2], f[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (c0 + e < c) {
          long o = (n * c + c0 + e) * (long)hw + p;
          out[o] = px ? px[o] * (1.f - m) + f[e] * m : f[e];
        }
      }
    }
  }
}

template <typename T>
__global__ void resize_nearest_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int c, int h_in,
                                      int w_in, int cs_in, int h_out, int w_out, int cs_out, float sy, float sx,
                                      long total) {
  const int g_out = cs_out / 4;
  // (32-bit index arithmetic: a map below 2 GiB has fewer than 2^28 four-channel groups)
  for (long idl = (long)blockIdx.x * blockDim.x + threadIdx.x; idl < total; idl += (long)gridDim.x * blockDim.x) {
    const unsigned idx = (unsigned)idl;
    const unsigned pix = idx / (unsigned)g_out;
    int g = (int)(idx - pix * (unsigned)g_out);
    const unsigned r = pix / (unsigned)w_out;
    int ox = (int)(pix - r * (unsigned)w_out);
    const long n = (long)(r / (unsigned)h_out);
    int oy = (int)(r - (unsigned)n * (unsigned)h_out);
    int iy = nearest_src(oy, sy, h_in), ix = nearest_src(ox, sx, w_in);
    const uint16_t* src = x + ((n * h_in + iy) * (long)w_in + ix) * cs_in;
    uint16_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int ch = g * 4 + e;
      o[e] = (ch < c) ? src[ch] : (uint16_t)0;
    }
    u32x2 pk;
    pk[0] = o[0] | ((uint32_t)o[1] << 16);
    pk[1] = o[2] | ((uint32_t)o[3] << 16);
    *reinterpret_cast<u32x2*>(y + (size_t)pix * cs_out + g * 4) = pk;
  }
}

// AvgPool2d(3, stride=2, padding=1, count_include_pad=False)
template <typename T>
__global__ void avgpool3x3s2_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h_in, int w_in,
                                    int h_out, int w_out, int cs, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int cg = (int)(idx % cg_total);
    long pix = idx / cg_total;
    int ox = (int)(pix % w_out);
    long r = pix / w_out;
    int oy = (int)(r % h_out);
    long n = r / h_out;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    int cnt = 0;
    for (int dy = 0; dy < 3; ++dy) {
      int iy = oy * 2 - 1 + dy;
      if (iy < 0 || iy >= h_in) continue;
      for (int dx = 0; dx < 3; ++dx) {
        int ix = ox * 2 - 1 + dx;
        if (ix < 0 || ix >= w_in) continue;
        u32x4 v = *reinterpret_cast<const u32x4*>(x + ((n * h_in + iy) * (long)w_in + ix) * cs + cg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a, b;
          unpack2<T>(v[e], a, b);
          acc[2 * e] += a;
          acc[2 * e + 1] += b;
        }
        ++cnt;
      }
    }
    float inv = 1.f / (float)cnt;
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(acc[2 * e] * inv, acc[2 * e + 1] * inv);
    *reinterpret_cast<u32x4*>(y + pix * cs + cg * 8) = o;
  }
}

// MaxPool2d(3, stride=2, padding=1) (ResNet stem, reference deeplab/resnet101_v3.py:74)
template <typename T>
__global__ void maxpool3x3s2_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h_in, int w_in,
                                    int h_out, int w_out, int cs, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int cg = (int)(idx % cg_total);
    long pix = idx / cg_total;
    int ox = (int)(pix % w_out);
    long r = pix / w_out;
    int oy = (int)(r % h_out);
    long n = r / h_out;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = -INFINITY;
    for (int dy = 0; dy < 3; ++dy) {
      int iy = oy * 2 - 1 + dy;
      if (iy < 0 || iy >= h_in) continue;
      for (int dx = 0; dx < 3; ++dx) {
        int ix = ox * 2 - 1 + dx;
        if (ix < 0 || ix >= w_in) continue;
        u32x4 v = *reinterpret_cast<const u32x4*>(x + ((n * h_in + iy) * (long)w_in + ix) * cs + cg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a, b;
          unpack2<T>(v[e], a, b);
          acc[2 * e] = fmaxf(acc[2 * e], a);
          acc[2 * e + 1] = fmaxf(acc[2 * e + 1], b);
        }
      }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(acc[2 * e], acc[2 * e + 1]);
    *reinterpret_cast<u32x4*>(y + pix * cs + cg * 8) = o;
  }
}

// F.interpolate(mode="bilinear", align_corners=ac) on NHWC (reference deeplab_v3.py:136-138,262-264,
// blocks.py:300-302).  Source index as torch's area_pixel_compute_source_index: align_corners -> dst*(in-1)/(out-1);
// otherwise max((dst+0.5)*in/out - 0.5, 0).
template <typename T>
__global__ void resize_bilinear_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h_in, int w_in,
                                       int h_out, int w_out, int cs, float sy, float sx, int align, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int cg = (int)(idx % cg_total);
    long pix = idx / cg_total;
    int ox = (int)(pix % w_out);
    long r = pix / w_out;
    int oy = (int)(r % h_out);
    long n = r / h_out;
    float fy = align ? oy * sy : fmaxf((oy + 0.5f) * sy - 0.5f, 0.f);
    float fx = align ? ox * sx : fmaxf((ox + 0.5f) * sx - 0.5f, 0.f);
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < h_in - 1 ? y0 : h_in - 1;
    x0 = x0 < w_in - 1 ? x0 : w_in - 1;
    int y1 = y0 < h_in - 1 ? y0 + 1 : y0, x1 = x0 < w_in - 1 ? x0 + 1 : x0;
    float ly = fy - y0, lx = fx - x0;
    const uint16_t* base = x + n * (long)h_in * w_in * cs + cg * 8;
    u32x4 v00 = *reinterpret_cast<const u32x4*>(base + ((long)y0 * w_in + x0) * cs);
    u32x4 v01 = *reinterpret_cast<const u32x4*>(base + ((long)y0 * w_in + x1) * cs);
    u32x4 v10 = *reinterpret_cast<const u32x4*>(base + ((long)y1 * w_in + x0) * cs);
    u32x4 v11 = *reinterpret_cast<const u32x4*>(base + ((long)y1 * w_in + x1) * cs);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a00, b00, a01, b01, a10, b10, a11, b11;
      unpack2<T>(v00[e], a00, b00); unpack2<T>(v01[e], a01, b01);
      unpack2<T>(v10[e], a10, b10); unpack2<T>(v11[e], a11, b11);
      float a = (1.f - ly) * ((1.f - lx) * a00 + lx * a01) + ly * ((1.f - lx) * a10 + lx * a11);
      float b = (1.f - ly) * ((1.f - lx) * b00 + lx * b01) + ly * ((1.f - lx) * b10 + lx * b11);
      o[e] = pack2<T>(a, b);
    }
    *reinterpret_cast<u32x4*>(y + pix * cs + cg * 8) = o;
  }
}

// F.interpolate(mode="bicubic", align_corners=False) (climategan/depth.py:144-149): torch's cubic convolution with
// A = -0.75, source index (dst + 0.5) * in/out - 0.5 (not clamped), taps clamped to the border.
__device__ __forceinline__ void cubic_coeffs(float t, float* w) {
  const float A = -0.75f;
  float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
  w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
  w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
  w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
  w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

template <typename T>
__global__ void resize_bicubic_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h_in, int w_in,
                                      int h_out, int w_out, int cs, float sy, float sx, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int cg = (int)(idx % cg_total);
    long pix = idx / cg_total;
    int ox = (int)(pix % w_out);
    long r = pix / w_out;
    int oy = (int)(r % h_out);
    long n = r / h_out;
    float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
    float fly = floorf(fy), flx = floorf(fx);
    int iy = (int)fly, ix = (int)flx;
    float wy[4], wx[4];
    cubic_coeffs(fy - fly, wy);
    cubic_coeffs(fx - flx, wx);
    const uint16_t* base = x + n * (long)h_in * w_in * cs + cg * 8;
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
        u32x4 v = *reinterpret_cast<const u32x4*>(base + ((long)yy * w_in + xx) * cs);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a, b;
          unpack2<T>(v[e], a, b);
          row[2 * e] += wx[j] * a;
          row[2 * e + 1] += wx[j] * b;
        }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += wy[i] * row[e];
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(acc[2 * e], acc[2 * e + 1]);
    *reinterpret_cast<u32x4*>(y + pix * cs + cg * 8) = o;
  }
}

// adjoint of resize_bicubic_kernel in gather form: every INPUT pixel sums w_y w_x dy over the output pixels whose 4 x 4
// (border-clamped) taps touch it, re-evaluating the forward's index rule (the DADA depth decoder's 384^2 resize under
// autograd, depth.py:143-149); candidates: output rows whose source position lies within 3 input rows of this one
template <typename T>
__global__ void resize_bicubic_bwd_kernel(const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, int h_in, int w_in,
                                          int h_out, int w_out, int cs, float sy, float sx, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int ix = (int)(pix % w_in);
    const long r = pix / w_in;
    const int iy = (int)(r % h_in);
    const long n = r / h_in;
    int oy0 = (int)floorf((iy - 3 + 0.5f) / sy - 0.5f), oy1 = (int)ceilf((iy + 3 + 0.5f) / sy - 0.5f);
    int ox0 = (int)floorf((ix - 3 + 0.5f) / sx - 0.5f), ox1 = (int)ceilf((ix + 3 + 0.5f) / sx - 0.5f);
    // border pixels also receive the clamped taps of every output row / column beyond them
    if (iy == 0) oy0 = 0;
    if (iy == h_in - 1) oy1 = h_out - 1;
    if (ix == 0) ox0 = 0;
    if (ix == w_in - 1) ox1 = w_out - 1;
    oy0 = oy0 < 0 ? 0 : oy0; ox0 = ox0 < 0 ? 0 : ox0;
    oy1 = oy1 > h_out - 1 ? h_out - 1 : oy1; ox1 = ox1 > w_out - 1 ? w_out - 1 : ox1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy) {
      const float fy = (oy + 0.5f) * sy - 0.5f, fly = floorf(fy);
      float wy[4];
      cubic_coeffs(fy - fly, wy);
      float cy = 0.f;
      for (int i = 0; i < 4; ++i) {
        int yy = (int)fly - 1 + i;
        yy = yy < 0 ? 0 : (yy > h_in - 1 ? h_in - 1 : yy);
        if (yy == iy) cy += wy[i];
      }
      if (cy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        const float fx = (ox + 0.5f) * sx - 0.5f, flx = floorf(fx);
        float wx[4];
        cubic_coeffs(fx - flx, wx);
        float cx = 0.f;
        for (int j = 0; j < 4; ++j) {
          int xx = (int)flx - 1 + j;
          xx = xx < 0 ? 0 : (xx > w_in - 1 ? w_in - 1 : xx);
          if (xx == ix) cx += wx[j];
        }
        if (cx == 0.f) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(dy + ((n * h_out + oy) * (long)w_out + ox) * cs + cg * 8);
        const float wgt = cy * cx;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a, b;
          unpack2<T>(v[e], a, b);
          acc[2 * e] += wgt * a;
          acc[2 * e + 1] += wgt * b;
        }
      }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(acc[2 * e], acc[2 * e + 1]);
    *reinterpret_cast<u32x4*>(dx + pix * cs + cg * 8) = o;
  }
}

// backward of the nearest x2 upsample (InterpolateNearest2d, blocks.py:28-43): y[n][oy][ox] = sum of the 2x2 block of x
template <typename T>
__global__ void sumpool2x2_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h_out, int w_out, int cs,
                                  long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int cg = (int)(idx % cg_total);
    long pix = idx / cg_total;
    int ox = (int)(pix % w_out);
    long r = pix / w_out;
    int oy = (int)(r % h_out);
    long n = r / h_out;
    const uint16_t* base = x + ((n * 2 * h_out + 2 * oy) * (long)(2 * w_out) + 2 * ox) * cs + cg * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        u32x4 v = *reinterpret_cast<const u32x4*>(base + ((long)dy * 2 * w_out + dx) * cs);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a, b;
          unpack2<T>(v[e], a, b);
          acc[2 * e] += a;
          acc[2 * e + 1] += b;
        }
      }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(acc[2 * e], acc[2 * e + 1]);
    *reinterpret_cast<u32x4*>(y + pix * cs + cg * 8) = o;
  }
}

// backward of nn.ReflectionPad2d(p): dx[y][x] = sum of the padded-gradient entries that mirror onto (y, x)
template <typename T>
__global__ void reflect_pad_bwd_kernel(const uint16_t* __restrict__ dxp, uint16_t* __restrict__ dx, int h, int w, int pad,
                                       int cs, long total) {
  const int cg_total = cs / 8;
  const int hp = h + 2 * pad, wp = w + 2 * pad;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int x = (int)(pix % w);
    const long r = pix / w;
    const int y = (int)(r % h);
    const long n = r / h;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + pad;
    if (y >= 1 && y <= pad) ys[ny++] = pad - y;
    if (y <= h - 2 && y >= h - 1 - pad) ys[ny++] = 2 * (h - 1) - y + pad;
    xs[nx++] = x + pad;
    if (x >= 1 && x <= pad) xs[nx++] = pad - x;
    if (x <= w - 2 && x >= w - 1 - pad) xs[nx++] = 2 * (w - 1) - x + pad;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int a = 0; a < ny; ++a)
      for (int b = 0; b < nx; ++b) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(dxp + ((n * hp + ys[a]) * (long)wp + xs[b]) * cs + cg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float p0, p1;
          unpack2<T>(v[e], p0, p1);
          acc[2 * e] += p0;
          acc[2 * e + 1] += p1;
        }
      }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(acc[2 * e], acc[2 * e + 1]);
    *reinterpret_cast<u32x4*>(dx + pix * cs + cg * 8) = o;
  }
}

// copy the c channels of src [n*hw][cs_src] into channels [c_off, c_off + c) of dst [n*hw][cs_dst]
// (torch.cat along channels = one call per input; c_off must be a multiple of 8 -- true for every concat of the path)
__global__ void copy_channels_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int cs_src,
                                     int cs_dst, int c_off, int groups, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int gidx = (int)(idx % groups);
    long pix = idx / groups;
    *reinterpret_cast<u32x4*>(dst + pix * cs_dst + c_off + gidx * 8) =
        *reinterpret_cast<const u32x4*>(src + pix * cs_src + gidx * 8);
  }
}

// y = op(a, b): 0 = a * b (DADA feature fusion z * z_depth, reference deeplab_v3.py:253-254, blocks.py:304-305),
// 1 = sigmoid(a) (generator.py:277); 2 = a * s with s a device fp32 scalar (chain rule through a scalar loss weight)
template <typename T>
__global__ void eltwise_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ y,
                               int op, long total_groups) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total_groups; idx += (long)gridDim.x * blockDim.x) {
    u32x4 va = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(a) + idx);
    u32x4 vb = op == 0 ? CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(b) + idx) : va;
    const float sc = op == 2 ? reinterpret_cast<const float*>(b)[0] : 1.f;   // op 2: b is a device fp32 scalar
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a0, a1, b0, b1;
      unpack2<T>(va[e], a0, a1);
      unpack2<T>(vb[e], b0, b1);
      float r0 = op == 0 ? a0 * b0 : (op == 2 ? a0 * sc : 1.f / (1.f + __expf(-a0)));
      float r1 = op == 0 ? a1 * b1 : (op == 2 ? a1 * sc : 1.f / (1.f + __expf(-a1)));
      o[e] = pack2<T>(r0, r1);
    }
    CGAN_ST_STREAM(o, reinterpret_cast<u32x4*>(y) + idx);
  }
}

// Fold an eval-mode BatchNorm2d into the preceding convolution:  w' = w * s[o],  b' = (b - mean[o]) * s[o] + beta[o],
// s[o] = gamma[o] / sqrt(var[o] + eps)     (reference climategan/bn_fusion.py:121-132 states the same algebra)
__global__ void fold_bn_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                               const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps,
                               float* __restrict__ w_out, float* __restrict__ b_out, int cout, long per_out) {
  long total = (long)cout * per_out;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int o = (int)(idx / per_out);
    float s = (gamma ? gamma[o] : 1.f) * rsqrtf(var[o] + eps);
    w_out[idx] = w[idx] * s;
    if (idx % per_out == 0) b_out[o] = ((bias ? bias[o] : 0.f) - mean[o]) * s + (beta ? beta[o] : 0.f);
  }
}

inline int grid_for(long total, int threads = 256, int cap = 8192) {
  long b = (total + threads - 1) / threads;
  return (int)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace

extern "C" int cgan_nchw_to_nhwc(const float* x, const float* mask, void* y, int32_t dtype, int32_t n, int32_t c,
                                 int32_t h, int32_t w, int32_t cs, void* stream) {
  CGAN_REQUIRE(x && y, "nchw_to_nhwc: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "nchw_to_nhwc: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && cs >= c && (cs % 4) == 0, "nchw_to_nhwc: bad shape");
  long total = (long)n * h * w;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, x, mask, (uint16_t*)y, c,
                       h * w, cs, total);
  else
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, x, mask, (uint16_t*)y, c,
                       h * w, cs, total);
  CGAN_CHECK_LAUNCH("nchw_to_nhwc");
  return CGAN_OK;
}

extern "C" int cgan_nhwc_to_nchw(const void* y, const float* paste_x, const float* paste_m, float* out, int32_t dtype,
                                 int32_t n, int32_t c, int32_t h, int32_t w, int32_t cs, void* stream) {
  CGAN_REQUIRE(y && out, "nhwc_to_nchw: null pointer");
  CGAN_REQUIRE((paste_x == nullptr) == (paste_m == nullptr), "nhwc_to_nchw: paste_x and paste_m go together");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "nhwc_to_nchw: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && cs >= c && (cs % 4) == 0, "nhwc_to_nchw: bad shape");
  long total = (long)n * h * w;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)y, paste_x,
                       paste_m, out, c, h * w, cs, total);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)y, paste_x,
                       paste_m, out, c, h * w, cs, total);
  CGAN_CHECK_LAUNCH("nhwc_to_nchw");
  return CGAN_OK;
}

extern "C" int cgan_resize_nearest_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                        int32_t w_in, int32_t cs_in, int32_t h_out, int32_t w_out, int32_t cs_out,
                                        void* stream) {
  CGAN_REQUIRE(x && y, "resize_nearest: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "resize_nearest: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "resize_nearest: bad shape");
  CGAN_REQUIRE(cs_in >= c && cs_out >= c && (cs_out % 4) == 0, "resize_nearest: bad channel storage");
  long total = (long)n * h_out * w_out * (cs_out / 4);
  float sy = (float)h_in / (float)h_out, sx = (float)w_in / (float)w_out;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(resize_nearest_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, c, h_in, w_in, cs_in, h_out, w_out, cs_out, sy, sx, total);
  else
    hipLaunchKernelGGL(resize_nearest_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, c, h_in, w_in, cs_in, h_out, w_out, cs_out, sy, sx, total);
  CGAN_CHECK_LAUNCH("resize_nearest");
  return CGAN_OK;
}

extern "C" int cgan_avgpool3x3s2_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                      int32_t w_in, void* stream) {
  CGAN_REQUIRE(x && y, "avgpool3x3s2: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "avgpool3x3s2: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0, "avgpool3x3s2: bad shape");
  int h_out = (h_in + 2 - 3) / 2 + 1, w_out = (w_in + 2 - 3) / 2 + 1;
  int cs = cgan_cs(c);
  long total = (long)n * h_out * w_out * (cs / 8);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(avgpool3x3s2_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, h_in, w_in, h_out, w_out, cs, total);
  else
    hipLaunchKernelGGL(avgpool3x3s2_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, h_in, w_in, h_out, w_out, cs, total);
  CGAN_CHECK_LAUNCH("avgpool3x3s2");
  return CGAN_OK;
}

extern "C" int cgan_maxpool3x3s2_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                      int32_t w_in, void* stream) {
  CGAN_REQUIRE(x && y, "maxpool3x3s2: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "maxpool3x3s2: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0, "maxpool3x3s2: bad shape");
  int h_out = (h_in + 2 - 3) / 2 + 1, w_out = (w_in + 2 - 3) / 2 + 1;
  int cs = cgan_cs(c);
  long total = (long)n * h_out * w_out * (cs / 8);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(maxpool3x3s2_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, h_in, w_in, h_out, w_out, cs, total);
  else
    hipLaunchKernelGGL(maxpool3x3s2_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, h_in, w_in, h_out, w_out, cs, total);
  CGAN_CHECK_LAUNCH("maxpool3x3s2");
  return CGAN_OK;
}

extern "C" int cgan_resize_bilinear_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                         int32_t w_in, int32_t h_out, int32_t w_out, int32_t align_corners,
                                         void* stream) {
  CGAN_REQUIRE(x && y, "resize_bilinear: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "resize_bilinear: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "resize_bilinear: bad shape");
  int cs = cgan_cs(c);
  long total = (long)n * h_out * w_out * (cs / 8);
  float sy, sx;
  if (align_corners) {
    sy = h_out > 1 ? (float)(h_in - 1) / (float)(h_out - 1) : 0.f;
    sx = w_out > 1 ? (float)(w_in - 1) / (float)(w_out - 1) : 0.f;
  } else {
    sy = (float)h_in / (float)h_out;
    sx = (float)w_in / (float)w_out;
  }
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(resize_bilinear_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, h_in, w_in, h_out, w_out, cs, sy, sx, align_corners, total);
  else
    hipLaunchKernelGGL(resize_bilinear_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, h_in, w_in, h_out, w_out, cs, sy, sx, align_corners, total);
  CGAN_CHECK_LAUNCH("resize_bilinear");
  return CGAN_OK;
}

extern "C" int cgan_resize_bicubic_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                        int32_t w_in, int32_t h_out, int32_t w_out, void* stream) {
  CGAN_REQUIRE(x && y, "resize_bicubic: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "resize_bicubic: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "resize_bicubic: bad shape");
  int cs = cgan_cs(c);
  long total = (long)n * h_out * w_out * (cs / 8);
  float sy = (float)h_in / (float)h_out, sx = (float)w_in / (float)w_out;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(resize_bicubic_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, h_in, w_in, h_out, w_out, cs, sy, sx, total);
  else
    hipLaunchKernelGGL(resize_bicubic_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x,
                       (uint16_t*)y, h_in, w_in, h_out, w_out, cs, sy, sx, total);
  CGAN_CHECK_LAUNCH("resize_bicubic");
  return CGAN_OK;
}

extern "C" int cgan_resize_bicubic_bwd_nhwc(const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                            int32_t w_in, int32_t h_out, int32_t w_out, void* stream) {
  CGAN_REQUIRE(dy && dx, "resize_bicubic_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "resize_bicubic_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "resize_bicubic_bwd: bad shape");
  const int cs = cgan_cs(c);
  const long total = (long)n * h_in * w_in * (cs / 8);
  const float sy = (float)h_in / (float)h_out, sx = (float)w_in / (float)w_out;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(resize_bicubic_bwd_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)dy,
                       (uint16_t*)dx, h_in, w_in, h_out, w_out, cs, sy, sx, total);
  else
    hipLaunchKernelGGL(resize_bicubic_bwd_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)dy,
                       (uint16_t*)dx, h_in, w_in, h_out, w_out, cs, sy, sx, total);
  CGAN_CHECK_LAUNCH("resize_bicubic_bwd");
  return CGAN_OK;
}

extern "C" int cgan_sumpool2x2_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_out,
                                    int32_t w_out, void* stream) {
  CGAN_REQUIRE(x && y, "sumpool2x2: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "sumpool2x2: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_out > 0 && w_out > 0, "sumpool2x2: bad shape");
  int cs = cgan_cs(c);
  long total = (long)n * h_out * w_out * (cs / 8);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(sumpool2x2_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)y,
                       h_out, w_out, cs, total);
  else
    hipLaunchKernelGGL(sumpool2x2_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)y,
                       h_out, w_out, cs, total);
  CGAN_CHECK_LAUNCH("sumpool2x2");
  return CGAN_OK;
}

extern "C" int cgan_reflect_pad_bwd_nhwc(const void* dx_padded, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h,
                                         int32_t w, int32_t pad, void* stream) {
  CGAN_REQUIRE(dx_padded && dx, "reflect_pad_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "reflect_pad_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0 && pad >= 0 && pad < h && pad < w, "reflect_pad_bwd: bad shape");
  const int cs = cgan_cs(c);
  const long total = (long)n * h * w * (cs / 8);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(reflect_pad_bwd_kernel<F16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)dx_padded,
                       (uint16_t*)dx, h, w, pad, cs, total);
  else
    hipLaunchKernelGGL(reflect_pad_bwd_kernel<BF16>, dim3(grid_for(total)), dim3(256), 0, s, (const uint16_t*)dx_padded,
                       (uint16_t*)dx, h, w, pad, cs, total);
  CGAN_CHECK_LAUNCH("reflect_pad_bwd");
  return CGAN_OK;
}

extern "C" int cgan_copy_channels_nhwc(const void* src, void* dst, int64_t npix, int32_t c, int32_t cs_src,
                                       int32_t cs_dst, int32_t c_off, void* stream) {
  CGAN_REQUIRE(src && dst, "copy_channels: null pointer");
  CGAN_REQUIRE(npix > 0 && c > 0 && (cs_src % 8) == 0 && (cs_dst % 8) == 0 && (c_off % 8) == 0 && cs_src >= c &&
                   c_off + cgan_cs(c) <= cs_dst,
               "copy_channels: bad channel layout");
  int groups = cgan_cs(c) / 8;
  long total = npix * groups;
  hipLaunchKernelGGL(copy_channels_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, (uint16_t*)dst, cs_src, cs_dst, c_off, groups, total);
  CGAN_CHECK_LAUNCH("copy_channels");
  return CGAN_OK;
}

extern "C" int cgan_eltwise_nhwc(const void* a, const void* b, void* y, int32_t dtype, int32_t op, int64_t numel,
                                 void* stream) {
  CGAN_REQUIRE(a && y && (op == 1 || b), "eltwise: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "eltwise: bad dtype %d", dtype);
  CGAN_REQUIRE(op >= 0 && op <= 2, "eltwise: unknown op %d", op);
  CGAN_REQUIRE(numel > 0 && (numel % 8) == 0, "eltwise: numel must be a positive multiple of 8");
  long groups = numel / 8;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(eltwise_kernel<F16>, dim3(grid_for(groups)), dim3(256), 0, s, (const uint16_t*)a,
                       (const uint16_t*)b, (uint16_t*)y, op, groups);
  else
    hipLaunchKernelGGL(eltwise_kernel<BF16>, dim3(grid_for(groups)), dim3(256), 0, s, (const uint16_t*)a,
                       (const uint16_t*)b, (uint16_t*)y, op, groups);
  CGAN_CHECK_LAUNCH("eltwise");
  return CGAN_OK;
}

extern "C" int cgan_fold_bn(const float* w, const float* bias, const float* gamma, const float* beta, const float* mean,
                            const float* var, float eps, float* w_out, float* b_out, int32_t c_out, int64_t per_out,
                            void* stream) {
  CGAN_REQUIRE(w && mean && var && w_out && b_out, "fold_bn: null pointer");
  CGAN_REQUIRE(c_out > 0 && per_out > 0, "fold_bn: bad shape");
  long total = (long)c_out * per_out;
  hipLaunchKernelGGL(fold_bn_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, bias, gamma, beta,
                     mean, var, eps, w_out, b_out, c_out, (long)per_out);
  CGAN_CHECK_LAUNCH("fold_bn");
  return CGAN_OK;
}
