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
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    int g = (int)(idx % g_out);
    long pix = idx / g_out;
    int ox = (int)(pix % w_out);
    long r = pix / w_out;
    int oy = (int)(r % h_out);
    long n = r / h_out;
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
    *reinterpret_cast<u32x2*>(y + pix * cs_out + g * 4) = pk;
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
