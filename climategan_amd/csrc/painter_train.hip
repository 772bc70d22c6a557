// Glue kernels of the Painter's training step (reference trainer.py:1256-1387, 1073-1107): the tensors derived from
// the painted image for the discriminator and the VGG loss, their backward, and the pooling backward passes.
#include "cgan_common.h"

namespace {

__host__ __device__ inline int grid_pt(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// p = fake ? x (1 - m) + fake m : x          (OmniGenerator.paint's paste, generator.py:295-296)
// d_in  = [m, p_r, p_g, p_b]                  (torch.cat([m, x], axis=1), trainer.py:1101-1102)        4 ch -> cs 8
// vgg_in = vgg_preprocess(p * m)              (tutils.py:416-427: BGR, [0,255], mean-subtracted)        3 ch -> cs 8
template <typename T>
__global__ void painter_heads_fwd_kernel(const uint16_t* __restrict__ fake, const float* __restrict__ x,
                                         const float* __restrict__ m, uint16_t* __restrict__ d_in,
                                         uint16_t* __restrict__ vgg_in, long hw, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, p = i - n * hw;
    const float mv = m[i];
    float pc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xv = x[(n * 3 + c) * hw + p];
      pc[c] = fake ? xv * (1.f - mv) + f32_of_bits<T>(fake[i * 8 + c]) * mv : xv;
    }
    if (d_in) {
      u32x4 o;
      o[0] = pack2<T>(mv, pc[0]);
      o[1] = pack2<T>(pc[1], pc[2]);
      o[2] = 0u;
      o[3] = 0u;
      reinterpret_cast<u32x4*>(d_in)[i] = o;
    }
    if (vgg_in) {
      const float b = (pc[2] * mv + 1.f) * 255.f * 0.5f - 103.939f;
      const float g = (pc[1] * mv + 1.f) * 255.f * 0.5f - 116.779f;
      const float r = (pc[0] * mv + 1.f) * 255.f * 0.5f - 123.680f;
      u32x4 o;
      o[0] = pack2<T>(b, g);
      o[1] = pack2<T>(r, 0.f);
      o[2] = 0u;
      o[3] = 0u;
      reinterpret_cast<u32x4*>(vgg_in)[i] = o;
    }
  }
}

// d_fake[c] = m * (d_d_in[1 + c] + 127.5 * m * d_vgg_in[2 - c])
template <typename T>
__global__ void painter_heads_bwd_kernel(const uint16_t* __restrict__ dd, const uint16_t* __restrict__ dv,
                                         const float* __restrict__ m, uint16_t* __restrict__ dfake, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float mv = m[i];
    float g[3] = {0.f, 0.f, 0.f};
    if (dd) {
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] += f32_of_bits<T>(dd[i * 8 + 1 + c]);
    }
    if (dv) {
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] += 127.5f * mv * f32_of_bits<T>(dv[i * 8 + 2 - c]);
    }
    u32x4 o;
    o[0] = pack2<T>(g[0] * mv, g[1] * mv);
    o[1] = pack2<T>(g[2] * mv, 0.f);
    o[2] = 0u;
    o[3] = 0u;
    reinterpret_cast<u32x4*>(dfake)[i] = o;
  }
}

// backward of nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False): every input pixel collects dy / count
// from the (at most 4) windows that contain it
template <typename T>
__global__ void avgpool3x3s2_bwd_kernel(const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, int h_in, int w_in,
                                        int h_out, int w_out, int cs, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int ix = (int)(pix % w_in);
    const long r = pix / w_in;
    const int iy = (int)(r % h_in);
    const long n = r / h_in;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // windows oy with 2 oy - 1 <= iy <= 2 oy + 1
    for (int oy = (iy) / 2; oy <= (iy + 1) / 2; ++oy) {
      if (oy < 0 || oy >= h_out) continue;
      const int y0 = max(2 * oy - 1, 0), y1 = min(2 * oy + 1, h_in - 1);
      for (int ox = (ix) / 2; ox <= (ix + 1) / 2; ++ox) {
        if (ox < 0 || ox >= w_out) continue;
        const int x0 = max(2 * ox - 1, 0), x1 = min(2 * ox + 1, w_in - 1);
        const float inv = 1.f / (float)((y1 - y0 + 1) * (x1 - x0 + 1));
        const u32x4 v = *reinterpret_cast<const u32x4*>(dy + ((n * h_out + oy) * (long)w_out + ox) * cs + cg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a, b;
          unpack2<T>(v[e], a, b);
          acc[2 * e] += a * inv;
          acc[2 * e + 1] += b * inv;
        }
      }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(acc[2 * e], acc[2 * e + 1]);
    *reinterpret_cast<u32x4*>(dx + pix * cs + cg * 8) = o;
  }
}

// nn.MaxPool2d(2, 2) forward / backward (VGG19 features 4, 9, 18, 27).  Backward routes dy to the FIRST maximum of
// the window in row-major order (torch's choice) and writes zeros elsewhere (rows / columns beyond 2*h_out stay as the
// host entry zeroed them).
template <typename T>
__global__ void maxpool2x2_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int h_in, int w_in,
                                      int h_out, int w_out, int cs, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int ox = (int)(pix % w_out);
    const long r = pix / w_out;
    const int oy = (int)(r % h_out);
    const long n = r / h_out;
    const uint16_t* base = x + ((n * h_in + 2 * oy) * (long)w_in + 2 * ox) * cs + cg * 8;
    float best[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = -__builtin_inff();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(base + ((long)(k >> 1) * w_in + (k & 1)) * cs);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a, b;
        unpack2<T>(v[e], a, b);
        best[2 * e] = fmaxf(best[2 * e], a);
        best[2 * e + 1] = fmaxf(best[2 * e + 1], b);
      }
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(best[2 * e], best[2 * e + 1]);
    *reinterpret_cast<u32x4*>(y + pix * cs + cg * 8) = o;
  }
}

template <typename T>
__global__ void maxpool2x2_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                      uint16_t* __restrict__ dx, int h_in, int w_in, int h_out, int w_out, int cs,
                                      long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int ox = (int)(pix % w_out);
    const long r = pix / w_out;
    const int oy = (int)(r % h_out);
    const long n = r / h_out;
    const long ibase = ((n * h_in + 2 * oy) * (long)w_in + 2 * ox) * cs + cg * 8;
    float v[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(x + ibase + ((long)(k >> 1) * w_in + (k & 1)) * cs);
#pragma unroll
      for (int e = 0; e < 4; ++e) unpack2<T>(t[e], v[k][2 * e], v[k][2 * e + 1]);
    }
    const u32x4 g = *reinterpret_cast<const u32x4*>(dy + pix * cs + cg * 8);
    float gv[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) unpack2<T>(g[e], gv[2 * e], gv[2 * e + 1]);
    int arg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int a = 0;
      float best = v[0][e];
#pragma unroll
      for (int k = 1; k < 4; ++k)
        if (v[k][e] > best) { best = v[k][e]; a = k; }
      arg[e] = a;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = pack2<T>(arg[2 * e] == k ? gv[2 * e] : 0.f, arg[2 * e + 1] == k ? gv[2 * e + 1] : 0.f);
      *reinterpret_cast<u32x4*>(dx + ibase + ((long)(k >> 1) * w_in + (k & 1)) * cs) = o;
    }
  }
}

}  // namespace

#define DISPATCH_PT(dtype, KERNEL, ...)                                     \
  do {                                                                      \
    if ((dtype) == CGAN_F16) hipLaunchKernelGGL(KERNEL<F16>, __VA_ARGS__);  \
    else hipLaunchKernelGGL(KERNEL<BF16>, __VA_ARGS__);                     \
  } while (0)

extern "C" int cgan_painter_heads_fwd(const void* fake_nhwc, const float* x_nchw, const float* m_nchw, void* d_in,
                                      void* vgg_in, int32_t dtype, int32_t n, int32_t h, int32_t w, void* stream) {
  CGAN_REQUIRE(x_nchw && m_nchw && (d_in || vgg_in), "painter_heads_fwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "painter_heads_fwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && h > 0 && w > 0, "painter_heads_fwd: bad shape");
  const long hw = (long)h * w, total = (long)n * hw;
  DISPATCH_PT(dtype, painter_heads_fwd_kernel, dim3(grid_pt(total)), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)fake_nhwc, x_nchw, m_nchw, (uint16_t*)d_in, (uint16_t*)vgg_in, hw, total);
  CGAN_CHECK_LAUNCH("painter_heads_fwd");
  return CGAN_OK;
}

extern "C" int cgan_painter_heads_bwd(const void* d_d_in, const void* d_vgg_in, const float* m_nchw, void* d_fake,
                                      int32_t dtype, int32_t n, int32_t h, int32_t w, void* stream) {
  CGAN_REQUIRE(m_nchw && d_fake && (d_d_in || d_vgg_in), "painter_heads_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "painter_heads_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && h > 0 && w > 0, "painter_heads_bwd: bad shape");
  const long total = (long)n * h * w;
  DISPATCH_PT(dtype, painter_heads_bwd_kernel, dim3(grid_pt(total)), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)d_d_in, (const uint16_t*)d_vgg_in, m_nchw, (uint16_t*)d_fake, total);
  CGAN_CHECK_LAUNCH("painter_heads_bwd");
  return CGAN_OK;
}

extern "C" int cgan_avgpool3x3s2_bwd_nhwc(const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                          int32_t w_in, void* stream) {
  CGAN_REQUIRE(dy && dx, "avgpool3x3s2_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "avgpool3x3s2_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0, "avgpool3x3s2_bwd: bad shape");
  const int h_out = (h_in + 2 - 3) / 2 + 1, w_out = (w_in + 2 - 3) / 2 + 1;
  const int cs = cgan_cs(c);
  const long total = (long)n * h_in * w_in * (cs / 8);
  DISPATCH_PT(dtype, avgpool3x3s2_bwd_kernel, dim3(grid_pt(total)), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)dy, (uint16_t*)dx, h_in, w_in, h_out, w_out, cs, total);
  CGAN_CHECK_LAUNCH("avgpool3x3s2_bwd");
  return CGAN_OK;
}

extern "C" int cgan_maxpool2x2_nhwc(const void* x, void* y, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                    int32_t w_in, void* stream) {
  CGAN_REQUIRE(x && y, "maxpool2x2: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "maxpool2x2: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 1 && w_in > 1, "maxpool2x2: bad shape");
  const int h_out = h_in / 2, w_out = w_in / 2, cs = cgan_cs(c);
  const long total = (long)n * h_out * w_out * (cs / 8);
  DISPATCH_PT(dtype, maxpool2x2_fwd_kernel, dim3(grid_pt(total)), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)x, (uint16_t*)y, h_in, w_in, h_out, w_out, cs, total);
  CGAN_CHECK_LAUNCH("maxpool2x2");
  return CGAN_OK;
}

extern "C" int cgan_maxpool2x2_bwd_nhwc(const void* x, const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c,
                                        int32_t h_in, int32_t w_in, void* stream) {
  CGAN_REQUIRE(x && dy && dx, "maxpool2x2_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "maxpool2x2_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 1 && w_in > 1, "maxpool2x2_bwd: bad shape");
  const int h_out = h_in / 2, w_out = w_in / 2, cs = cgan_cs(c);
  hipStream_t s = (hipStream_t)stream;
  if ((h_in & 1) || (w_in & 1)) {
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)n * h_in * w_in * cs * 2, s);
    if (e != hipSuccess) {
      cgan_set_error("maxpool2x2_bwd: hipMemsetAsync failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
  }
  const long total = (long)n * h_out * w_out * (cs / 8);
  DISPATCH_PT(dtype, maxpool2x2_bwd_kernel, dim3(grid_pt(total)), dim3(256), 0, s, (const uint16_t*)x,
              (const uint16_t*)dy, (uint16_t*)dx, h_in, w_in, h_out, w_out, cs, total);
  CGAN_CHECK_LAUNCH("maxpool2x2_bwd");
  return CGAN_OK;
}
