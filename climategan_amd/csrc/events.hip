// Output post-ops of the inference harness (reference trainer.py:311-332, tutils.py:567-576, trainer.py:1870-1871):
// per-image min-max normalisation -> uint8 HWC, and mask binarisation.  HBM-bound byte work: one pass to reduce,
// one pass to convert; inputs are the NCHW float tensors the generator boundary returns.
#include "cgan_common.h"

namespace {

// order-preserving float <-> int key (signed compare)
__device__ __forceinline__ int f2key(float f) {
  int b = __float_as_int(f);
  return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float key2f(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

__global__ void minmax_init_kernel(int* ws, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    ws[2 * i] = 0x7fffffff;      // running min key
    ws[2 * i + 1] = (int)0x80000000;  // running max key
  }
}

template <bool HALF>
__device__ __forceinline__ float load_elem(const void* x, long i) {
  if (HALF) return (float)((const _Float16*)x)[i];
  return ((const float*)x)[i];
}

// grid: (blocks_per_image, n); each block strides over its image
template <bool HALF>
__global__ void __launch_bounds__(256) minmax_kernel(const void* x, int* ws, long per_image) {
  const int img = blockIdx.y;
  long base = (long)img * per_image;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_image; i += (long)gridDim.x * blockDim.x) {
    float v = load_elem<HALF>(x, base + i);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  __shared__ float smn[4], smx[4];
  int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    smn[wave] = mn;
    smx[wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mn = fminf(mn, smn[w]);
      mx = fmaxf(mx, smx[w]);
    }
    atomicMin(&ws[2 * img], f2key(mn));
    atomicMax(&ws[2 * img + 1], f2key(mx));
  }
}

__device__ __forceinline__ float rh(float v) { return (float)(_Float16)v; }

// out[img][p][ch] = uint8(trunc(((x - min) / (max - min)) * 255)); with HALF every intermediate is rounded to
// fp16 like the reference's `.half()` tensors and numpy float16 arithmetic do.
template <bool HALF>
__global__ void __launch_bounds__(256)
    normalize_u8_kernel(const void* x, const int* ws, uint8_t* out, int c, long hw, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // over n*hw pixels
  if (i >= total) return;
  long img = i / hw, p = i - img * hw;
  float mn = key2f(ws[2 * img]), mx = key2f(ws[2 * img + 1]);
  float den = mx - mn;
  if (HALF) den = rh(den);
  for (int ch = 0; ch < c; ++ch) {
    float v = load_elem<HALF>(x, (img * c + ch) * hw + p);
    float a = v - mn;
    if (HALF) a = rh(a);
    float b = __fdiv_rn(a, den);
    if (HALF) b = rh(b);
    float s = b * 255.f;
    if (HALF) s = rh(s);
    out[i * c + ch] = (uint8_t)(int)s;
  }
}

template <bool HALF>
__global__ void __launch_bounds__(256)
    binarize_kernel(const void* x, void* y, uint8_t* y_u8, float thr, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  bool on = load_elem<HALF>(x, i) > thr;
  if (y) {
    if (HALF)
      ((_Float16*)y)[i] = on ? (_Float16)1.f : (_Float16)0.f;
    else
      ((float*)y)[i] = on ? 1.f : 0.f;
  }
  if (y_u8) y_u8[i] = on ? 255 : 0;
}


// ---- smog event (Trainer.compute_smog, climategan/trainer.py:1879-1939; HazeRD model) ---------------------------
// per-image min / max of channel 0 of an NHWC 16-bit map
template <typename T>
__global__ void __launch_bounds__(256) minmax_c0_kernel(const uint16_t* __restrict__ d, int* ws, long hw, int cs) {
  const int img = blockIdx.y;
  const uint16_t* base = d + (long)img * hw * cs;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
    float v = f32_of_bits<T>(base[i * cs]);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o));
    mx = fmaxf(mx, __shfl_xor(mx, o));
  }
  __shared__ float smn[4], smx[4];
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      mn = fminf(mn, smn[w]);
      mx = fmaxf(mx, smx[w]);
    }
    atomicMin(&ws[2 * img], f2key(mn));
    atomicMax(&ws[2 * img + 1], f2key(mx));
  }
}

struct SmogParams {
  float airlight, beta, alpha, yellow[3];
};

// depth -> normalize(.., 0.3, 1) -> 1/. -> normalize(.., 0.1, 1)   (trainer.py:1908-1910), at one low-res pixel.
// The extrema of the reciprocal map follow from those of the first normalisation (1/. is monotone).
__device__ __forceinline__ float smog_depth(float d, float dmin, float dmax) {
  const float t = __fdiv_rn(d - dmin, dmax - dmin);
  const float d1 = 0.3f + 0.7f * t;
  const float inv = __fdiv_rn(1.f, d1);
  const float inv_min = __fdiv_rn(1.f, 0.3f + 0.7f * 1.f);
  const float inv_max = __fdiv_rn(1.f, 0.3f + 0.7f * 0.f);
  const float t2 = __fdiv_rn(inv - inv_min, inv_max - inv_min);
  return 0.1f + 0.9f * t2;
}

template <typename T>
__global__ void __launch_bounds__(256)
    smog_kernel(const float* __restrict__ x, const uint16_t* __restrict__ d, const int* __restrict__ ws_x,
                const int* __restrict__ ws_d, float* __restrict__ out, int h, int w, int dh, int dw, int cs,
                SmogParams prm, float sy, float sx, long total) {
  const long hw = (long)h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, p = i - n * hw;
    const int oy = (int)(p / w), ox = (int)(p - (long)oy * w);
    // bilinear, align_corners=True (trainer.py:1915-1917)
    const float fy = oy * sy, fx = ox * sx;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < dh - 1 ? y0 : dh - 1;
    x0 = x0 < dw - 1 ? x0 : dw - 1;
    const int y1 = y0 < dh - 1 ? y0 + 1 : y0, x1 = x0 < dw - 1 ? x0 + 1 : x0;
    const float ly = fy - y0, lx = fx - x0;
    const float dmin = key2f(ws_d[2 * n]), dmax = key2f(ws_d[2 * n + 1]);
    const uint16_t* db = d + n * (long)dh * dw * cs;
    const float d00 = smog_depth(f32_of_bits<T>(db[((long)y0 * dw + x0) * cs]), dmin, dmax);
    const float d01 = smog_depth(f32_of_bits<T>(db[((long)y0 * dw + x1) * cs]), dmin, dmax);
    const float d10 = smog_depth(f32_of_bits<T>(db[((long)y1 * dw + x0) * cs]), dmin, dmax);
    const float d11 = smog_depth(f32_of_bits<T>(db[((long)y1 * dw + x1) * cs]), dmin, dmax);
    const float dd = (1.f - ly) * ((1.f - lx) * d00 + lx * d01) + ly * ((1.f - lx) * d10 + lx * d11);
    const float tr = expf(-prm.beta * dd);
    const float xmin = key2f(ws_x[2 * n]), xden = key2f(ws_x[2 * n + 1]) - xmin;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xn = __fdiv_rn(x[(n * 3 + c) * hw + p] - xmin, xden);                     // tutils.normalize
      const float irr = xn <= 0.04045f ? xn / 12.92f : powf((xn + 0.055f) / 1.055f, 2.4f);  // srgb2lrgb
      const float sm = tr * irr + (1.f - tr) * prm.airlight;
      const float srgb = sm <= 0.0031308f ? 12.92f * sm : 1.055f * powf(sm, 1.f / 2.4f) - 0.055f;   // lrgb2srgb
      out[(n * 3 + c) * hw + p] = srgb * (1.f - prm.alpha) + prm.yellow[c] * prm.alpha;
    }
  }
}


// ---- cloudy painting (OmniGenerator.paint_cloudy, generator.py:299-328; tutils.mix_noise / rand_perlin_2d :647-694) --
// Perlin noise map [h][w] from (res_y+1) x (res_x+1) gradient angles; also tracks the global minimum (ws[0], as a key).
__global__ void __launch_bounds__(256)
    perlin_kernel(const float* __restrict__ angles, float* __restrict__ noise, int* __restrict__ ws, int h, int w,
                  int res_y, int res_x) {
  const double dy = (double)res_y / (double)h, dx = (double)res_x / (double)w;
  const int ty = h / res_y, tx = w / res_x;   // pixels per lattice cell (repeat_interleave factors)
  float mn = __builtin_inff();
  const long total = (long)h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int y = (int)(i / w), x = (int)(i - (long)y * w);
    const float gy = fmodf((float)(y * dy), 1.f), gx = fmodf((float)(x * dx), 1.f);   // torch.arange(...) % 1
    const int cy = y / ty, cx = x / tx;
    auto dot = [&](int iy, int ix, float sy, float sx) {
      const float a = angles[iy * (res_x + 1) + ix];
      return (gy + sy) * cosf(a) + (gx + sx) * sinf(a);
    };
    const float n00 = dot(cy, cx, 0.f, 0.f), n10 = dot(cy + 1, cx, -1.f, 0.f);
    const float n01 = dot(cy, cx + 1, 0.f, -1.f), n11 = dot(cy + 1, cx + 1, -1.f, -1.f);
    auto fade = [](float t) { return 6.f * t * t * t * t * t - 15.f * t * t * t * t + 10.f * t * t * t; };
    const float t0 = fade(gy), t1 = fade(gx);
    const float a = n00 + t0 * (n10 - n00), b = n01 + t0 * (n11 - n01);           // torch.lerp(s, e, w) = s + w (e - s)
    const float v = 1.41421356237309515f * (a + t1 * (b - a));
    noise[i] = v;
    mn = fminf(mn, v);
  }
  for (int o = 32; o > 0; o >>= 1) mn = fminf(mn, __shfl_xor(mn, o));
  if ((threadIdx.x & 63) == 0) atomicMin(&ws[0], f2key(mn));
}

// cond = noised_x * (1 - m): noised_x = sky ? weight * (noise - min) + (1 - weight) * x : x, sky = argmax_c of the
// bilinearly up-sampled (align_corners=False) segmentation logits == sky_idx.  cond: NHWC, 3 channels stored as 4.
template <typename T>
__global__ void __launch_bounds__(256)
    cloudy_cond_kernel(const float* __restrict__ x, const float* __restrict__ m, const uint16_t* __restrict__ seg,
                       const float* __restrict__ noise, const int* __restrict__ ws, uint16_t* __restrict__ cond,
                       int h, int w, int sh, int sw, int sc, int scs, int sky_idx, float weight, float ry, float rx,
                       long total) {
  const long hw = (long)h * w;
  const float nmin = key2f(ws[0]);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, p = i - n * hw;
    const int oy = (int)(p / w), ox = (int)(p - (long)oy * w);
    const float fy = fmaxf((oy + 0.5f) * ry - 0.5f, 0.f), fx = fmaxf((ox + 0.5f) * rx - 0.5f, 0.f);
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 < sh - 1 ? y0 : sh - 1;
    x0 = x0 < sw - 1 ? x0 : sw - 1;
    const int y1 = y0 < sh - 1 ? y0 + 1 : y0, x1 = x0 < sw - 1 ? x0 + 1 : x0;
    const float ly = fy - y0, lx = fx - x0;
    const uint16_t* sb = seg + n * (long)sh * sw * scs;
    const uint16_t* p00 = sb + ((long)y0 * sw + x0) * scs;
    const uint16_t* p01 = sb + ((long)y0 * sw + x1) * scs;
    const uint16_t* p10 = sb + ((long)y1 * sw + x0) * scs;
    const uint16_t* p11 = sb + ((long)y1 * sw + x1) * scs;
    int best = 0;
    float bv = -__builtin_inff();
    for (int c = 0; c < sc; ++c) {
      const float v = (1.f - ly) * ((1.f - lx) * f32_of_bits<T>(p00[c]) + lx * f32_of_bits<T>(p01[c])) +
                      ly * ((1.f - lx) * f32_of_bits<T>(p10[c]) + lx * f32_of_bits<T>(p11[c]));
      if (v > bv) { bv = v; best = c; }   // first maximum, as torch.argmax
    }
    const bool sky = best == sky_idx;
    const float nz = noise[p] - nmin;
    const float keep = 1.f - m[i];
    float o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float xv = x[(n * 3 + c) * hw + p];
      o[c] = (sky ? weight * nz + (1.f - weight) * xv : xv) * keep;
    }
    u32x2 r;
    r[0] = pack2<T>(o[0], o[1]);
    r[1] = pack2<T>(o[2], 0.f);
    reinterpret_cast<u32x2*>(cond)[i] = r;
  }
}


// ---- eval-mode BatchNorm as per-(n, c) "mean / rstd" for the normalise(+act) and fused-SPADE kernels -----------------
// y = (x - rm) / sqrt(rv + eps) * gamma + beta  ==  (x - mean') * rstd'  with rstd' = gamma / sqrt(rv + eps),
// mean' = rm - beta / rstd'  (gamma, beta may be NULL: affine=False, the param-free norm of SPADE, norms.py:152-153)
__global__ void bn_eval_stats_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ rm, const float* __restrict__ rv, float eps,
                                     float* __restrict__ mean, float* __restrict__ rstd, int n, int c, int cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * cs) return;
  const int ch = i % cs;
  float m = 0.f, r = 0.f;
  if (ch < c) {
    r = rsqrtf(rv[ch] + eps) * (gamma ? gamma[ch] : 1.f);
    m = rm[ch] - ((beta && r != 0.f) ? beta[ch] / r : 0.f);
  }
  mean[i] = m;
  rstd[i] = r;
}

// ---- conditioning of the SPADE mask decoder (OmniGenerator.make_m_cond, generator.py:196-230) ---------------------------
// cond = cat[normalize(d) (1), softmax(s, dim=1) (sc), bilinear(x -> (h, w), align_corners=True) (3)]  NHWC
template <typename T>
__global__ void __launch_bounds__(256)
    make_m_cond_kernel(const uint16_t* __restrict__ d, const uint16_t* __restrict__ seg, const float* __restrict__ x,
                       const int* __restrict__ ws_d, uint16_t* __restrict__ cond, int h, int w, int sc, int scs, int xh,
                       int xw, int with_x, int ccs, float sy, float sx, long total) {
  const long hw = (long)h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, p = i - n * hw;
    uint16_t* o = cond + i * ccs;
    const float dmin = key2f(ws_d[2 * n]), dmax = key2f(ws_d[2 * n + 1]);
    o[0] = bits_of<T>(__fdiv_rn(f32_of_bits<T>(d[i * 8]) - dmin, dmax - dmin));
    const uint16_t* sp = seg + i * scs;
    float mx = -__builtin_inff();
    for (int c = 0; c < sc; ++c) mx = fmaxf(mx, f32_of_bits<T>(sp[c]));
    float sum = 0.f;
    for (int c = 0; c < sc; ++c) sum += __expf(f32_of_bits<T>(sp[c]) - mx);
    const float inv = 1.f / sum;
    for (int c = 0; c < sc; ++c) o[1 + c] = bits_of<T>(__expf(f32_of_bits<T>(sp[c]) - mx) * inv);
    int used = 1 + sc;
    if (with_x) {
      const int oy = (int)(p / w), ox = (int)(p - (long)oy * w);
      const float fy = oy * sy, fx = ox * sx;
      int y0 = (int)fy, x0 = (int)fx;
      y0 = y0 < xh - 1 ? y0 : xh - 1;
      x0 = x0 < xw - 1 ? x0 : xw - 1;
      const int y1 = y0 < xh - 1 ? y0 + 1 : y0, x1 = x0 < xw - 1 ? x0 + 1 : x0;
      const float ly = fy - y0, lx = fx - x0;
      for (int c = 0; c < 3; ++c) {
        const float* xb = x + (n * 3 + c) * (long)xh * xw;
        const float v = (1.f - ly) * ((1.f - lx) * xb[(long)y0 * xw + x0] + lx * xb[(long)y0 * xw + x1]) +
                        ly * ((1.f - lx) * xb[(long)y1 * xw + x0] + lx * xb[(long)y1 * xw + x1]);
        o[used + c] = bits_of<T>(v);
      }
      used += 3;
    }
    for (int c = used; c < ccs; ++c) o[c] = 0;
  }
}

}  // namespace

extern "C" size_t cgan_normalize_u8_workspace_bytes(int32_t n) { return n > 0 ? (size_t)n * 2 * sizeof(int) : 0; }

extern "C" int cgan_normalize_u8_nhwc(const void* x_nchw, int32_t is_half, uint8_t* out_nhwc, int32_t n, int32_t c,
                                      int32_t h, int32_t w, void* workspace, size_t workspace_bytes, void* stream) {
  CGAN_REQUIRE(x_nchw && out_nhwc && workspace, "normalize_u8: null pointer");
  CGAN_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "normalize_u8: bad shape");
  CGAN_REQUIRE(workspace_bytes >= cgan_normalize_u8_workspace_bytes(n), "normalize_u8: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  long hw = (long)h * w, per_image = hw * c, total = (long)n * hw;
  int* ws = (int*)workspace;
  hipLaunchKernelGGL(minmax_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ws, n);
  long want = (per_image + 256 * 16 - 1) / (256 * 16);
  int bpi = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
  dim3 g1(bpi, n), g2((unsigned)((total + 255) / 256));
  if (is_half) {
    hipLaunchKernelGGL(minmax_kernel<true>, g1, dim3(256), 0, s, x_nchw, ws, per_image);
    hipLaunchKernelGGL(normalize_u8_kernel<true>, g2, dim3(256), 0, s, x_nchw, ws, out_nhwc, c, hw, total);
  } else {
    hipLaunchKernelGGL(minmax_kernel<false>, g1, dim3(256), 0, s, x_nchw, ws, per_image);
    hipLaunchKernelGGL(normalize_u8_kernel<false>, g2, dim3(256), 0, s, x_nchw, ws, out_nhwc, c, hw, total);
  }
  CGAN_CHECK_LAUNCH("normalize_u8");
  return CGAN_OK;
}

extern "C" int cgan_binarize(const void* x, int32_t is_half, void* y, uint8_t* y_u8, float threshold, int64_t numel,
                             void* stream) {
  CGAN_REQUIRE(x && (y || y_u8), "binarize: null pointer");
  CGAN_REQUIRE(numel > 0, "binarize: bad size");
  dim3 g((unsigned)((numel + 255) / 256));
  if (is_half)
    hipLaunchKernelGGL(binarize_kernel<true>, g, dim3(256), 0, (hipStream_t)stream, x, y, y_u8, threshold, (long)numel);
  else
    hipLaunchKernelGGL(binarize_kernel<false>, g, dim3(256), 0, (hipStream_t)stream, x, y, y_u8, threshold, (long)numel);
  CGAN_CHECK_LAUNCH("binarize");
  return CGAN_OK;
}

extern "C" size_t cgan_smog_workspace_bytes(int32_t n) { return n > 0 ? (size_t)n * 4 * sizeof(int) : 0; }

extern "C" int cgan_smog_nchw(const float* x_nchw, const void* depth_nhwc, int32_t dtype, float* out_nchw, int32_t n,
                              int32_t h, int32_t w, int32_t dh, int32_t dw, float airlight, float beta, float alpha,
                              const float* yellow_rgb01, void* workspace, size_t workspace_bytes, void* stream) {
  CGAN_REQUIRE(x_nchw && depth_nhwc && out_nchw && yellow_rgb01 && workspace, "smog: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "smog: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && h > 0 && w > 0 && dh > 0 && dw > 0, "smog: bad shape");
  CGAN_REQUIRE(workspace_bytes >= cgan_smog_workspace_bytes(n), "smog: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int* ws_x = (int*)workspace;
  int* ws_d = ws_x + 2 * n;
  hipLaunchKernelGGL(minmax_init_kernel, dim3((2 * n + 255) / 256), dim3(256), 0, s, ws_x, 2 * n);
  const long per_image = 3L * h * w;
  long want = (per_image + 256 * 16 - 1) / (256 * 16);
  int bpi = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
  hipLaunchKernelGGL(minmax_kernel<false>, dim3(bpi, n), dim3(256), 0, s, (const void*)x_nchw, ws_x, per_image);
  const long dhw = (long)dh * dw;
  want = (dhw + 256 * 4 - 1) / (256 * 4);
  bpi = (int)(want < 1 ? 1 : (want > 64 ? 64 : want));
  const int cs = 8;   // one-channel map stored with 8 channels
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(minmax_c0_kernel<F16>, dim3(bpi, n), dim3(256), 0, s, (const uint16_t*)depth_nhwc, ws_d, dhw, cs);
  else
    hipLaunchKernelGGL(minmax_c0_kernel<BF16>, dim3(bpi, n), dim3(256), 0, s, (const uint16_t*)depth_nhwc, ws_d, dhw, cs);
  SmogParams prm;
  prm.airlight = airlight; prm.beta = beta; prm.alpha = alpha;
  prm.yellow[0] = yellow_rgb01[0]; prm.yellow[1] = yellow_rgb01[1]; prm.yellow[2] = yellow_rgb01[2];
  const float sy = h > 1 ? (float)(dh - 1) / (float)(h - 1) : 0.f, sx = w > 1 ? (float)(dw - 1) / (float)(w - 1) : 0.f;
  const long total = (long)n * h * w;
  const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(smog_kernel<F16>, dim3(grid), dim3(256), 0, s, x_nchw, (const uint16_t*)depth_nhwc, ws_x, ws_d,
                       out_nchw, h, w, dh, dw, cs, prm, sy, sx, total);
  else
    hipLaunchKernelGGL(smog_kernel<BF16>, dim3(grid), dim3(256), 0, s, x_nchw, (const uint16_t*)depth_nhwc, ws_x, ws_d,
                       out_nchw, h, w, dh, dw, cs, prm, sy, sx, total);
  CGAN_CHECK_LAUNCH("smog");
  return CGAN_OK;
}

extern "C" size_t cgan_cloudy_cond_workspace_bytes(int32_t h, int32_t w) {
  return (h > 0 && w > 0) ? (size_t)h * w * sizeof(float) + 16 : 0;
}

extern "C" int cgan_cloudy_cond_nhwc(const float* x_nchw, const float* m_nchw, const void* seg_nhwc,
                                     const float* angles, void* cond_nhwc, int32_t dtype, int32_t n, int32_t h,
                                     int32_t w, int32_t seg_h, int32_t seg_w, int32_t seg_c, int32_t sky_idx,
                                     int32_t res_y, int32_t res_x, float weight, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  CGAN_REQUIRE(x_nchw && m_nchw && seg_nhwc && angles && cond_nhwc && workspace, "cloudy_cond: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "cloudy_cond: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && h > 0 && w > 0 && seg_h > 0 && seg_w > 0 && seg_c > 0, "cloudy_cond: bad shape");
  CGAN_REQUIRE(res_y > 0 && res_x > 0 && (h % res_y) == 0 && (w % res_x) == 0,
               "cloudy_cond: the image extent must be a multiple of the Perlin resolution (%d, %d)", res_y, res_x);
  CGAN_REQUIRE(sky_idx >= 0 && sky_idx < seg_c, "cloudy_cond: bad sky index");
  CGAN_REQUIRE(workspace_bytes >= cgan_cloudy_cond_workspace_bytes(h, w), "cloudy_cond: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int* ws = (int*)workspace;                 // [0]: min key (+ padding to 16 B), then the noise map
  float* noise = (float*)((char*)workspace + 16);
  hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(256), 0, s, ws, 1);
  const long hw = (long)h * w;
  hipLaunchKernelGGL(perlin_kernel, dim3((unsigned)((hw + 255) / 256 > 2048 ? 2048 : (hw + 255) / 256)), dim3(256), 0, s,
                     angles, noise, ws, h, w, res_y, res_x);
  const long total = (long)n * hw;
  const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  const float ry = (float)seg_h / (float)h, rx = (float)seg_w / (float)w;
  const int scs = cgan_cs(seg_c);
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(cloudy_cond_kernel<F16>, dim3(grid), dim3(256), 0, s, x_nchw, m_nchw, (const uint16_t*)seg_nhwc,
                       (const float*)noise, (const int*)ws, (uint16_t*)cond_nhwc, h, w, seg_h, seg_w, seg_c, scs, sky_idx,
                       weight, ry, rx, total);
  else
    hipLaunchKernelGGL(cloudy_cond_kernel<BF16>, dim3(grid), dim3(256), 0, s, x_nchw, m_nchw, (const uint16_t*)seg_nhwc,
                       (const float*)noise, (const int*)ws, (uint16_t*)cond_nhwc, h, w, seg_h, seg_w, seg_c, scs, sky_idx,
                       weight, ry, rx, total);
  CGAN_CHECK_LAUNCH("cloudy_cond");
  return CGAN_OK;
}

extern "C" int cgan_bn_eval_stats(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* mean, float* rstd, int32_t n, int32_t c,
                                  void* stream) {
  CGAN_REQUIRE(running_mean && running_var && mean && rstd, "bn_eval_stats: null pointer");
  CGAN_REQUIRE(n > 0 && c > 0, "bn_eval_stats: bad shape");
  const int cs = cgan_cs(c);
  hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((n * cs + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                     running_mean, running_var, eps, mean, rstd, n, c, cs);
  CGAN_CHECK_LAUNCH("bn_eval_stats");
  return CGAN_OK;
}

extern "C" size_t cgan_make_m_cond_workspace_bytes(int32_t n) { return n > 0 ? (size_t)n * 2 * sizeof(int) : 0; }

extern "C" int cgan_make_m_cond_nhwc(const void* depth_nhwc, const void* seg_nhwc, const float* x_nchw, void* cond_nhwc,
                                     int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t seg_c, int32_t x_h,
                                     int32_t x_w, void* workspace, size_t workspace_bytes, void* stream) {
  CGAN_REQUIRE(depth_nhwc && seg_nhwc && cond_nhwc && workspace, "make_m_cond: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "make_m_cond: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && h > 0 && w > 0 && seg_c > 0, "make_m_cond: bad shape");
  CGAN_REQUIRE(!x_nchw || (x_h > 0 && x_w > 0), "make_m_cond: bad x shape");
  CGAN_REQUIRE(workspace_bytes >= cgan_make_m_cond_workspace_bytes(n), "make_m_cond: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int* ws = (int*)workspace;
  hipLaunchKernelGGL(minmax_init_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ws, n);
  const long hw = (long)h * w;
  long want = (hw + 256 * 4 - 1) / (256 * 4);
  const int bpi = (int)(want < 1 ? 1 : (want > 64 ? 64 : want));
  const int cond_c = 1 + seg_c + (x_nchw ? 3 : 0);
  const int ccs = (cond_c + 3) & ~3;
  const float sy = (x_nchw && h > 1) ? (float)(x_h - 1) / (float)(h - 1) : 0.f;
  const float sx = (x_nchw && w > 1) ? (float)(x_w - 1) / (float)(w - 1) : 0.f;
  const long total = (long)n * hw;
  const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
  if (dtype == CGAN_F16) {
    hipLaunchKernelGGL(minmax_c0_kernel<F16>, dim3(bpi, n), dim3(256), 0, s, (const uint16_t*)depth_nhwc, ws, hw, 8);
    hipLaunchKernelGGL(make_m_cond_kernel<F16>, dim3(grid), dim3(256), 0, s, (const uint16_t*)depth_nhwc,
                       (const uint16_t*)seg_nhwc, x_nchw, (const int*)ws, (uint16_t*)cond_nhwc, h, w, seg_c,
                       cgan_cs(seg_c), x_h, x_w, x_nchw ? 1 : 0, ccs, sy, sx, total);
  } else {
    hipLaunchKernelGGL(minmax_c0_kernel<BF16>, dim3(bpi, n), dim3(256), 0, s, (const uint16_t*)depth_nhwc, ws, hw, 8);
    hipLaunchKernelGGL(make_m_cond_kernel<BF16>, dim3(grid), dim3(256), 0, s, (const uint16_t*)depth_nhwc,
                       (const uint16_t*)seg_nhwc, x_nchw, (const int*)ws, (uint16_t*)cond_nhwc, h, w, seg_c,
                       cgan_cs(seg_c), x_h, x_w, x_nchw ? 1 : 0, ccs, sy, sx, total);
  }
  CGAN_CHECK_LAUNCH("make_m_cond");
  return CGAN_OK;
}
