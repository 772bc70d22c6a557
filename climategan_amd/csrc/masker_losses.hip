// Masker-side losses (climategan/losses.py:106-196, 444-524) with their gradients, on the NHWC 16-bit maps the
// decoders produce.  Every entry point ACCUMULATES `weight * sum(...)` into a device fp32 scalar and (optionally)
// writes the gradient of that term.  HBM-bound elementwise / reduction work.
#include "cgan_common.h"

namespace {

inline int grid_ml(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

__device__ __forceinline__ void block_add(float v, float* dst) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dst, part[0] + part[1] + part[2] + part[3]);
}

// ---- channel softmax (torch.softmax(s, dim=1)) and its backward; sigmoid pair [p, 1 - p] (trainer.py:1533-1534) ----
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                          int c, int cs, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const uint16_t* xp = x + i * cs;
    float mx = -__builtin_inff();
    for (int k = 0; k < c; ++k) mx = fmaxf(mx, f32_of_bits<T>(xp[k]));
    float sum = 0.f;
    for (int k = 0; k < c; ++k) sum += __expf(f32_of_bits<T>(xp[k]) - mx);
    const float inv = 1.f / sum;
    for (int k = 0; k < cs; ++k) y[i * cs + k] = k < c ? bits_of<T>(__expf(f32_of_bits<T>(xp[k]) - mx) * inv) : 0;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const uint16_t* __restrict__ y, const uint16_t* __restrict__ dy,
                                                          uint16_t* __restrict__ dx, int c, int cs, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    float dot = 0.f;
    for (int k = 0; k < c; ++k) dot += f32_of_bits<T>(y[i * cs + k]) * f32_of_bits<T>(dy[i * cs + k]);
    for (int k = 0; k < cs; ++k)
      dx[i * cs + k] = k < c ? bits_of<T>(f32_of_bits<T>(y[i * cs + k]) * (f32_of_bits<T>(dy[i * cs + k]) - dot)) : 0;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void sigmoid_pair_fwd_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                               long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float p = 1.f / (1.f + __expf(-f32_of_bits<T>(x[i * 8])));
    u32x4 o = (u32x4){pack2<T>(p, 1.f - p), 0u, 0u, 0u};
    reinterpret_cast<u32x4*>(y)[i] = o;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void sigmoid_pair_bwd_kernel(const uint16_t* __restrict__ y, const uint16_t* __restrict__ dy,
                                                               uint16_t* __restrict__ dx, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float p = f32_of_bits<T>(y[i * 8]);
    const float g = (f32_of_bits<T>(dy[i * 8]) - f32_of_bits<T>(dy[i * 8 + 1])) * p * (1.f - p);
    u32x4 o = (u32x4){pack2<T>(g, 0.f), 0u, 0u, 0u};
    reinterpret_cast<u32x4*>(dx)[i] = o;
  }
}

// ---- nn.CrossEntropyLoss(logits, target) (losses.py:106-112): mean over pixels of logsumexp(x) - x[target] ----------
template <typename T>
__global__ __launch_bounds__(256) void softmax_ce_kernel(const uint16_t* __restrict__ x, const long long* __restrict__ tgt,
                                                         float weight, float* __restrict__ loss, uint16_t* __restrict__ dx,
                                                         int c, int cs, long npix) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const uint16_t* xp = x + i * cs;
    const int t = (int)tgt[i];
    float mx = -__builtin_inff();
    for (int k = 0; k < c; ++k) mx = fmaxf(mx, f32_of_bits<T>(xp[k]));
    float sum = 0.f;
    for (int k = 0; k < c; ++k) sum += __expf(f32_of_bits<T>(xp[k]) - mx);
    acc += mx + __logf(sum) - f32_of_bits<T>(xp[t]);
    if (dx) {
      const float inv = 1.f / sum;
      for (int k = 0; k < cs; ++k)
        dx[i * cs + k] = k < c ? bits_of<T>(weight * (__expf(f32_of_bits<T>(xp[k]) - mx) * inv - (k == t ? 1.f : 0.f))) : 0;
    }
  }
  block_add(acc * weight, loss);
}

// ---- TVLoss (losses.py:142-169): wh * sum (x[y] - x[y-1])^2 + ww * sum (x[x] - x[x-1])^2 ----------------------------
template <typename T>
__global__ __launch_bounds__(256) void tv_kernel(const uint16_t* __restrict__ x, float wh, float ww, float* __restrict__ loss,
                                                 uint16_t* __restrict__ dx, int h, int w, int c, int cs, long total) {
  float acc = 0.f;
  // (32-bit index arithmetic: a map below 2 GiB has fewer than 2^30 elements; the 64-bit divisions per element cost 10 % of this kernel)
  for (long il = (long)blockIdx.x * blockDim.x + threadIdx.x; il < total; il += (long)gridDim.x * blockDim.x) {
    const unsigned i = (unsigned)il;
    const unsigned pix = i / (unsigned)cs;
    const int k = (int)(i - pix * (unsigned)cs);
    if (k >= c) {
      if (dx) dx[i] = 0;
      continue;
    }
    const unsigned row = pix / (unsigned)w;
    const int xx = (int)(pix - row * (unsigned)w), yy = (int)(row % (unsigned)h);
    const float v = f32_of_bits<T>(x[i]);
    float g = 0.f;
    if (yy > 0) {
      const float d = v - f32_of_bits<T>(x[i - (unsigned)(w * cs)]);
      acc += wh * d * d;
      g += 2.f * wh * d;
    }
    if (yy < h - 1) g -= 2.f * wh * (f32_of_bits<T>(x[i + (unsigned)(w * cs)]) - v);
    if (xx > 0) {
      const float d = v - f32_of_bits<T>(x[i - cs]);
      acc += ww * d * d;
      g += 2.f * ww * d;
    }
    if (xx < w - 1) g -= 2.f * ww * (f32_of_bits<T>(x[i + cs]) - v);
    if (dx) dx[i] = bits_of<T>(g);
  }
  block_add(acc, loss);
}

// ---- entropy maps: prob_2_entropy (losses.py:453-458) and MinentLoss (losses.py:172-196) ----------------------------
__device__ __forceinline__ float ent(float p, float inv_log2c) { return -p * log2f(p + 1e-30f) * inv_log2c; }
__device__ __forceinline__ float ent_grad(float p, float inv_log2c) {
  return -(log2f(p + 1e-30f) + p / ((p + 1e-30f) * 0.6931471805599453f)) * inv_log2c;
}

// y = ent(p) [* depth]  (the DADA weighting of trainer.py:1455-1456); backward dp = dy * ent'(p) [* depth]
template <typename T>
__global__ __launch_bounds__(256) void entropy_fwd_kernel(const uint16_t* __restrict__ p, const uint16_t* __restrict__ depth,
                                                          uint16_t* __restrict__ y, int c, int cs, long total) {
  const float ilc = 1.f / log2f((float)c);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % cs);
    const long pix = i / cs;
    float v = 0.f;
    if (k < c) {
      v = ent(f32_of_bits<T>(p[i]), ilc);
      if (depth) v *= f32_of_bits<T>(depth[pix * 8]);
    }
    y[i] = bits_of<T>(v);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void entropy_bwd_kernel(const uint16_t* __restrict__ p, const uint16_t* __restrict__ depth,
                                                          const uint16_t* __restrict__ dy, uint16_t* __restrict__ dp, int c,
                                                          int cs, long total) {
  const float ilc = 1.f / log2f((float)c);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % cs);
    const long pix = i / cs;
    float v = 0.f;
    if (k < c) {
      v = f32_of_bits<T>(dy[i]) * ent_grad(f32_of_bits<T>(p[i]), ilc);
      if (depth) v *= f32_of_bits<T>(depth[pix * 8]);
    }
    dp[i] = bits_of<T>(v);
  }
}
// ---- the ADVENT discriminators' input straight from the logits, as a 16-bit PAIR --------------------------------------
// ent(softmax(s)) [* depth] (trainer.py:1433, 1455-1456, losses.py:453-458, 517-519), or ent([sigmoid(x), 1 - sigmoid(x)])
// for the mask (trainer.py:1533-1534), evaluated in fp32 from the 16-bit logits and stored as hi = round16(v),
// lo = round16(v - hi) in channels [0, C) and [C, 2C) of the output.  An untrained prediction has p ~ 1/C, i.e.
// ent = const - O((p - 1/C)^2): one 16-bit value (8 or 11 significant bits) is coarser than the signal the discriminator
// is meant to see.  The discriminator's first conv runs on the 2C channels with its weights duplicated.
constexpr int ADV_MAXC = 16;
template <typename T, bool SIGMOID>
__device__ __forceinline__ int advent_probs(const uint16_t* __restrict__ xp, int c, float* pr) {
  if (SIGMOID) {
    const float x = f32_of_bits<T>(xp[0]);
    pr[0] = 1.f / (1.f + __expf(-x));
    pr[1] = 1.f / (1.f + __expf(x));
    return 2;
  }
  float mx = -__builtin_inff();
  for (int k = 0; k < c; ++k) mx = fmaxf(mx, f32_of_bits<T>(xp[k]));
  float sum = 0.f;
  for (int k = 0; k < c; ++k) {
    pr[k] = __expf(f32_of_bits<T>(xp[k]) - mx);
    sum += pr[k];
  }
  const float inv = 1.f / sum;
  for (int k = 0; k < c; ++k) pr[k] *= inv;
  return c;
}
template <typename T, bool SIGMOID>
__global__ __launch_bounds__(256) void advent_pair_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ depth,
                                                              uint16_t* __restrict__ y, int c, int cs_in, int cs_out,
                                                              long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    float pr[ADV_MAXC];
    const int C = advent_probs<T, SIGMOID>(x + i * cs_in, c, pr);
    const float ilc = 1.f / log2f((float)C);
    const float dv = depth ? f32_of_bits<T>(depth[i * 8]) : 1.f;
    uint16_t* yp = y + i * cs_out;
    for (int k = 0; k < C; ++k) {
      const float v = ent(pr[k], ilc) * dv;
      const uint16_t hi = bits_of<T>(v);
      yp[k] = hi;
      yp[C + k] = bits_of<T>(v - f32_of_bits<T>(hi));
    }
    for (int k = 2 * C; k < cs_out; ++k) yp[k] = 0;
  }
}
// d(logits) from d(pair): the hi and the lo half of d(pair) are the same data gradient (duplicated weights): the hi half
// is read.  softmax: dl_j = p_j (g_j - sum_i g_i p_i), g_i = dy_i ent'(p_i) depth; sigmoid pair: dx = p (1 - p)(g_0 - g_1)
template <typename T, bool SIGMOID>
__global__ __launch_bounds__(256) void advent_pair_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ depth,
                                                              const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
                                                              int c, int cs_in, int cs_out, long npix) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    float pr[ADV_MAXC], g[ADV_MAXC];
    const int C = advent_probs<T, SIGMOID>(x + i * cs_in, c, pr);
    const float ilc = 1.f / log2f((float)C);
    const float dv = depth ? f32_of_bits<T>(depth[i * 8]) : 1.f;
    float dot = 0.f;
    for (int k = 0; k < C; ++k) {
      g[k] = f32_of_bits<T>(dy[i * cs_out + k]) * ent_grad(pr[k], ilc) * dv;
      dot += g[k] * pr[k];
    }
    uint16_t* dp = dx + i * cs_in;
    if (SIGMOID) {
      dp[0] = bits_of<T>(pr[0] * pr[1] * (g[0] - g[1]));
      for (int k = 1; k < cs_in; ++k) dp[k] = 0;
    } else {
      for (int k = 0; k < cs_in; ++k) dp[k] = k < c ? bits_of<T>(pr[k] * (g[k] - dot)) : 0;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void minent_sum_kernel(const uint16_t* __restrict__ p, float* __restrict__ sum, int c,
                                                         int cs, long total) {
  const float ilc = 1.f / log2f((float)c);
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    if ((int)(i % cs) < c) acc += ent(f32_of_bits<T>(p[i]), ilc);
  // one partial per block, plain store: minent_kernel adds the rows in a fixed order (the mean entropy feeds the gradient of
  // version 2: with an atomic sum it differed in the last bits from run to run)
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sum[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// version 1: loss = S / nhw.  version 2: mu = S / nhw, loss = sum(e + lam (e - mu)^2) / nhw, whose gradient is
// e'/nhw * (1 + 2 lam (e - mu) + 2 lam (c - 1) mu)  (the "mean" divides by n h w while the sums run over n c h w).
template <typename T>
__global__ __launch_bounds__(256) void minent_kernel(const uint16_t* __restrict__ p, const float* __restrict__ sum, int version,
                                                     float lam, float weight, float inv_nhw, float* __restrict__ loss,
                                                     uint16_t* __restrict__ dp, int c, int cs, long total, int sum_rows) {
  const float ilc = 1.f / log2f((float)c);
  float tot = 0.f;
  for (int r = threadIdx.x; r < sum_rows; r += 256) tot += sum[r];
  for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
  __shared__ float tpart[4];
  if ((threadIdx.x & 63) == 0) tpart[threadIdx.x >> 6] = tot;
  __syncthreads();
  const float mu = (tpart[0] + tpart[1] + tpart[2] + tpart[3]) * inv_nhw;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % cs);
    float g = 0.f;
    if (k < c) {
      const float pv = f32_of_bits<T>(p[i]);
      const float e = ent(pv, ilc);
      if (version == 1) {
        acc += e;
        g = ent_grad(pv, ilc) * inv_nhw;
      } else {
        acc += e + lam * (e - mu) * (e - mu);
        g = ent_grad(pv, ilc) * inv_nhw * (1.f + 2.f * lam * (e - mu) + 2.f * lam * (float)(c - 1) * mu);
      }
    }
    if (dp) dp[i] = bits_of<T>(weight * g);
  }
  block_add(acc * inv_nhw * weight, loss);
}

// ---- nn.BCEWithLogitsLoss against a target MAP (masker_m_loss, trainer.py:1549-1553): x NHWC 1 channel, t fp32 -------
template <typename T>
__global__ __launch_bounds__(256) void bce_map_kernel(const uint16_t* __restrict__ x, const float* __restrict__ t, float weight,
                                                      float* __restrict__ loss, uint16_t* __restrict__ dx, long npix) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const float xv = f32_of_bits<T>(x[i * 8]), tv = t[i];
    acc += fmaxf(xv, 0.f) - xv * tv + log1pf(__expf(-fabsf(xv)));
    if (dx) {
      u32x4 o = (u32x4){pack2<T>(weight * (1.f / (1.f + __expf(-xv)) - tv), 0.f), 0u, 0u, 0u};
      reinterpret_cast<u32x4*>(dx)[i] = o;
    }
  }
  block_add(acc * weight, loss);
}

// ---- GroundIntersectionLoss (losses.py:444-450): mean(1.0 * ((g - p) > 0.5)); piecewise constant: no gradient -------
template <typename T>
__global__ __launch_bounds__(256) void ground_intersection_kernel(const uint16_t* __restrict__ p, const float* __restrict__ g,
                                                                  float weight, float* __restrict__ loss, long npix) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x)
    acc += (g[i] - f32_of_bits<T>(p[i * 8])) > 0.5f ? 1.f : 0.f;
  block_add(acc * weight, loss);
}

// ---- sum(a x + b) over the logical channels (the WGAN form of ADVENTAdversarialLoss, losses.py:498-499) ------------
template <typename T>
__global__ __launch_bounds__(256) void affine_sum_kernel(const uint16_t* __restrict__ x, float a, float b, float* __restrict__ loss,
                                                         uint16_t* __restrict__ dx, int c, int cs, long total) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const bool live = (int)(i % cs) < c;
    if (live) acc += a * f32_of_bits<T>(x[i]) + b;
    if (dx) dx[i] = live ? bits_of<T>(a) : 0;
  }
  block_add(acc, loss);
}

// ---- the same pair map from fp32 NCHW PROBABILITIES (the reference's call signature: ADVENTAdversarialLoss receives
// softmax(pred) / cat[p, 1 - p] as NCHW tensors, losses.py:517-519) and its backward (fp32 NCHW d(prob)) --------------------
template <typename T>
__global__ __launch_bounds__(256) void entropy_pair_nchw_fwd_kernel(const float* __restrict__ p, const float* __restrict__ depth,
                                                                    uint16_t* __restrict__ y, int c, long hw, int cs_out,
                                                                    long npix) {
  const float ilc = 1.f / log2f((float)c);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, r = i - n * hw;
    const float dv = depth ? depth[i] : 1.f;
    uint16_t* yp = y + i * cs_out;
    for (int k = 0; k < c; ++k) {
      const float v = ent(p[(n * c + k) * hw + r], ilc) * dv;
      const uint16_t hi = bits_of<T>(v);
      yp[k] = hi;
      yp[c + k] = bits_of<T>(v - f32_of_bits<T>(hi));
    }
    for (int k = 2 * c; k < cs_out; ++k) yp[k] = 0;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void entropy_pair_nchw_bwd_kernel(const float* __restrict__ p, const float* __restrict__ depth,
                                                                    const uint16_t* __restrict__ dy, float* __restrict__ dp,
                                                                    int c, long hw, int cs_out, long npix) {
  const float ilc = 1.f / log2f((float)c);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, r = i - n * hw;
    const float dv = depth ? depth[i] : 1.f;
    for (int k = 0; k < c; ++k) {
      const long j = (n * c + k) * hw + r;
      dp[j] = f32_of_bits<T>(dy[i * cs_out + k]) * ent_grad(p[j], ilc) * dv;     // the hi half carries d/d(value)
    }
  }
}
}  // namespace

#define ML_DISPATCH(dtype, KERNEL, ...)                                     \
  do {                                                                      \
    if ((dtype) == CGAN_F16) hipLaunchKernelGGL(KERNEL<F16>, __VA_ARGS__);  \
    else hipLaunchKernelGGL(KERNEL<BF16>, __VA_ARGS__);                     \
  } while (0)
#define ML_DISPATCH2(dtype, KERNEL, FLAG, ...)                                  \
  do {                                                                          \
    if ((dtype) == CGAN_F16) hipLaunchKernelGGL((KERNEL<F16, FLAG>), __VA_ARGS__);  \
    else hipLaunchKernelGGL((KERNEL<BF16, FLAG>), __VA_ARGS__);                 \
  } while (0)
#define ML_CHECK_DT(name) CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, name ": bad dtype %d", dtype)

extern "C" int cgan_softmax_nhwc(const void* x, void* y, int32_t dtype, int64_t npix, int32_t c, void* stream) {
  CGAN_REQUIRE(x && y && npix > 0 && c > 0, "softmax: bad arguments");
  ML_CHECK_DT("softmax");
  ML_DISPATCH(dtype, softmax_fwd_kernel, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
              (uint16_t*)y, c, cgan_cs(c), (long)npix);
  CGAN_CHECK_LAUNCH("softmax");
  return CGAN_OK;
}
extern "C" int cgan_softmax_bwd_nhwc(const void* y, const void* dy, void* dx, int32_t dtype, int64_t npix, int32_t c,
                                     void* stream) {
  CGAN_REQUIRE(y && dy && dx && npix > 0 && c > 0, "softmax_bwd: bad arguments");
  ML_CHECK_DT("softmax_bwd");
  ML_DISPATCH(dtype, softmax_bwd_kernel, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)y,
              (const uint16_t*)dy, (uint16_t*)dx, c, cgan_cs(c), (long)npix);
  CGAN_CHECK_LAUNCH("softmax_bwd");
  return CGAN_OK;
}
extern "C" int cgan_sigmoid_pair_nhwc(const void* x, void* y, int32_t dtype, int64_t npix, void* stream) {
  CGAN_REQUIRE(x && y && npix > 0, "sigmoid_pair: bad arguments");
  ML_CHECK_DT("sigmoid_pair");
  ML_DISPATCH(dtype, sigmoid_pair_fwd_kernel, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
              (uint16_t*)y, (long)npix);
  CGAN_CHECK_LAUNCH("sigmoid_pair");
  return CGAN_OK;
}
extern "C" int cgan_sigmoid_pair_bwd_nhwc(const void* y, const void* dy, void* dx, int32_t dtype, int64_t npix, void* stream) {
  CGAN_REQUIRE(y && dy && dx && npix > 0, "sigmoid_pair_bwd: bad arguments");
  ML_CHECK_DT("sigmoid_pair_bwd");
  ML_DISPATCH(dtype, sigmoid_pair_bwd_kernel, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)y,
              (const uint16_t*)dy, (uint16_t*)dx, (long)npix);
  CGAN_CHECK_LAUNCH("sigmoid_pair_bwd");
  return CGAN_OK;
}
extern "C" int cgan_softmax_ce_nhwc(const void* logits, const int64_t* target, int32_t dtype, int64_t npix, int32_t c,
                                    float weight, float* loss_accum, void* dlogits, void* stream) {
  CGAN_REQUIRE(logits && target && loss_accum && npix > 0 && c > 0, "softmax_ce: bad arguments");
  ML_CHECK_DT("softmax_ce");
  ML_DISPATCH(dtype, softmax_ce_kernel, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)logits,
              (const long long*)target, weight, loss_accum, (uint16_t*)dlogits, c, cgan_cs(c), (long)npix);
  CGAN_CHECK_LAUNCH("softmax_ce");
  return CGAN_OK;
}
extern "C" int cgan_tv_nhwc(const void* x, int32_t dtype, int32_t n, int32_t h, int32_t w, int32_t c, float weight_h,
                            float weight_w, float* loss_accum, void* dx, void* stream) {
  CGAN_REQUIRE(x && loss_accum && n > 0 && h > 0 && w > 0 && c > 0, "tv: bad arguments");
  ML_CHECK_DT("tv");
  const int cs = cgan_cs(c);
  const long total = (long)n * h * w * cs;
  ML_DISPATCH(dtype, tv_kernel, dim3(grid_ml(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, weight_h,
              weight_w, loss_accum, (uint16_t*)dx, h, w, c, cs, total);
  CGAN_CHECK_LAUNCH("tv");
  return CGAN_OK;
}
extern "C" int cgan_entropy_map_nhwc(const void* p, const void* depth, void* y, int32_t dtype, int64_t npix, int32_t c,
                                     void* stream) {
  CGAN_REQUIRE(p && y && npix > 0 && c > 1, "entropy_map: bad arguments");
  ML_CHECK_DT("entropy_map");
  const int cs = cgan_cs(c);
  ML_DISPATCH(dtype, entropy_fwd_kernel, dim3(grid_ml(npix * cs)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)p,
              (const uint16_t*)depth, (uint16_t*)y, c, cs, (long)npix * cs);
  CGAN_CHECK_LAUNCH("entropy_map");
  return CGAN_OK;
}
extern "C" int cgan_entropy_map_bwd_nhwc(const void* p, const void* depth, const void* dy, void* dp, int32_t dtype,
                                         int64_t npix, int32_t c, void* stream) {
  CGAN_REQUIRE(p && dy && dp && npix > 0 && c > 1, "entropy_map_bwd: bad arguments");
  ML_CHECK_DT("entropy_map_bwd");
  const int cs = cgan_cs(c);
  ML_DISPATCH(dtype, entropy_bwd_kernel, dim3(grid_ml(npix * cs)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)p,
              (const uint16_t*)depth, (const uint16_t*)dy, (uint16_t*)dp, c, cs, (long)npix * cs);
  CGAN_CHECK_LAUNCH("entropy_map_bwd");
  return CGAN_OK;
}
extern "C" int cgan_entropy_pair_from_nchw(const float* prob, const float* depth, void* y, int32_t dtype, int32_t n,
                                           int32_t c, int32_t h, int32_t w, void* stream) {
  CGAN_REQUIRE(prob && y && n > 0 && h > 0 && w > 0, "entropy_pair_from_nchw: bad arguments");
  CGAN_REQUIRE(c > 1 && c <= ADV_MAXC, "entropy_pair_from_nchw: %d channels (2..%d)", c, ADV_MAXC);
  ML_CHECK_DT("entropy_pair_from_nchw");
  const long npix = (long)n * h * w;
  ML_DISPATCH(dtype, entropy_pair_nchw_fwd_kernel, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream, prob, depth,
              (uint16_t*)y, c, (long)h * w, cgan_cs(2 * c), npix);
  CGAN_CHECK_LAUNCH("entropy_pair_from_nchw");
  return CGAN_OK;
}
extern "C" int cgan_entropy_pair_from_nchw_bwd(const float* prob, const float* depth, const void* dy, float* dprob,
                                               int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w, void* stream) {
  CGAN_REQUIRE(prob && dy && dprob && n > 0 && h > 0 && w > 0, "entropy_pair_from_nchw_bwd: bad arguments");
  CGAN_REQUIRE(c > 1 && c <= ADV_MAXC, "entropy_pair_from_nchw_bwd: %d channels (2..%d)", c, ADV_MAXC);
  ML_CHECK_DT("entropy_pair_from_nchw_bwd");
  const long npix = (long)n * h * w;
  ML_DISPATCH(dtype, entropy_pair_nchw_bwd_kernel, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream, prob, depth,
              (const uint16_t*)dy, dprob, c, (long)h * w, cgan_cs(2 * c), npix);
  CGAN_CHECK_LAUNCH("entropy_pair_from_nchw_bwd");
  return CGAN_OK;
}
extern "C" int cgan_advent_entropy_pair_nhwc(const void* logits, const void* depth, void* y, int32_t dtype, int64_t npix,
                                             int32_t c, int32_t sigmoid_pair, void* stream) {
  CGAN_REQUIRE(logits && y && npix > 0, "advent_entropy_pair: bad arguments");
  CGAN_REQUIRE(sigmoid_pair ? c == 1 : (c > 1 && c <= ADV_MAXC), "advent_entropy_pair: %d channels (softmax: 2..%d, sigmoid pair: 1)",
               c, ADV_MAXC);
  ML_CHECK_DT("advent_entropy_pair");
  const int C = sigmoid_pair ? 2 : c;
  if (sigmoid_pair)
    ML_DISPATCH2(dtype, advent_pair_fwd_kernel, true, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream,
                 (const uint16_t*)logits, (const uint16_t*)depth, (uint16_t*)y, c, cgan_cs(c), cgan_cs(2 * C), (long)npix);
  else
    ML_DISPATCH2(dtype, advent_pair_fwd_kernel, false, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream,
                 (const uint16_t*)logits, (const uint16_t*)depth, (uint16_t*)y, c, cgan_cs(c), cgan_cs(2 * C), (long)npix);
  CGAN_CHECK_LAUNCH("advent_entropy_pair");
  return CGAN_OK;
}
extern "C" int cgan_advent_entropy_pair_bwd_nhwc(const void* logits, const void* depth, const void* dy, void* dlogits,
                                                 int32_t dtype, int64_t npix, int32_t c, int32_t sigmoid_pair, void* stream) {
  CGAN_REQUIRE(logits && dy && dlogits && npix > 0, "advent_entropy_pair_bwd: bad arguments");
  CGAN_REQUIRE(sigmoid_pair ? c == 1 : (c > 1 && c <= ADV_MAXC), "advent_entropy_pair_bwd: %d channels", c);
  ML_CHECK_DT("advent_entropy_pair_bwd");
  const int C = sigmoid_pair ? 2 : c;
  if (sigmoid_pair)
    ML_DISPATCH2(dtype, advent_pair_bwd_kernel, true, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream,
                 (const uint16_t*)logits, (const uint16_t*)depth, (const uint16_t*)dy, (uint16_t*)dlogits, c, cgan_cs(c),
                 cgan_cs(2 * C), (long)npix);
  else
    ML_DISPATCH2(dtype, advent_pair_bwd_kernel, false, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream,
                 (const uint16_t*)logits, (const uint16_t*)depth, (const uint16_t*)dy, (uint16_t*)dlogits, c, cgan_cs(c),
                 cgan_cs(2 * C), (long)npix);
  CGAN_CHECK_LAUNCH("advent_entropy_pair_bwd");
  return CGAN_OK;
}
extern "C" int cgan_minent_nhwc(const void* p, int32_t dtype, int64_t npix, int32_t c, int32_t version, float lambda_var,
                                float weight, float* loss_accum, void* dp, float* workspace, void* stream) {
  CGAN_REQUIRE(p && loss_accum && workspace && npix > 0 && c > 1, "minent: bad arguments");
  CGAN_REQUIRE(version == 1 || version == 2, "minent: version must be 1 or 2");
  ML_CHECK_DT("minent");
  hipStream_t s = (hipStream_t)stream;
  const int cs = cgan_cs(c);
  const long total = (long)npix * cs;
  const int rows = grid_ml(total);          // <= CGAN_MINENT_WORKSPACE_FLOATS
  ML_DISPATCH(dtype, minent_sum_kernel, dim3(rows), dim3(256), 0, s, (const uint16_t*)p, workspace, c, cs, total);
  ML_DISPATCH(dtype, minent_kernel, dim3(grid_ml(total)), dim3(256), 0, s, (const uint16_t*)p, (const float*)workspace,
              version, lambda_var, weight, 1.f / (float)npix, loss_accum, (uint16_t*)dp, c, cs, total, rows);
  CGAN_CHECK_LAUNCH("minent");
  return CGAN_OK;
}
extern "C" int cgan_bce_logits_map_nhwc(const void* x, const float* target, int32_t dtype, int64_t npix, float weight,
                                        float* loss_accum, void* dx, void* stream) {
  CGAN_REQUIRE(x && target && loss_accum && npix > 0, "bce_logits_map: bad arguments");
  ML_CHECK_DT("bce_logits_map");
  ML_DISPATCH(dtype, bce_map_kernel, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, target,
              weight, loss_accum, (uint16_t*)dx, (long)npix);
  CGAN_CHECK_LAUNCH("bce_logits_map");
  return CGAN_OK;
}
extern "C" int cgan_ground_intersection_nhwc(const void* p, const float* ground, int32_t dtype, int64_t npix, float weight,
                                             float* loss_accum, void* stream) {
  CGAN_REQUIRE(p && ground && loss_accum && npix > 0, "ground_intersection: bad arguments");
  ML_CHECK_DT("ground_intersection");
  ML_DISPATCH(dtype, ground_intersection_kernel, dim3(grid_ml(npix)), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)p, ground, weight, loss_accum, (long)npix);
  CGAN_CHECK_LAUNCH("ground_intersection");
  return CGAN_OK;
}
extern "C" int cgan_affine_sum_nhwc(const void* x, int32_t dtype, int64_t npix, int32_t c, float a, float b,
                                    float* loss_accum, void* dx, void* stream) {
  CGAN_REQUIRE(x && loss_accum && npix > 0 && c > 0, "affine_sum: bad arguments");
  ML_CHECK_DT("affine_sum");
  const int cs = cgan_cs(c);
  ML_DISPATCH(dtype, affine_sum_kernel, dim3(grid_ml(npix * cs)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, a,
              b, loss_accum, (uint16_t*)dx, c, cs, (long)npix * cs);
  CGAN_CHECK_LAUNCH("affine_sum");
  return CGAN_OK;
}
