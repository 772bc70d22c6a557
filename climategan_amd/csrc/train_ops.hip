// Elementwise / reduction kernels of the training path (backward of activations and instance norm, GAN and
// feature-matching losses with their gradients, spectral-norm weight-gradient transform).  All HBM-bound.
#include "cgan_common.h"

namespace {

__host__ __device__ inline int grid_for_n(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// d/dz of act, expressed through the OUTPUT of the activation
__device__ __forceinline__ float act_grad_from_out(float out, int act, float slope) {
  switch (act) {
    case CGAN_ACT_RELU: return out > 0.f ? 1.f : 0.f;
    case CGAN_ACT_LRELU: return out > 0.f ? 1.f : slope;
    case CGAN_ACT_TANH: return 1.f - out * out;
    case CGAN_ACT_SIGMOID: return out * (1.f - out);
    default: return 1.f;
  }
}

// dx = dy * act'(.)   (16-bit NHWC storage, 8 elements per thread)
template <typename T>
__global__ void act_bwd_kernel(const uint16_t* __restrict__ out, const uint16_t* __restrict__ dy,
                               uint16_t* __restrict__ dx, int act, float slope, long groups) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const u32x4 o = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(out) + i);
    const u32x4 g = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(dy) + i);
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float o0, o1, g0, g1;
      unpack2<T>(o[e], o0, o1);
      unpack2<T>(g[e], g0, g1);
      r[e] = pack2<T>(g0 * act_grad_from_out(o0, act, slope), g1 * act_grad_from_out(o1, act, slope));
    }
    CGAN_ST_STREAM(r, reinterpret_cast<u32x4*>(dx) + i);
  }
}

// ---- instance norm (+ LeakyReLU) backward -------------------------------------------------------------------
// out = act(y), y = (x - mean) * rstd.   dz = dy * act'(out), y = act^-1(out)
// dx = rstd * (dz - mean_hw(dz) - y * mean_hw(dz * y))
__device__ __forceinline__ void in_bwd_terms(float out, float dy, int act, float slope, float& dz, float& y) {
  if (act == CGAN_ACT_LRELU) {
    const bool pos = out > 0.f;
    dz = pos ? dy : dy * slope;
    y = pos ? out : out / slope;
  } else {
    dz = dy;
    y = out;
  }
}

// partial[n][chunk][cs][2] = (sum dz, sum dz*y) over this block's pixel range; grid (chunks, channel-group blocks, n).
// Plain stores, one row per block; in_bwd_finalize_kernel sums the rows in a fixed order: the result does not depend on the
// order the blocks ran in (the first version added into sums[n][cs][2] with fp32 atomics behind a memset: run-to-run
// different in the last bits, and the chunk count depended on the batch size).
template <typename T>
__global__ __launch_bounds__(256) void in_bwd_reduce_kernel(const uint16_t* __restrict__ out,
                                                            const uint16_t* __restrict__ dy, float* __restrict__ partial,
                                                            int hw, int cs, int ppb, int act, float slope) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [PL][cgb*8][2]
  const int cg_total = cs / 8;
  const int cgb = cg_total < 256 ? cg_total : 256;
  const int PL = 256 / cgb;
  const int n = blockIdx.z, cg0 = blockIdx.y * cgb;
  const int p0 = blockIdx.x * ppb, p1 = min(hw, p0 + ppb);
  const int t = threadIdx.x, cgl = t % cgb, pl = t / cgb;
  const int cg = cg0 + cgl;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  if (pl < PL && cg < cg_total) {
    const size_t base = (size_t)n * hw * cs + cg * 8;
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += PL) {
      const u32x4 o = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(out + base + (size_t)p * cs));
      const u32x4 g = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(dy + base + (size_t)p * cs));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float o0, o1, g0, g1, dz, y;
        unpack2<T>(o[e], o0, o1);
        unpack2<T>(g[e], g0, g1);
        in_bwd_terms(o0, g0, act, slope, dz, y);
        s1[2 * e] += dz; s2[2 * e] += dz * y;
        in_bwd_terms(o1, g1, act, slope, dz, y);
        s1[2 * e + 1] += dz; s2[2 * e + 1] += dz * y;
      }
    }
  }
  if (pl < PL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sm[((pl * cgb + cgl) * 8 + e) * 2] = s1[e];
      sm[((pl * cgb + cgl) * 8 + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  for (int c = t; c < cgb * 8; c += 256) {
    if (cg0 * 8 + c >= cs) continue;
    float a = 0.f, b = 0.f;
    for (int l = 0; l < PL; ++l) {
      a += sm[((l * cgb) * 8 + c) * 2];
      b += sm[((l * cgb) * 8 + c) * 2 + 1];
    }
    float* o = partial + (((size_t)n * gridDim.x + blockIdx.x) * cs + cg0 * 8 + c) * 2;
    o[0] = a;
    o[1] = b;
  }
}

// sums[n][cs][2] = sum over the chunk rows: one wave per (n, c); lane l adds rows l, l + 64, ... in order, then a fixed
// butterfly over the lanes
__global__ __launch_bounds__(256) void in_bwd_finalize_kernel(const float* __restrict__ partial, float* __restrict__ sums,
                                                              int n_total, int cs, int chunks) {
  // block = 8 channels x 32 chunk lanes of one sample (see instnorm_finalize_kernel, norm_stats.hip): 64-byte row segments,
  // 8 loads in flight, fixed-order tree
  __shared__ float red[4][8][2];
  const int cl = threadIdx.x & 7, kl = threadIdx.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cblocks = cs >> 3;
  const int n = blockIdx.x / cblocks, c = (blockIdx.x % cblocks) * 8 + cl;
  const float* rows = partial + ((size_t)n * chunks * cs + c) * 2;
  float a = 0.f, b = 0.f;
  for (int k0 = kl; k0 < chunks; k0 += 256) {
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + 32 * j;
      v[j] = k < chunks ? *reinterpret_cast<const float2*>(rows + (size_t)k * cs * 2) : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a += v[j].x;
      b += v[j].y;
    }
  }
#pragma unroll
  for (int off = 32; off >= 8; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  if (lane < 8) {
    red[wave][cl][0] = a;
    red[wave][cl][1] = b;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    a = b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a += red[w][cl][0];
      b += red[w][cl][1];
    }
    *reinterpret_cast<float2*>(sums + ((size_t)n * cs + c) * 2) = make_float2(a, b);
  }
}

// A thread keeps its 8 channels' per-(n, c) terms in registers and walks the pixels of ONE sample (blockIdx.y): no 64-bit
// index divisions and no per-element statistic loads in the loop (the flat version spent more on those than on its three maps:
// 2.2 TB/s).  Same expression, same operation order per element as before.
template <typename T>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const uint16_t* __restrict__ out, const uint16_t* __restrict__ dy,
                                                           const float* __restrict__ rstd, const float* __restrict__ sums,
                                                           uint16_t* __restrict__ dx, int hw, int cs, int c, int act,
                                                           float slope) {
  const int cg_total = cs / 8;
  const int tpp = cg_total < 256 ? cg_total : 256;   // threads per pixel
  const int rows = 256 / tpp;
  const int cgl = threadIdx.x % tpp, prow = threadIdx.x / tpp;
  if (prow >= rows) return;
  const int n = blockIdx.y;
  const float inv_hw = 1.f / (float)hw;
  const size_t nofs = (size_t)n * hw * cs;
  out += nofs; dy += nofs; dx += nofs;
  for (int cg = cgl; cg < cg_total; cg += tpp) {
    float rs[8], m1[8], m2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = cg * 8 + e;
      const float* sp = sums + ((size_t)n * cs + ch) * 2;
      rs[e] = rstd[(size_t)n * cs + ch];
      m1[e] = sp[0] * inv_hw;
      m2[e] = sp[1] * inv_hw;
    }
#pragma unroll 2
    for (int p = blockIdx.x * rows + prow; p < hw; p += gridDim.x * rows) {
      const size_t off = (size_t)p * cs + cg * 8;
      const u32x4 o = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(out + off));
      const u32x4 g = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(dy + off));
      u32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float o0, o1, g0, g1;
        unpack2<T>(o[e], o0, o1);
        unpack2<T>(g[e], g0, g1);
        float res[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k = 2 * e + h;
          float dz, y;
          in_bwd_terms(h ? o1 : o0, h ? g1 : g0, act, slope, dz, y);
          const float v = rs[k] * (dz - m1[k] - y * m2[k]);
          res[h] = cg * 8 + k < c ? v : 0.f;
        }
        r[e] = pack2<T>(res[0], res[1]);
      }
      CGAN_ST_STREAM(r, reinterpret_cast<u32x4*>(dx + off));
    }
  }
}

// ---- SPADE backward, elementwise stage ------------------------------------------------------------------------
// forward: y = act(xh * (1 + gamma) + beta), xh = (x - mean) * rstd.  Given dy and y:
//   dz = dy * act'(y);  d_gamma = dz * xh;  d_beta = dz;  d_xh = dz * (1 + gamma)
// dgb holds [d_gamma (C) | d_beta (C)] as 2C logical channels (storage round_up(2C, 8), pre-zeroed by the host entry).
template <typename T>
__global__ void spade_bwd_prepare_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ y,
                                         const uint16_t* __restrict__ x, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, const uint16_t* __restrict__ gamma,
                                         uint16_t* __restrict__ dgb, uint16_t* __restrict__ xhat,
                                         uint16_t* __restrict__ dxhat, int h, int w, int c, int cs, int cs2, int x_ups,
                                         int act, float slope, long groups) {
  const unsigned cg_total = cs / 8;
  // (32-bit index arithmetic: a map handed to the C ABI is below 2 GiB, i.e. below 2^27 groups; the four 64-bit divisions per
  // group of the first version cost more than the seven maps the kernel moves)
  for (long il = (long)blockIdx.x * blockDim.x + threadIdx.x; il < groups; il += (long)gridDim.x * blockDim.x) {
    const unsigned i = (unsigned)il;
    const unsigned pix = i / cg_total;
    const int cg = (int)(i - pix * cg_total);
    const unsigned r = pix / (unsigned)w;
    const int ox = (int)(pix - r * (unsigned)w);
    const int n = (int)(r / (unsigned)h);
    const int oy = (int)(r - (unsigned)n * (unsigned)h);
    const long xoff = x_ups ? (((long)n * (h >> 1) + (oy >> 1)) * (w >> 1) + (ox >> 1)) * cs + cg * 8 : (long)i * 8;
    const u32x4 vdy = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(dy) + i);
    const u32x4 vy = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(y) + i);
    const u32x4 vg = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(gamma) + i);
    const u32x4 vx = *reinterpret_cast<const u32x4*>(x + xoff);
    u32x4 oxh, odx, odg;
    float dbeta[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a[2], b[2], g2[2], xv[2], rxh[2], rdx[2], rdg[2];
      unpack2<T>(vdy[e], a[0], a[1]);
      unpack2<T>(vy[e], b[0], b[1]);
      unpack2<T>(vg[e], g2[0], g2[1]);
      unpack2<T>(vx[e], xv[0], xv[1]);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int ch = cg * 8 + 2 * e + hh;
        const bool live = ch < c;
        const float dz = live ? a[hh] * act_grad_from_out(b[hh], act, slope) : 0.f;
        const float xh = live ? (xv[hh] - mean[(long)n * cs + ch]) * rstd[(long)n * cs + ch] : 0.f;
        rxh[hh] = xh;
        rdx[hh] = dz * (1.f + g2[hh]);
        rdg[hh] = dz * xh;
        dbeta[2 * e + hh] = dz;
      }
      oxh[e] = pack2<T>(rxh[0], rxh[1]);
      odx[e] = pack2<T>(rdx[0], rdx[1]);
      odg[e] = pack2<T>(rdg[0], rdg[1]);
    }
    CGAN_ST_STREAM(oxh, reinterpret_cast<u32x4*>(xhat) + i);
    CGAN_ST_STREAM(odx, reinterpret_cast<u32x4*>(dxhat) + i);
    uint16_t* row = dgb + pix * cs2;
    // d_gamma: channels cg*8 .. (8-aligned: one vector unless it would spill into the d_beta range)
    if (cg * 8 + 8 <= c) {
      *reinterpret_cast<u32x4*>(row + cg * 8) = odg;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (cg * 8 + e < c) row[cg * 8 + e] = (uint16_t)((odg[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    }
    // d_beta: channels c + cg*8 ..: one 16-byte store when c is a multiple of 8, two 8-byte stores when of 4 (C = 20: the
    // range starts 40 bytes into the row), element-wise only for a group that straddles the end of the range
    if (cg * 8 + 8 <= c && (c & 3) == 0) {
      const u32x4 obt = {pack2<T>(dbeta[0], dbeta[1]), pack2<T>(dbeta[2], dbeta[3]), pack2<T>(dbeta[4], dbeta[5]),
                         pack2<T>(dbeta[6], dbeta[7])};
      if ((c & 7) == 0) {
        *reinterpret_cast<u32x4*>(row + c + cg * 8) = obt;
      } else {
        *reinterpret_cast<u32x2*>(row + c + cg * 8) = (u32x2){obt[0], obt[1]};
        *reinterpret_cast<u32x2*>(row + c + cg * 8 + 4) = (u32x2){obt[2], obt[3]};
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (cg * 8 + e < c) row[c + cg * 8 + e] = bits_of<T>(dbeta[e]);
    }
  }
}

// ---- training-mode BatchNorm2d (+ activation): statistics over (N, H, W), affine, running-stat update, backward ------
// The batch statistics come from cgan_instnorm_stats on the tensor viewed as ONE image of n*h*w pixels.
// bn_train_prepare: (mean, rstd) of the batch -> the (mean', rstd') pair cgan_norm_act_apply consumes
//   rstd' = gamma * rstd, mean' = mean - beta / rstd'   and   running <- (1 - mom) running + mom * (mean, unbiased var)
__global__ void bn_train_prepare_kernel(const float* __restrict__ mean, const float* __restrict__ rstd,
                                        const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                        float momentum, float count, float* __restrict__ running_mean,
                                        float* __restrict__ running_var, float* __restrict__ mean_out,
                                        float* __restrict__ rstd_out, long long* __restrict__ num_batches_tracked, int c,
                                        int cs) {
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch == 0 && num_batches_tracked) *num_batches_tracked += 1;     // nn.BatchNorm2d's step counter
  if (ch >= cs) return;
  float m = 0.f, r = 0.f;
  if (ch < c) {
    const float mu = mean[ch], rs = rstd[ch];
    r = rs * (gamma ? gamma[ch] : 1.f);
    m = mu - ((beta && r != 0.f) ? beta[ch] / r : 0.f);
    if (running_mean) {
      const float var_b = 1.f / (rs * rs) - eps;                               // biased batch variance
      const float var_u = count > 1.f ? var_b * count / (count - 1.f) : var_b;  // nn.BatchNorm2d tracks the unbiased one
      running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mu;
      running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * var_u;
    }
  }
  mean_out[ch] = m;
  rstd_out[ch] = r;
}

// out = act(gamma xh + beta), xh = (x - mean) rstd.  dz = dy act'(out).
// Stage 1: partial[chunk][cs][2] = (sum dz, sum dz xh) over the chunk's pixels (plain stores: fp32 atomics from 1000+
// workgroups to the same 2*cs addresses cross the XCDs and were ~2/3 of this kernel's time).
constexpr int BN_BWD_MAX_CHUNKS = 512;
// MASK: where act'(.) comes from.  0 = no activation (nothing read), 1 = from `out` (a residual was fused into the
// forward's apply, so out is not a function of x alone), 2 = recomputed from x: z = (x - m') r' exactly as the forward's
// apply kernel formed it (fold_mean / fold_rstd = the m', r' rows cgan_bn_train_prepare wrote), so `out` is not read at all -- one 16-bit map less
// per pass for every BatchNorm + ReLU that has no residual (two of a bottleneck's three)
// gneg = act'(.) on the non-positive side (0 ReLU, the slope for LeakyReLU): one compare + select per element, no
// per-element switch on the activation
template <int MASK>
__device__ __forceinline__ float bn_bwd_act_grad(float ov, float xv, float fm, float fr, float gneg) {
  if (MASK == 0) return 1.f;
  if (MASK == 1) return ov > 0.f ? 1.f : gneg;
  return (xv - fm) * fr > 0.f ? 1.f : gneg;
}

template <typename T, int MASK>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ out,
                                                            const uint16_t* __restrict__ dy, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ fmean,
                                                            const float* __restrict__ frstd, float* __restrict__ partial,
                                                            uint16_t* __restrict__ dz_out, long npix, int cs, int ppb,
                                                            float gneg) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [PL][cgb*8][2]
  const int cg_total = cs / 8;
  const int cgb = cg_total < 256 ? cg_total : 256;
  const int PL = 256 / cgb;
  const int cg0 = blockIdx.y * cgb;
  // blockIdx.z = group (npix = pixels PER GROUP): its slice of the tensors, its statistics row, its partial rows
  const size_t gofs = (size_t)blockIdx.z * npix * cs;
  x += gofs; dy += gofs;
  if (MASK == 1) out += gofs;
  if (dz_out) dz_out += gofs;
  mean += (size_t)blockIdx.z * cs; rstd += (size_t)blockIdx.z * cs;
  if (MASK == 2) { fmean += (size_t)blockIdx.z * cs; frstd += (size_t)blockIdx.z * cs; }
  partial += (size_t)blockIdx.z * gridDim.x * cs * 2;
  const long p0 = (long)blockIdx.x * ppb, p1 = min(npix, p0 + ppb);
  const int t = threadIdx.x, cgl = t % cgb, pl = t / cgb;
  const int cg = cg0 + cgl;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  if (pl < PL && cg < cg_total) {
    float mu[8], rs[8], fm[8], fr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = mean[cg * 8 + e];
      rs[e] = rstd[cg * 8 + e];
      fm[e] = MASK == 2 ? fmean[cg * 8 + e] : 0.f;
      fr[e] = MASK == 2 ? frstd[cg * 8 + e] : 0.f;
    }
    const uint16_t* xp = x + cg * 8;
    const uint16_t* op = out + cg * 8;
    const uint16_t* gp = dy + cg * 8;
#pragma unroll 2
    for (long p = p0 + pl; p < p1; p += PL) {
      const size_t off = (size_t)p * cs;
      const u32x4 vx = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(xp + off));
      u32x4 vo = {0, 0, 0, 0};
      if (MASK == 1) vo = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(op + off));
      const u32x4 vg = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(gp + off));
      u32x4 z;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xv[2], ov[2], gv[2], dzv[2];
        unpack2<T>(vx[e], xv[0], xv[1]);
        unpack2<T>(vo[e], ov[0], ov[1]);
        unpack2<T>(vg[e], gv[0], gv[1]);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const float dz = gv[hh] * bn_bwd_act_grad<MASK>(ov[hh], xv[hh], fm[2 * e + hh], fr[2 * e + hh], gneg);
          const float xh = (xv[hh] - mu[2 * e + hh]) * rs[2 * e + hh];
          s1[2 * e + hh] += dz;
          s2[2 * e + hh] += dz * xh;
          dzv[hh] = dz;
        }
        z[e] = pack2<T>(dzv[0], dzv[1]);
      }
      // ReLU with a fused residual: dz = dy or 0 is exact in 16 bits -- written here (it is the residual branch's gradient),
      // the apply pass then reads (x, dz) instead of (x, out, dy) and writes dx only
      if (MASK == 1 && dz_out) CGAN_ST_STREAM(z, reinterpret_cast<u32x4*>(dz_out + cg * 8 + off));
    }
  }
  if (pl < PL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sm[((pl * cgb + cgl) * 8 + e) * 2] = s1[e];
      sm[((pl * cgb + cgl) * 8 + e) * 2 + 1] = s2[e];
    }
  }
  __syncthreads();
  float* prow = partial + (size_t)blockIdx.x * cs * 2;
  for (int c = t; c < cgb * 8; c += 256) {
    if (cg0 * 8 + c >= cs) continue;
    float a = 0.f, b = 0.f;
    for (int l = 0; l < PL; ++l) {
      a += sm[((l * cgb) * 8 + c) * 2];
      b += sm[((l * cgb) * 8 + c) * 2 + 1];
    }
    prow[(size_t)(cg0 * 8 + c) * 2] = a;
    prow[(size_t)(cg0 * 8 + c) * 2 + 1] = b;
  }
}

// Stage 2: totals over the chunks, dgamma = sum dz xh, dbeta = sum dz (written, not accumulated), and the three per-channel
// coefficients of
//   dx = rstd gamma (dz - mean(dz) - xh mean(dz xh))  =  A dz + B x + C
// block = 8 channels x 32 chunk lanes (a partial row's 8 pairs for the block = one 64-byte segment); a thread keeps 8 row
// loads in flight and adds them in row order; chunk lanes are summed by a 3-step shuffle tree and a 4-entry LDS row, every
// group's rows are read before the one barrier -- all in a fixed order
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int chunks,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd,
                                                              const float* __restrict__ gamma, float inv_count,
                                                              float* __restrict__ coef, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int c, int cs, int groups) {
  __shared__ float red[16][4][8][2];
  const int chl = threadIdx.x & 7, kl = threadIdx.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ch = blockIdx.x * 8 + chl;          // < cs: cs is a multiple of 8
  for (int grp = 0; grp < groups; ++grp) {
    const float* rows = partial + ((size_t)grp * chunks * cs + ch) * 2;
    float a = 0.f, b = 0.f;
    for (int k0 = kl; k0 < chunks; k0 += 256) {
      float2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + 32 * j;
        v[j] = k < chunks ? *reinterpret_cast<const float2*>(rows + (size_t)k * cs * 2) : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a += v[j].x;
        b += v[j].y;
      }
    }
#pragma unroll
    for (int off = 32; off >= 8; off >>= 1) {
      a += __shfl_down(a, off, 64);
      b += __shfl_down(b, off, 64);
    }
    if (lane < 8) {
      red[grp][wave][chl][0] = a;
      red[grp][wave][chl][1] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x >= 8) return;
  float dg_sum = 0.f, db_sum = 0.f;       // dgamma / dbeta: summed over the groups (one parameter, several forward calls)
  for (int grp = 0; grp < groups; ++grp) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      s1 += red[grp][w][chl][0];
      s2 += red[grp][w][chl][1];
    }
    float A = 0.f, B = 0.f, C = 0.f;
    if (ch < c) {
      dg_sum += s2;
      db_sum += s1;
      const float rs = rstd[ch], mu = mean[ch];
      A = rs * (gamma ? gamma[ch] : 1.f);
      const float m1 = s1 * inv_count, m2 = s2 * inv_count;
      B = -A * rs * m2;
      C = A * (mu * rs * m2 - m1);
    }
    coef[ch] = A;
    coef[cs + ch] = B;
    coef[2 * cs + ch] = C;
    mean += cs; rstd += cs; coef += 3 * (size_t)cs;
  }
  if (ch < c) {
    if (dgamma) dgamma[ch] = dg_sum;
    if (dbeta) dbeta[ch] = db_sum;
  }
}

// Stage 3: dx = A dz + B x + C.  A thread keeps its 8 channels (coefficients in registers) and walks pixels: no
// per-element index division, no per-element coefficient loads.
template <typename T, int MASK>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ out,
                                                           const uint16_t* __restrict__ dy, const float* __restrict__ coef,
                                                           const float* __restrict__ fmean, const float* __restrict__ frstd,
                                                           uint16_t* __restrict__ dx, uint16_t* __restrict__ dz_out,
                                                           long npix, int cs, float gneg) {
  const int cg_total = cs / 8;
  const int tpp = cg_total < 256 ? cg_total : 256;   // threads per pixel
  const int rows = 256 / tpp;
  const int cgl = threadIdx.x % tpp, prow = threadIdx.x / tpp;
  if (prow >= rows) return;
  {   // blockIdx.y = group (npix = pixels per group): its slice, its coefficient rows
    const size_t gofs = (size_t)blockIdx.y * npix * cs;
    x += gofs; dy += gofs; dx += gofs;
    if (MASK == 1) out += gofs;
    if (dz_out) dz_out += gofs;
    coef += (size_t)blockIdx.y * 3 * cs;
    if (MASK == 2) { fmean += (size_t)blockIdx.y * cs; frstd += (size_t)blockIdx.y * cs; }
  }
  for (int cg = cgl; cg < cg_total; cg += tpp) {
    float A[8], B[8], Cc[8], fm[8], fr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      A[e] = coef[cg * 8 + e];
      B[e] = coef[cs + cg * 8 + e];
      Cc[e] = coef[2 * cs + cg * 8 + e];
      fm[e] = MASK == 2 ? fmean[cg * 8 + e] : 0.f;
      fr[e] = MASK == 2 ? frstd[cg * 8 + e] : 0.f;
    }
#pragma unroll 2
    for (long p = (long)blockIdx.x * rows + prow; p < npix; p += (long)gridDim.x * rows) {
      const size_t off = (size_t)p * cs + cg * 8;
      const u32x4 vx = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(x + off));
      u32x4 vo = {0, 0, 0, 0};
      if (MASK == 1) vo = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(out + off));
      const u32x4 vg = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(dy + off));
      u32x4 r, z;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float xv[2], ov[2], gv[2], res[2], dzv[2];
        unpack2<T>(vx[e], xv[0], xv[1]);
        unpack2<T>(vo[e], ov[0], ov[1]);
        unpack2<T>(vg[e], gv[0], gv[1]);
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int k = 2 * e + hh;
          dzv[hh] = gv[hh] * bn_bwd_act_grad<MASK>(ov[hh], xv[hh], fm[k], fr[k], gneg);
          res[hh] = A[k] * dzv[hh] + B[k] * xv[hh] + Cc[k];
        }
        r[e] = pack2<T>(res[0], res[1]);
        z[e] = pack2<T>(dzv[0], dzv[1]);
      }
      CGAN_ST_STREAM(r, reinterpret_cast<u32x4*>(dx + off));
      if (dz_out) CGAN_ST_STREAM(z, reinterpret_cast<u32x4*>(dz_out + off));   // the fused residual branch's gradient
    }
  }
}

// ---- losses ---------------------------------------------------------------------------------------------------
// block-level sum -> one atomic per block
__device__ __forceinline__ void block_atomic_add(float v, float* dst) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(dst, part[0] + part[1] + part[2] + part[3]);
}

// nn.BCEWithLogitsLoss against a constant target over the c logical channels of an NHWC tensor:
// loss += weight * sum(max(x,0) - x t + log1p(exp(-|x|)));  dx = weight * (sigmoid(x) - t) (pad channels 0)
template <typename T>
__global__ __launch_bounds__(256) void bce_logits_kernel(const uint16_t* __restrict__ x, float target, float weight,
                                                         float* __restrict__ loss, uint16_t* __restrict__ dx, int cs,
                                                         int c, long groups) {
  const int cg_total = cs / 8;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_total);
    const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a[2], gr[2];
      unpack2<T>(v[e], a[0], a[1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool live = cg * 8 + 2 * e + h < c;
        const float xv = a[h];
        const float l = fmaxf(xv, 0.f) - xv * target + log1pf(__expf(-fabsf(xv)));
        acc += live ? l : 0.f;
        const float sg = 1.f / (1.f + __expf(-xv));
        gr[h] = live ? weight * (sg - target) : 0.f;
      }
      r[e] = pack2<T>(gr[0], gr[1]);
    }
    if (dx) reinterpret_cast<u32x4*>(dx)[i] = r;
  }
  block_atomic_add(acc * weight, loss);
}

// nn.MSELoss against a constant target (GANLoss with use_lsgan=True, climategan/losses.py:50-52) over the c logical channels:
// loss += weight * sum (x - t)^2;  dx = weight * 2 (x - t) (pad channels 0)
template <typename T>
__global__ __launch_bounds__(256) void mse_const_kernel(const uint16_t* __restrict__ x, float target, float weight,
                                                        float* __restrict__ loss, uint16_t* __restrict__ dx, int cs, int c,
                                                        long groups) {
  const int cg_total = cs / 8;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_total);
    const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a[2], gr[2];
      unpack2<T>(v[e], a[0], a[1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool live = cg * 8 + 2 * e + h < c;
        const float d = a[h] - target;
        acc += live ? d * d : 0.f;
        gr[h] = live ? weight * 2.f * d : 0.f;
      }
      r[e] = pack2<T>(gr[0], gr[1]);
    }
    if (dx) reinterpret_cast<u32x4*>(dx)[i] = r;
  }
  block_atomic_add(acc * weight, loss);
}

// HingeLoss (climategan/losses.py:550-593) over the c logical channels: discriminator side (hinged)
// loss += weight * sum max(0, 1 - sgn x)   [= -mean(min(sgn x - 1, 0))],  dx = -sgn weight where 1 - sgn x > 0 (half of it
// on an exact tie, torch.min's rule); generator side (not hinged) loss += weight * sum(-sgn x), dx = -sgn weight
template <typename T>
__global__ __launch_bounds__(256) void hinge_kernel(const uint16_t* __restrict__ x, float sgn, int hinged, float weight,
                                                    float* __restrict__ loss, uint16_t* __restrict__ dx, int cs, int c,
                                                    long groups) {
  const int cg_total = cs / 8;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(i % cg_total);
    const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a[2], gr[2];
      unpack2<T>(v[e], a[0], a[1]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool live = cg * 8 + 2 * e + h < c;
        const float mval = sgn * a[h] - 1.f;                 // min(mval, 0) is what the reference averages
        float l, g;
        if (hinged) {
          l = mval < 0.f ? -mval : 0.f;
          g = mval < 0.f ? -sgn : (mval == 0.f ? -0.5f * sgn : 0.f);
        } else {
          l = -sgn * a[h];
          g = -sgn;
        }
        acc += live ? l : 0.f;
        gr[h] = live ? weight * g : 0.f;
      }
      r[e] = pack2<T>(gr[0], gr[1]);
    }
    if (dx) reinterpret_cast<u32x4*>(dx)[i] = r;
  }
  block_atomic_add(acc * weight, loss);
}

// nn.L1Loss pieces: loss += weight * sum|a - b|;  da = weight * sign(a - b)
template <typename T>
__global__ __launch_bounds__(256) void l1_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                 float weight, float* __restrict__ loss, uint16_t* __restrict__ da,
                                                 long groups) {
  float acc = 0.f;
#pragma unroll 2
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const u32x4 va = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(a) + i);
    const u32x4 vb = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(b) + i);
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a0, a1, b0, b1;
      unpack2<T>(va[e], a0, a1);
      unpack2<T>(vb[e], b0, b1);
      const float d0 = a0 - b0, d1 = a1 - b1;
      acc += fabsf(d0) + fabsf(d1);
      r[e] = pack2<T>(d0 > 0.f ? weight : (d0 < 0.f ? -weight : 0.f), d1 > 0.f ? weight : (d1 < 0.f ? -weight : 0.f));
    }
    if (da) CGAN_ST_STREAM(r, reinterpret_cast<u32x4*>(da) + i);
  }
  block_atomic_add(acc * weight, loss);
}

// ---- spectral norm: gradient w.r.t. w_bar from the gradient w.r.t. w = w_bar / sigma, sigma = u^T w_bar v ------
// Deterministic: block b leaves its partial <g, w_bar> in part[b] (plain store); every block of the apply kernel adds the
// rows in the same fixed order (SN_BWD_ROWS floats from L2) -- no memset, no atomics, run-to-run identical gradients.
constexpr int SN_BWD_ROWS = 1024;
__global__ __launch_bounds__(256) void sn_bwd_dot_kernel(const float* __restrict__ g, const float* __restrict__ w_bar,
                                                         float* __restrict__ part, long numel) {
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long)gridDim.x * blockDim.x)
    acc += g[i] * w_bar[i];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ float wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(float* __restrict__ g, const float* __restrict__ u,
                                                           const float* __restrict__ v, const float* __restrict__ sigma,
                                                           const float* __restrict__ part, int rows_n, int cols, long numel) {
  float acc = 0.f;
  for (int r = threadIdx.x; r < rows_n; r += 256) acc += part[r];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ float wsum[4];
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
  __syncthreads();
  const float dot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  const float inv = 1.f / sigma[0];
  const float k = dot * inv * inv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (long)r * cols);
    g[i] = g[i] * inv - k * u[r] * v[c];
  }
}

}  // namespace

#define DISPATCH_T(dtype, KERNEL, ...)                               \
  do {                                                               \
    if ((dtype) == CGAN_F16) hipLaunchKernelGGL(KERNEL<F16>, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<BF16>, __VA_ARGS__);              \
  } while (0)

#define DISPATCH_T2(dtype, KERNEL, M, ...)                                  \
  do {                                                                      \
    if ((dtype) == CGAN_F16) hipLaunchKernelGGL((KERNEL<F16, M>), __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<BF16, M>), __VA_ARGS__);                \
  } while (0)

extern "C" int cgan_act_bwd(const void* out, const void* dy, void* dx, int32_t dtype, int32_t act, float act_slope,
                            int64_t numel, void* stream) {
  CGAN_REQUIRE(out && dy && dx, "act_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "act_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(numel > 0 && (numel % 8) == 0, "act_bwd: numel must be a positive multiple of 8");
  CGAN_REQUIRE(act >= CGAN_ACT_NONE && act <= CGAN_ACT_SIGMOID, "act_bwd: Unsupported activation: %d", act);
  const long groups = numel / 8;
  DISPATCH_T(dtype, act_bwd_kernel, dim3(grid_for_n(groups)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)out,
             (const uint16_t*)dy, (uint16_t*)dx, act, act_slope, groups);
  CGAN_CHECK_LAUNCH("act_bwd");
  return CGAN_OK;
}

// pixel chunks of the reduction: a function of the image size and the channel count only (NOT of the batch size: a batch
// slice must give the same per-sample result as the whole batch), at most IN_BWD_MAX_CHUNKS rows per sample
constexpr int IN_BWD_MAX_CHUNKS = 1024;
static void in_bwd_plan(const CganNormStatsDesc* d, int& cgb, int& PL, int& ppb, int& chunks) {
  const int cg_total = cgan_cs(d->c) / 8;
  cgb = cg_total < 256 ? cg_total : 256;
  PL = 256 / cgb;
  ppb = ceil_div(d->hw, IN_BWD_MAX_CHUNKS);
  if (ppb < 4 * PL) ppb = 4 * PL;            // at least 4 pixels per pixel lane
  chunks = ceil_div(d->hw, ppb);
}

extern "C" size_t cgan_instnorm_act_bwd_workspace_bytes(const CganNormStatsDesc* d) {
  if (!d || d->n <= 0 || d->c <= 0 || d->hw <= 0) return 0;
  int cgb, PL, ppb, chunks;
  in_bwd_plan(d, cgb, PL, ppb, chunks);
  return (size_t)d->n * cgan_cs(d->c) * 2 * sizeof(float) * (1 + (size_t)chunks);      // sums | chunk rows
}

extern "C" int cgan_instnorm_act_bwd(const void* out, const void* dy, const float* rstd, void* dx,
                                     const CganNormStatsDesc* d, int32_t act, float act_slope, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  CGAN_REQUIRE(d && out && dy && rstd && dx && workspace, "instnorm_act_bwd: null pointer");
  CGAN_REQUIRE(d->dtype == CGAN_F16 || d->dtype == CGAN_BF16, "instnorm_act_bwd: bad dtype %d", d->dtype);
  CGAN_REQUIRE(d->n > 0 && d->hw > 0 && d->c > 0, "instnorm_act_bwd: bad shape");
  CGAN_REQUIRE(act == CGAN_ACT_NONE || act == CGAN_ACT_LRELU,
               "instnorm_act_bwd: only invertible activations (none, LeakyReLU) are supported, got %d", act);
  CGAN_REQUIRE(act != CGAN_ACT_LRELU || act_slope > 0.f, "instnorm_act_bwd: LeakyReLU slope must be > 0");
  CGAN_REQUIRE(workspace_bytes >= cgan_instnorm_act_bwd_workspace_bytes(d), "instnorm_act_bwd: workspace too small");
  const int cs = cgan_cs(d->c);
  hipStream_t s = (hipStream_t)stream;
  const int cg_total = cs / 8;
  int cgb, PL, ppb, chunks;
  in_bwd_plan(d, cgb, PL, ppb, chunks);
  float* sums = (float*)workspace;
  float* partial = sums + (size_t)d->n * cs * 2;
  const size_t smem = (size_t)PL * cgb * 8 * 2 * sizeof(float);
  DISPATCH_T(d->dtype, in_bwd_reduce_kernel, dim3(chunks, ceil_div(cg_total, cgb), d->n), dim3(256), smem, s,
             (const uint16_t*)out, (const uint16_t*)dy, partial, d->hw, cs, ppb, act, act_slope);
  hipLaunchKernelGGL(in_bwd_finalize_kernel, dim3(d->n * cs / 8), dim3(256), 0, s, (const float*)partial, sums,
                     d->n, cs, chunks);
  {
    const int tpp = cg_total < 256 ? cg_total : 256, rows = 256 / tpp;
    long blocks = (d->hw + (long)rows * 4 - 1) / ((long)rows * 4);          // ~4 pixels per thread
    const long cap = 8192 / d->n > 0 ? 8192 / d->n : 1;
    blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
    DISPATCH_T(d->dtype, in_bwd_apply_kernel, dim3((unsigned)blocks, d->n), dim3(256), 0, s, (const uint16_t*)out,
               (const uint16_t*)dy, rstd, (const float*)workspace, (uint16_t*)dx, d->hw, cs, d->c, act, act_slope);
  }
  CGAN_CHECK_LAUNCH("instnorm_act_bwd");
  return CGAN_OK;
}

extern "C" int cgan_spade_bwd_prepare(const void* dy, const void* y, const void* x, const float* mean, const float* rstd,
                                      const void* gamma, void* dgb, void* xhat, void* dxhat, const CganSpadeDesc* d,
                                      void* stream) {
  CGAN_REQUIRE(d && dy && y && x && mean && rstd && gamma && dgb && xhat && dxhat, "spade_bwd_prepare: null pointer");
  CGAN_REQUIRE(d->dtype == CGAN_F16 || d->dtype == CGAN_BF16, "spade_bwd_prepare: bad dtype %d", d->dtype);
  CGAN_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->c > 0, "spade_bwd_prepare: bad shape");
  CGAN_REQUIRE(d->act == CGAN_ACT_NONE || d->act == CGAN_ACT_LRELU, "spade_bwd_prepare: Unsupported activation: %d", d->act);
  if (d->x_upsample) CGAN_REQUIRE((d->h % 2) == 0 && (d->w % 2) == 0, "spade_bwd_prepare: x_upsample needs even h/w");
  const int cs = cgan_cs(d->c), cs2 = cgan_cs(2 * d->c);
  const long npix = (long)d->n * d->h * d->w;
  hipStream_t s = (hipStream_t)stream;
  if (cs2 != 2 * d->c) {     // pad channels past 2c exist and the kernel does not write them (C = 20 / 40: none)
    hipError_t e = hipMemsetAsync(dgb, 0, (size_t)npix * cs2 * 2, s);
    if (e != hipSuccess) {
      cgan_set_error("spade_bwd_prepare: hipMemsetAsync failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
  }
  const long groups = npix * (cs / 8);
  DISPATCH_T(d->dtype, spade_bwd_prepare_kernel, dim3(grid_for_n(groups)), dim3(256), 0, s, (const uint16_t*)dy,
             (const uint16_t*)y, (const uint16_t*)x, mean, rstd, (const uint16_t*)gamma, (uint16_t*)dgb, (uint16_t*)xhat,
             (uint16_t*)dxhat, d->h, d->w, d->c, cs, cs2, d->x_upsample, d->act, d->act_slope, groups);
  CGAN_CHECK_LAUNCH("spade_bwd_prepare");
  return CGAN_OK;
}

extern "C" int cgan_bn_train_prepare(const float* batch_mean, const float* batch_rstd, const float* gamma,
                                     const float* beta, float eps, float momentum, int64_t count, float* running_mean,
                                     float* running_var, float* mean_out, float* rstd_out,
                                     int64_t* num_batches_tracked, int32_t c, void* stream) {
  CGAN_REQUIRE(batch_mean && batch_rstd && mean_out && rstd_out, "bn_train_prepare: null pointer");
  CGAN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_prepare: running stats go together");
  CGAN_REQUIRE(c > 0 && count > 0, "bn_train_prepare: bad shape");
  const int cs = cgan_cs(c);
  hipLaunchKernelGGL(bn_train_prepare_kernel, dim3((cs + 255) / 256), dim3(256), 0, (hipStream_t)stream, batch_mean,
                     batch_rstd, gamma, beta, eps, momentum, (float)count, running_mean, running_var, mean_out, rstd_out,
                     (long long*)num_batches_tracked, c, cs);
  CGAN_CHECK_LAUNCH("bn_train_prepare");
  return CGAN_OK;
}

extern "C" size_t cgan_batchnorm_act_bwd_workspace_bytes(int32_t c) {
  // per group: 3 coefficient rows + up to BN_BWD_MAX_CHUNKS rows of (sum dz, sum dz xh) partials
  return c > 0 ? (size_t)cgan_cs(c) * (3 + 2 * (size_t)BN_BWD_MAX_CHUNKS) * sizeof(float) : 0;
}

extern "C" int cgan_batchnorm_act_bwd(const void* x, const void* out, const void* dy, const float* batch_mean,
                                      const float* batch_rstd, const float* gamma, const float* fold_mean,
                                      const float* fold_rstd, void* dx, float* dgamma, float* dbeta, void* dz_out,
                                      int32_t dtype, int64_t npix, int32_t c, int32_t act, float act_slope,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  return cgan_batchnorm_act_bwd_grouped(x, out, dy, batch_mean, batch_rstd, gamma, fold_mean, fold_rstd, dx, dgamma, dbeta,
                                        dz_out, dtype, npix, c, 1, act, act_slope, workspace, workspace_bytes, stream);
}

extern "C" int cgan_batchnorm_act_bwd_grouped(const void* x, const void* out, const void* dy, const float* batch_mean,
                                              const float* batch_rstd, const float* gamma, const float* fold_mean,
                                              const float* fold_rstd, void* dx, float* dgamma, float* dbeta,
                                              void* dz_out, int32_t dtype, int64_t npix_total, int32_t c, int32_t groups,
                                              int32_t act, float act_slope, void* workspace, size_t workspace_bytes,
                                              void* stream) {
  CGAN_REQUIRE(x && dy && batch_mean && batch_rstd && dx && workspace, "batchnorm_act_bwd: null pointer");
  CGAN_REQUIRE(out || !dz_out, "batchnorm_act_bwd: a fused residual (dz_out) needs out");
  CGAN_REQUIRE(out || act == CGAN_ACT_NONE || (fold_mean && fold_rstd),
               "batchnorm_act_bwd: without out the activation mask needs fold_mean / fold_rstd");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "batchnorm_act_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(groups >= 1 && groups <= 16 && npix_total > 0 && c > 0 && npix_total % groups == 0,
               "batchnorm_act_bwd: bad shape (the pixel count must divide into the groups)");
  CGAN_REQUIRE(act == CGAN_ACT_NONE || act == CGAN_ACT_RELU || act == CGAN_ACT_LRELU,
               "batchnorm_act_bwd: Unsupported activation: %d", act);
  CGAN_REQUIRE(workspace_bytes >= (size_t)groups * cgan_batchnorm_act_bwd_workspace_bytes(c),
               "batchnorm_act_bwd: workspace too small");
  const long npix = npix_total / groups;      // per group
  const int cs = cgan_cs(c);
  hipStream_t s = (hipStream_t)stream;
  float* coef = (float*)workspace;            // [groups][3][cs]
  float* partial = coef + (size_t)groups * 3 * cs;   // [groups][chunks][cs][2]
  const int cg_total = cs / 8;
  const int cgb = cg_total < 256 ? cg_total : 256;
  const int PL = 256 / cgb;
  const int cgblocks = ceil_div(cg_total, cgb);
  long chunks = BN_BWD_MAX_CHUNKS / cgblocks;
  if (chunks < 1) chunks = 1;
  long ppb = (npix + chunks - 1) / chunks;
  if (ppb < 8 * PL) ppb = 8 * PL;
  chunks = (npix + ppb - 1) / ppb;
  const size_t smem = (size_t)PL * cgb * 8 * 2 * sizeof(float);
  // where act' comes from: nothing to mask / from out (given: a residual may have been fused) / recomputed from x
  const int mask = act == CGAN_ACT_NONE ? 0 : (out ? 1 : 2);
  const float gneg = act == CGAN_ACT_RELU ? 0.f : act_slope;
  // ReLU + fused residual: the reduce pass writes dz (exact in 16 bits), the apply pass runs on (x, dz) without a mask
  uint16_t* dz_early = (mask == 1 && dz_out && act == CGAN_ACT_RELU) ? (uint16_t*)dz_out : nullptr;
#define BN_BWD_REDUCE(M)                                                                                              \
  DISPATCH_T2(dtype, bn_bwd_reduce_kernel, M, dim3((unsigned)chunks, cgblocks, groups), dim3(256), smem, s,           \
              (const uint16_t*)x, (const uint16_t*)out, (const uint16_t*)dy, batch_mean, batch_rstd, fold_mean,       \
              fold_rstd, partial, dz_early, (long)npix, cs, (int)ppb, gneg)
  if (mask == 0) { BN_BWD_REDUCE(0); } else if (mask == 1) { BN_BWD_REDUCE(1); } else { BN_BWD_REDUCE(2); }
#undef BN_BWD_REDUCE
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cs / 8), dim3(256), 0, s, (const float*)partial, (int)chunks,
                     batch_mean, batch_rstd, gamma, 1.f / (float)npix, coef, dgamma, dbeta, (int)c, cs, (int)groups);
  const int tpp = cg_total < 256 ? cg_total : 256;
  const int rows = 256 / tpp;
  long blocks = (npix + (long)rows * 4 - 1) / ((long)rows * 4);
  blocks = blocks < 1 ? 1 : (blocks > 4096 / groups ? 4096 / groups : blocks);
#define BN_BWD_APPLY(M)                                                                                               \
  DISPATCH_T2(dtype, bn_bwd_apply_kernel, M, dim3((unsigned)blocks, groups), dim3(256), 0, s, (const uint16_t*)x,     \
              (const uint16_t*)out, (const uint16_t*)dy, (const float*)coef, fold_mean, fold_rstd, (uint16_t*)dx,     \
              (uint16_t*)dz_out, (long)npix, cs, gneg)
  if (dz_early) {
    DISPATCH_T2(dtype, bn_bwd_apply_kernel, 0, dim3((unsigned)blocks, groups), dim3(256), 0, s, (const uint16_t*)x,
                (const uint16_t*)nullptr, (const uint16_t*)dz_early, (const float*)coef, fold_mean, fold_rstd,
                (uint16_t*)dx, (uint16_t*)nullptr, (long)npix, cs, gneg);
  } else if (mask == 0) { BN_BWD_APPLY(0); } else if (mask == 1) { BN_BWD_APPLY(1); } else { BN_BWD_APPLY(2); }
#undef BN_BWD_APPLY
  CGAN_CHECK_LAUNCH("batchnorm_act_bwd");
  return CGAN_OK;
}

extern "C" int cgan_bce_logits_nhwc(const void* x, int32_t dtype, int64_t npix, int32_t c, float target, float weight,
                                    float* loss_accum, void* dx, void* stream) {
  CGAN_REQUIRE(x && loss_accum, "bce_logits: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "bce_logits: bad dtype %d", dtype);
  CGAN_REQUIRE(npix > 0 && c > 0, "bce_logits: bad shape");
  const int cs = cgan_cs(c);
  const long groups = npix * (cs / 8);
  DISPATCH_T(dtype, bce_logits_kernel, dim3(grid_for_n(groups)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
             target, weight, loss_accum, (uint16_t*)dx, cs, c, groups);
  CGAN_CHECK_LAUNCH("bce_logits");
  return CGAN_OK;
}

extern "C" int cgan_mse_const_nhwc(const void* x, int32_t dtype, int64_t npix, int32_t c, float target, float weight,
                                   float* loss_accum, void* dx, void* stream) {
  CGAN_REQUIRE(x && loss_accum, "mse_const: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "mse_const: bad dtype %d", dtype);
  CGAN_REQUIRE(npix > 0 && c > 0, "mse_const: bad shape");
  const int cs = cgan_cs(c);
  const long groups = npix * (cs / 8);
  DISPATCH_T(dtype, mse_const_kernel, dim3(grid_for_n(groups)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
             target, weight, loss_accum, (uint16_t*)dx, cs, c, groups);
  CGAN_CHECK_LAUNCH("mse_const");
  return CGAN_OK;
}

extern "C" int cgan_hinge_nhwc(const void* x, int32_t dtype, int64_t npix, int32_t c, int32_t target_is_real,
                               int32_t for_discriminator, float weight, float* loss_accum, void* dx, void* stream) {
  CGAN_REQUIRE(x && loss_accum, "hinge: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "hinge: bad dtype %d", dtype);
  CGAN_REQUIRE(npix > 0 && c > 0, "hinge: bad shape");
  CGAN_REQUIRE(for_discriminator || target_is_real, "hinge: The generator's hinge loss must be aiming for real");
  const int cs = cgan_cs(c);
  const long groups = npix * (cs / 8);
  DISPATCH_T(dtype, hinge_kernel, dim3(grid_for_n(groups)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
             target_is_real ? 1.f : -1.f, for_discriminator ? 1 : 0, weight, loss_accum, (uint16_t*)dx, cs, c, groups);
  CGAN_CHECK_LAUNCH("hinge");
  return CGAN_OK;
}

extern "C" int cgan_l1_nhwc(const void* a, const void* b, int32_t dtype, int64_t numel, float weight, float* loss_accum,
                            void* da, void* stream) {
  CGAN_REQUIRE(a && b && loss_accum, "l1: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "l1: bad dtype %d", dtype);
  CGAN_REQUIRE(numel > 0 && (numel % 8) == 0, "l1: numel must be a positive multiple of 8");
  const long groups = numel / 8;
  // at most 2048 blocks (one full round of the chip): every block ends in an atomic on the same address, and 8192 of them
  // were most of this kernel's time on the large VGG / discriminator maps
  const int l1_grid = grid_for_n(groups) < 2048 ? grid_for_n(groups) : 2048;
  DISPATCH_T(dtype, l1_kernel, dim3(l1_grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a,
             (const uint16_t*)b, weight, loss_accum, (uint16_t*)da, groups);
  CGAN_CHECK_LAUNCH("l1");
  return CGAN_OK;
}

extern "C" int cgan_spectral_norm_bwd(float* grad_w, const float* w_bar, const float* u, const float* v,
                                      const float* sigma, int32_t rows, int32_t cols, float* workspace,
                                      void* stream) {
  CGAN_REQUIRE(grad_w && w_bar && u && v && sigma && workspace, "spectral_norm_bwd: null pointer");
  CGAN_REQUIRE(rows > 0 && cols > 0, "spectral_norm_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const long numel = (long)rows * cols;
  int g = grid_for_n(numel);
  if (g > SN_BWD_ROWS) g = SN_BWD_ROWS;
  hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(g), dim3(256), 0, s, (const float*)grad_w, w_bar, workspace, numel);
  hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(grid_for_n(numel)), dim3(256), 0, s, grad_w, u, v, sigma,
                     (const float*)workspace, g, cols, numel);
  CGAN_CHECK_LAUNCH("spectral_norm_bwd");
  return CGAN_OK;
}
