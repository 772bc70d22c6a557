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
// vgg_in = vgg_preprocess(p * m)              (tutils.py:416-427: BGR, [0,255], mean-subtracted)  3 ch as hi | lo -> cs 8
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
      // values of magnitude 100-150 in 16 bit would carry +-0.5 (bf16) / +-0.06 (fp16) of rounding into the first VGG
      // conv: store v = hi + lo (hi = v rounded to 16 bit, lo = the remainder, also 16 bit) in channels 0-2 / 3-5; the
      // first conv is linear, so running it on 6 input channels with its weights repeated gives conv(w, hi + lo)
      const float b = (pc[2] * mv + 1.f) * 255.f * 0.5f - 103.939f;
      const float g = (pc[1] * mv + 1.f) * 255.f * 0.5f - 116.779f;
      const float r = (pc[0] * mv + 1.f) * 255.f * 0.5f - 123.680f;
      const float bh = f32_of_bits<T>(bits_of<T>(b)), gh = f32_of_bits<T>(bits_of<T>(g)), rh = f32_of_bits<T>(bits_of<T>(r));
      u32x4 o;
      o[0] = pack2<T>(bh, gh);
      o[1] = pack2<T>(rh, b - bh);
      o[2] = pack2<T>(g - gh, r - rh);
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

// The Painter's optional image-space terms (get_painter_loss, climategan/trainer.py:1289-1315; lambdas 0 in defaults.yaml)
// on the pasted image p = x (1 - m) + fake m, value and gradient w.r.t. ``fake`` in one pass:
//   TVLoss(p m)               losses.py:142-169   wh sum (q[y+1] - q[y])^2 + ww sum (q[x+1] - q[x])^2,  q = p m
//   ContextLoss(p, x, m)      losses.py:281-287   w_ctx sum |(p - x)(1 - m)|
//   ReconstructionLoss(p,x,m) losses.py:290-296   w_rec sum |(p - x) m|
// (the weights carry lambda and the means' 1 / count).  Gradient in gather form (every pixel adds its own five-point
// stencil: no atomics): d/dfake = m dL/dp, dL/dp = m dTV/dq + sign((p - x)(1 - m)) (1 - m) w_ctx + sign((p - x) m) m w_rec.
// loss[0..2] += the three values (tv, context, reconstruction).
template <typename T>
__global__ __launch_bounds__(256) void painter_aux_kernel(const uint16_t* __restrict__ fake, const float* __restrict__ x,
                                                          const float* __restrict__ m, float wh, float ww, float w_ctx,
                                                          float w_rec, float* __restrict__ loss, uint16_t* __restrict__ dfake,
                                                          int h, int w, long total) {
  const long hw = (long)h * w;
  float a_tv = 0.f, a_ctx = 0.f, a_rec = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / hw, p = i - n * hw;
    const int py = (int)(p / w), px = (int)(p - (long)py * w);
    const float mv = m[i];
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float* xc = x + (n * 3 + c) * hw;
      auto q_at = [&](long j) {                       // q = (x (1 - m) + fake m) m at pixel j of this image
        const float mj = m[n * hw + j];
        return (xc[j] * (1.f - mj) + f32_of_bits<T>(fake[(n * hw + j) * 8 + c]) * mj) * mj;
      };
      const float xv = xc[p], fv = f32_of_bits<T>(fake[i * 8 + c]);
      const float pv = xv * (1.f - mv) + fv * mv;
      const float q0 = pv * mv;
      float dq = 0.f;
      if (wh != 0.f || ww != 0.f) {
        if (py + 1 < h) { const float d = q_at(p + w) - q0; a_tv += wh * d * d; dq -= 2.f * wh * d; }
        if (py > 0) dq += 2.f * wh * (q0 - q_at(p - w));
        if (px + 1 < w) { const float d = q_at(p + 1) - q0; a_tv += ww * d * d; dq -= 2.f * ww * d; }
        if (px > 0) dq += 2.f * ww * (q0 - q_at(p - 1));
      }
      const float tc = (pv - xv) * (1.f - mv), tr = (pv - xv) * mv;
      a_ctx += w_ctx * fabsf(tc);
      a_rec += w_rec * fabsf(tr);
      const float sc = tc > 0.f ? 1.f : (tc < 0.f ? -1.f : 0.f), sr = tr > 0.f ? 1.f : (tr < 0.f ? -1.f : 0.f);
      g[c] = mv * (mv * dq + sc * (1.f - mv) * w_ctx + sr * mv * w_rec);
    }
    if (dfake) {
      u32x4 o;
      o[0] = pack2<T>(g[0], g[1]);
      o[1] = pack2<T>(g[2], 0.f);
      o[2] = 0u;
      o[3] = 0u;
      reinterpret_cast<u32x4*>(dfake)[i] = o;
    }
  }
  // block sums -> one atomic per block and term (logged loss scalars only: the gradient above has no atomics)
  __shared__ float part[3][4];
  float v[3] = {a_tv, a_ctx, a_rec};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o);
    if ((threadIdx.x & 63) == 0) part[k][threadIdx.x >> 6] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd(loss + threadIdx.x, part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3]);
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

// RELU: x is itself the output of a ReLU whose derivative is taken here as well (cgan_maxpool2x2_relu_bwd_nhwc): the routed
// gradient survives only where the window's maximum is positive -- what a separate activation-backward pass over dx would leave
template <typename T, bool RELU>
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
      arg[e] = (RELU && !(best > 0.f)) ? -1 : a;
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


// ---- backward passes the Masker's graph needs ----------------------------------------------------------------------
// bilinear resize backward (F.interpolate(mode="bilinear"), both align_corners conventions), GATHER form: every input
// pixel sums the output pixels that read it.  The forward maps an output coordinate o to f(o) (monotonic), taps floor(f)
// and floor(f) + 1 (clamped) with weights 1 - frac, frac; the outputs that touch input coordinate i form a short
// contiguous range around i / scale, which is re-evaluated with the forward's own index rule (so clamped borders and the
// y0 == y1 case get exactly the forward's weights).  No atomics, no fp32 accumulation buffer, one pass.  (The first
// version scattered with fp32 atomics: 0.5 ms for the 256-channel 160^2 -> 82^2 map of the DeepLab decoder.)
struct AxisTap {
  int lo, hi;   // candidate output range [lo, hi]
};
__device__ __forceinline__ AxisTap axis_candidates(int i, float s, int align, int n_out) {
  AxisTap t;
  if (s <= 0.f) {   // n_out == 1 with align_corners: everything reads coordinate 0
    t.lo = 0; t.hi = i == 0 ? n_out - 1 : -1;
    return t;
  }
  const float inv = 1.f / s;
  float a = align ? (i - 1) * inv : (i - 0.5f) * inv - 0.5f;
  float b = align ? (i + 1) * inv : (i + 1.5f) * inv - 0.5f;
  t.lo = max(0, (int)floorf(a) - 1);
  t.hi = min(n_out - 1, (int)ceilf(b) + 1);
  return t;
}
// weight with which output coordinate o reads input coordinate i (0 if it does not)
__device__ __forceinline__ float axis_weight(int o, int i, float s, int align, int n_in) {
  const float f = align ? o * s : fmaxf((o + 0.5f) * s - 0.5f, 0.f);
  int i0 = (int)f;
  i0 = i0 < n_in - 1 ? i0 : n_in - 1;
  const int i1 = i0 < n_in - 1 ? i0 + 1 : i0;
  const float l = f - i0;
  return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}

template <typename T>
__global__ void bilinear_bwd_gather_kernel(const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, int h_in, int w_in,
                                           int h_out, int w_out, int cs, float sy, float sx, int align, long total) {
  const int cg_total = cs / 8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long pix = idx / cg_total;
    const int ix = (int)(pix % w_in);
    const long r = pix / w_in;
    const int iy = (int)(r % h_in);
    const long n = r / h_in;
    const AxisTap ty = axis_candidates(iy, sy, align, h_out), tx = axis_candidates(ix, sx, align, w_out);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const uint16_t* base = dy + n * (long)h_out * w_out * cs + cg * 8;
    for (int oy = ty.lo; oy <= ty.hi; ++oy) {
      const float wy = axis_weight(oy, iy, sy, align, h_in);
      if (wy == 0.f) continue;
      for (int ox = tx.lo; ox <= tx.hi; ++ox) {
        const float wgt = wy * axis_weight(ox, ix, sx, align, w_in);
        if (wgt == 0.f) continue;
        const u32x4 g = *reinterpret_cast<const u32x4*>(base + ((long)oy * w_out + ox) * cs);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float g0, g1;
          unpack2<T>(g[e], g0, g1);
          acc[2 * e] += wgt * g0;
          acc[2 * e + 1] += wgt * g1;
        }
      }
    }
    u32x4 o = (u32x4){pack2<T>(acc[0], acc[1]), pack2<T>(acc[2], acc[3]), pack2<T>(acc[4], acc[5]), pack2<T>(acc[6], acc[7])};
    *reinterpret_cast<u32x4*>(dx + pix * cs + cg * 8) = o;
  }
}

// nn.MaxPool2d(3, 2, 1) backward (ResNet stem).  A thread owns a 2 x 2 block of input pixels (rows 2a, 2a+1; columns 2b,
// 2b+1) x 8 channels: the only windows that contain them are (a..a+1) x (b..b+1), whose inputs are the 5 x 5 patch around
// the block.  It finds each window's FIRST maximum in row-major order (torch's tie rule: a later element wins only if
// strictly greater) and gives dy to that position: 25 + 4 loads per 4 pixels.  (The first version worked per input pixel
// and re-scanned every window that contains it: 32 neighbour loads per pixel, 464 us for the 8 x 320 x 320 x 64 stem map,
// 9x its HBM time.)
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                               uint16_t* __restrict__ dx, int h_in, int w_in, int h_out,
                                                               int w_out, int cs, long total) {
  const int cg_total = cs / 8;
  const int hb = (h_in + 1) >> 1, wb = (w_in + 1) >> 1;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int cg = (int)(idx % cg_total);
    const long blk = idx / cg_total;
    const int b = (int)(blk % wb);
    const long r0 = blk / wb;
    const int a = (int)(r0 % hb);
    const long n = r0 / hb;
    const uint16_t* xb = x + n * (long)h_in * w_in * cs + cg * 8;
    u32x4 patch[5][5];
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int yy = 2 * a - 1 + r, xx = 2 * b - 1 + c;
        patch[r][c] = (u32x4){0u, 0u, 0u, 0u};
        if (yy >= 0 && yy < h_in && xx >= 0 && xx < w_in)
          patch[r][c] = *reinterpret_cast<const u32x4*>(xb + ((long)yy * w_in + xx) * cs);
      }
    u32x4 g[2][2];
#pragma unroll
    for (int wy = 0; wy < 2; ++wy)
#pragma unroll
      for (int wx = 0; wx < 2; ++wx) {
        g[wy][wx] = (u32x4){0u, 0u, 0u, 0u};
        if (a + wy < h_out && b + wx < w_out)
          g[wy][wx] = *reinterpret_cast<const u32x4*>(dy + ((n * h_out + a + wy) * (long)w_out + b + wx) * cs + cg * 8);
      }
    u32x4 o[2][2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc[2][2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};
#pragma unroll
      for (int wy = 0; wy < 2; ++wy)
#pragma unroll
        for (int wx = 0; wx < 2; ++wx) {
          if (!(a + wy < h_out && b + wx < w_out)) continue;
          float best[2] = {0.f, 0.f};
          int at[2] = {-1, -1};                       // local position r * 5 + c of the first maximum
#pragma unroll
          for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const int r = 2 * wy + ky, c = 2 * wx + kx;
              const int yy = 2 * a - 1 + r, xx = 2 * b - 1 + c;
              if (yy < 0 || yy >= h_in || xx < 0 || xx >= w_in) continue;
              float v[2];
              unpack2<T>(patch[r][c][e], v[0], v[1]);
#pragma unroll
              for (int h = 0; h < 2; ++h)
                if (at[h] < 0 || v[h] > best[h]) { best[h] = v[h]; at[h] = r * 5 + c; }
            }
          float d[2];
          unpack2<T>(g[wy][wx][e], d[0], d[1]);
#pragma unroll
          for (int py = 0; py < 2; ++py)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
              for (int h = 0; h < 2; ++h)
                if (at[h] == (1 + py) * 5 + 1 + px) acc[py][px][h] += d[h];
        }
#pragma unroll
      for (int py = 0; py < 2; ++py)
#pragma unroll
        for (int px = 0; px < 2; ++px) o[py][px][e] = pack2<T>(acc[py][px][0], acc[py][px][1]);
    }
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px) {
        const int iy = 2 * a + py, ix = 2 * b + px;
        if (iy < h_in && ix < w_in)
          *reinterpret_cast<u32x4*>(dx + ((n * h_in + iy) * (long)w_in + ix) * cs + cg * 8) = o[py][px];
      }
  }
}

// y = act(a + b) (the residual add + ReLU of a ResNet bottleneck in training mode, resnet101_v3.py:46-48)
template <typename T>
__global__ void add_act_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, uint16_t* __restrict__ y,
                               int act, float slope, long groups) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < groups; i += (long)gridDim.x * blockDim.x) {
    const u32x4 va = reinterpret_cast<const u32x4*>(a)[i], vb = reinterpret_cast<const u32x4*>(b)[i];
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a0, a1, b0, b1;
      unpack2<T>(va[e], a0, a1);
      unpack2<T>(vb[e], b0, b1);
      o[e] = pack2<T>(act_apply(a0 + b0, act, slope), act_apply(a1 + b1, act, slope));
    }
    reinterpret_cast<u32x4*>(y)[i] = o;
  }
}

// dst[pix][0..c) = src[pix][c_off_src .. c_off_src + c)   (backward of torch.cat along channels: a channel slice)
__global__ void slice_channels_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, int cs_src, int cs_dst,
                                      int c_off_src, int c, long total) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % cs_dst);
    const long pix = idx / cs_dst;
    dst[idx] = k < c ? src[pix * cs_src + c_off_src + k] : 0;
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

extern "C" int cgan_painter_aux_losses(const void* fake_nhwc, const float* x_nchw, const float* m_nchw, int32_t dtype, int32_t n,
                                       int32_t h, int32_t w, float w_tv_h, float w_tv_w, float w_context,
                                       float w_reconstruction, float* loss3, void* d_fake, void* stream) {
  CGAN_REQUIRE(fake_nhwc && x_nchw && m_nchw && loss3, "painter_aux_losses: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "painter_aux_losses: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && h > 1 && w > 1, "painter_aux_losses: bad shape");
  const long total = (long)n * h * w;
  DISPATCH_PT(dtype, painter_aux_kernel, dim3(grid_pt(total)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)fake_nhwc,
              x_nchw, m_nchw, w_tv_h, w_tv_w, w_context, w_reconstruction, loss3, (uint16_t*)d_fake, h, w, total);
  CGAN_CHECK_LAUNCH("painter_aux_losses");
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

static int maxpool2x2_bwd_impl(const void* x, const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                               int32_t w_in, void* stream, bool relu) {
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
#define MP_BWD(TT, RR)                                                                                                     \
  hipLaunchKernelGGL((maxpool2x2_bwd_kernel<TT, RR>), dim3(grid_pt(total)), dim3(256), 0, s, (const uint16_t*)x,         \
                     (const uint16_t*)dy, (uint16_t*)dx, h_in, w_in, h_out, w_out, cs, total)
  if (dtype == CGAN_F16) { if (relu) MP_BWD(F16, true); else MP_BWD(F16, false); }
  else { if (relu) MP_BWD(BF16, true); else MP_BWD(BF16, false); }
#undef MP_BWD
  CGAN_CHECK_LAUNCH("maxpool2x2_bwd");
  return CGAN_OK;
}

extern "C" int cgan_maxpool2x2_bwd_nhwc(const void* x, const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c,
                                        int32_t h_in, int32_t w_in, void* stream) {
  return maxpool2x2_bwd_impl(x, dy, dx, dtype, n, c, h_in, w_in, stream, false);
}

extern "C" int cgan_maxpool2x2_relu_bwd_nhwc(const void* x, const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c,
                                             int32_t h_in, int32_t w_in, void* stream) {
  return maxpool2x2_bwd_impl(x, dy, dx, dtype, n, c, h_in, w_in, stream, true);
}

extern "C" size_t cgan_resize_bilinear_bwd_workspace_bytes(int32_t n, int32_t c, int32_t h_in, int32_t w_in) {
  (void)n; (void)c; (void)h_in; (void)w_in;
  return 0;   // the gather form needs none (kept in the ABI: callers size and pass a workspace)
}

extern "C" int cgan_resize_bilinear_bwd_nhwc(const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                             int32_t w_in, int32_t h_out, int32_t w_out, int32_t align_corners,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  (void)workspace; (void)workspace_bytes;
  CGAN_REQUIRE(dy && dx, "resize_bilinear_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "resize_bilinear_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "resize_bilinear_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int cs = cgan_cs(c);
  float sy, sx;
  if (align_corners) {
    sy = h_out > 1 ? (float)(h_in - 1) / (float)(h_out - 1) : 0.f;
    sx = w_out > 1 ? (float)(w_in - 1) / (float)(w_out - 1) : 0.f;
  } else {
    sy = (float)h_in / (float)h_out;
    sx = (float)w_in / (float)w_out;
  }
  const long total = (long)n * h_in * w_in * (cs / 8);
  DISPATCH_PT(dtype, bilinear_bwd_gather_kernel, dim3(grid_pt(total)), dim3(256), 0, s, (const uint16_t*)dy,
              (uint16_t*)dx, h_in, w_in, h_out, w_out, cs, sy, sx, align_corners, total);
  CGAN_CHECK_LAUNCH("resize_bilinear_bwd");
  return CGAN_OK;
}

extern "C" int cgan_maxpool3x3s2_bwd_nhwc(const void* x, const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c,
                                          int32_t h_in, int32_t w_in, void* stream) {
  CGAN_REQUIRE(x && dy && dx, "maxpool3x3s2_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "maxpool3x3s2_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0, "maxpool3x3s2_bwd: bad shape");
  const int h_out = (h_in + 2 - 3) / 2 + 1, w_out = (w_in + 2 - 3) / 2 + 1, cs = cgan_cs(c);
  const long total = (long)n * ((h_in + 1) / 2) * ((w_in + 1) / 2) * (cs / 8);     // 2 x 2 input blocks x channel groups
  DISPATCH_PT(dtype, maxpool3x3s2_bwd_kernel, dim3(grid_pt(total)), dim3(256), 0, (hipStream_t)stream,
              (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, h_in, w_in, h_out, w_out, cs, total);
  CGAN_CHECK_LAUNCH("maxpool3x3s2_bwd");
  return CGAN_OK;
}

extern "C" int cgan_add_act_nhwc(const void* a, const void* b, void* y, int32_t dtype, int32_t act, float act_slope,
                                 int64_t numel, void* stream) {
  CGAN_REQUIRE(a && b && y, "add_act: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "add_act: bad dtype %d", dtype);
  CGAN_REQUIRE(numel > 0 && (numel % 8) == 0, "add_act: numel must be a positive multiple of 8");
  CGAN_REQUIRE(act >= CGAN_ACT_NONE && act <= CGAN_ACT_SIGMOID, "add_act: Unsupported activation: %d", act);
  const long groups = numel / 8;
  DISPATCH_PT(dtype, add_act_kernel, dim3(grid_pt(groups)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a,
              (const uint16_t*)b, (uint16_t*)y, act, act_slope, groups);
  CGAN_CHECK_LAUNCH("add_act");
  return CGAN_OK;
}

extern "C" int cgan_slice_channels_nhwc(const void* src, void* dst, int64_t npix, int32_t c, int32_t cs_src,
                                        int32_t c_off_src, void* stream) {
  CGAN_REQUIRE(src && dst, "slice_channels: null pointer");
  CGAN_REQUIRE(npix > 0 && c > 0 && c_off_src >= 0 && c_off_src + c <= cs_src, "slice_channels: bad channel range");
  const int cs_dst = cgan_cs(c);
  const long total = npix * cs_dst;
  hipLaunchKernelGGL(slice_channels_kernel, dim3(grid_pt(total)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)src, (uint16_t*)dst, cs_src, cs_dst, c_off_src, c, total);
  CGAN_CHECK_LAUNCH("slice_channels");
  return CGAN_OK;
}
