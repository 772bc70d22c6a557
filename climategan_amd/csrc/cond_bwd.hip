// Gradient of the SPADE mask decoder's conditioning map back into the depth / segmentation predictions
// (OmniGenerator.make_m_cond with gen.m.spade.detach = false, climategan/generator.py:196-230, and the
// F.interpolate(segmap, mode="nearest") of every SPADE, climategan/norms.py:179).
#include <hip/hip_runtime.h>

#include "cgan_common.h"

namespace {

inline int grid_for_n(long total) {
  long g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

// Adjoint of the nearest resize: every source pixel sums the destination pixels that read it.  One thread per
// (source pixel, 4-channel group); the candidate destination range is bracketed from the scale and every candidate is
// checked with the forward's own index rule, so float rounding cannot lose or double-count a pixel.
template <typename T>
__global__ void __launch_bounds__(256)
    resize_nearest_bwd_kernel(const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx, int c, int h_in, int w_in,
                              int cs_in, int h_out, int w_out, int cs_out, float sy, float sx, long total) {
  const int g_in = cs_in / 4;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % g_in);
    const long pix = idx / g_in;
    const int ix = (int)(pix % w_in);
    const long r = pix / w_in;
    const int iy = (int)(r % h_in);
    const long n = r / h_in;
    int oy0 = (int)floorf((float)iy / sy) - 1, oy1 = (int)ceilf((float)(iy + 1) / sy) + 1;
    int ox0 = (int)floorf((float)ix / sx) - 1, ox1 = (int)ceilf((float)(ix + 1) / sx) + 1;
    if (iy == h_in - 1) oy1 = h_out - 1;                     // the forward clamps to the last source row / column
    if (ix == w_in - 1) ox1 = w_out - 1;
    oy0 = oy0 < 0 ? 0 : oy0;
    ox0 = ox0 < 0 ? 0 : ox0;
    oy1 = oy1 > h_out - 1 ? h_out - 1 : oy1;
    ox1 = ox1 > w_out - 1 ? w_out - 1 : ox1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oy = oy0; oy <= oy1; ++oy) {
      if (nearest_src(oy, sy, h_in) != iy) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        if (nearest_src(ox, sx, w_in) != ix) continue;
        const uint16_t* src = dy + ((n * h_out + oy) * (long)w_out + ox) * cs_out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ch = g * 4 + e;
          if (ch < c) acc[e] += f32_of_bits<T>(src[ch]);
        }
      }
    }
    u32x2 pk;
    pk[0] = pack2<T>(g * 4 + 0 < c ? acc[0] : 0.f, g * 4 + 1 < c ? acc[1] : 0.f);
    pk[1] = pack2<T>(g * 4 + 2 < c ? acc[2] : 0.f, g * 4 + 3 < c ? acc[3] : 0.f);
    *reinterpret_cast<u32x2*>(dx + pix * cs_in + g * 4) = pk;
  }
}

// block-wide reductions over 256 threads
__device__ __forceinline__ float block_sum(float v, float* sh) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// (value, first index) extremum: sign = +1 -> minimum, -1 -> maximum
__device__ __forceinline__ void block_argext(float& v, int& i, float sign, float* shv, int* shi) {
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(v, o);
    const int oi = __shfl_xor(i, o);
    if (sign * ov < sign * v || (ov == v && oi < i)) {
      v = ov;
      i = oi;
    }
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    shv[threadIdx.x >> 6] = v;
    shi[threadIdx.x >> 6] = i;
  }
  __syncthreads();
  v = shv[0];
  i = shi[0];
  for (int k = 1; k < 4; ++k)
    if (sign * shv[k] < sign * v || (shv[k] == v && shi[k] < i)) {
      v = shv[k];
      i = shi[k];
    }
}

// normalize(d) = (d - min_n) / max_n(d - min_n) per sample (tutils.py:567-576): with g the gradient of the normalised
// map, nrm its value, G = sum g, S = sum g * nrm:   dd_i = (g_i - [i = argmax] S - [i = argmin] (G - S)) / (max - min)
// (torch's min(1) / max(1) send their gradient to ONE index; ties, which only 16-bit storage produces, go to the first).
// One workgroup per sample.
template <typename T>
__global__ void __launch_bounds__(256)
    m_cond_bwd_depth_kernel(const uint16_t* __restrict__ dcond, const uint16_t* __restrict__ d, uint16_t* __restrict__ dd,
                            int hw, int ccs) {
  __shared__ float shv[4];
  __shared__ int shi[4];
  const long base = (long)blockIdx.x * hw;
  float mn = __builtin_inff(), mx = -__builtin_inff();
  int imn = 0x7fffffff, imx = 0x7fffffff;
  for (int p = threadIdx.x; p < hw; p += 256) {
    const float v = f32_of_bits<T>(d[(base + p) * 8]);
    if (v < mn) { mn = v; imn = p; }
    if (v > mx) { mx = v; imx = p; }
  }
  block_argext(mn, imn, 1.f, shv, shi);
  block_argext(mx, imx, -1.f, shv, shi);
  const float M = mx - mn;
  float G = 0.f, S = 0.f;
  for (int p = threadIdx.x; p < hw; p += 256) {
    const float g = f32_of_bits<T>(dcond[(base + p) * ccs]);
    const float nrm = __fdiv_rn(f32_of_bits<T>(d[(base + p) * 8]) - mn, M);
    G += g;
    S += g * nrm;
  }
  G = block_sum(G, shv);
  S = block_sum(S, shv);
  for (int p = threadIdx.x; p < hw; p += 256) {
    float g = f32_of_bits<T>(dcond[(base + p) * ccs]);
    if (p == imx) g -= S;
    if (p == imn) g -= G - S;
    u32x4 o = {0u, 0u, 0u, 0u};
    o[0] = pack2<T>(__fdiv_rn(g, M), 0.f);
    *reinterpret_cast<u32x4*>(dd + (base + p) * 8) = o;
  }
}

// softmax(s, dim=1): ds_k = p_k (g_k - sum_j g_j p_j), one thread per pixel
template <typename T>
__global__ void __launch_bounds__(256)
    m_cond_bwd_seg_kernel(const uint16_t* __restrict__ dcond, const uint16_t* __restrict__ seg, uint16_t* __restrict__ ds,
                          int sc, int scs, int ccs, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const uint16_t* sp = seg + i * scs;
    const uint16_t* gp = dcond + i * ccs + 1;
    uint16_t* op = ds + i * scs;
    float mx = -__builtin_inff();
    for (int c = 0; c < sc; ++c) mx = fmaxf(mx, f32_of_bits<T>(sp[c]));
    float sum = 0.f, dot = 0.f;
    for (int c = 0; c < sc; ++c) {
      const float e = __expf(f32_of_bits<T>(sp[c]) - mx);
      sum += e;
      dot += e * f32_of_bits<T>(gp[c]);
    }
    const float inv = 1.f / sum;
    dot *= inv;
    for (int c = 0; c < sc; ++c) {
      const float pk = __expf(f32_of_bits<T>(sp[c]) - mx) * inv;
      op[c] = bits_of<T>(pk * (f32_of_bits<T>(gp[c]) - dot));
    }
    for (int c = sc; c < scs; ++c) op[c] = 0;
  }
}

}  // namespace

#define DISPATCH_T(dtype, KERNEL, ...)                                     \
  do {                                                                     \
    if ((dtype) == CGAN_F16) hipLaunchKernelGGL(KERNEL<F16>, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<BF16>, __VA_ARGS__);                    \
  } while (0)

extern "C" int cgan_resize_nearest_bwd_nhwc(const void* dy, void* dx, int32_t dtype, int32_t n, int32_t c, int32_t h_in,
                                            int32_t w_in, int32_t cs_in, int32_t h_out, int32_t w_out, int32_t cs_out,
                                            void* stream) {
  CGAN_REQUIRE(dy && dx, "resize_nearest_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "resize_nearest_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && c > 0 && h_in > 0 && w_in > 0 && h_out > 0 && w_out > 0, "resize_nearest_bwd: bad shape");
  CGAN_REQUIRE(cs_in >= c && cs_out >= c && (cs_in % 4) == 0, "resize_nearest_bwd: bad channel storage");
  const long total = (long)n * h_in * w_in * (cs_in / 4);
  const float sy = (float)h_in / (float)h_out, sx = (float)w_in / (float)w_out;
  DISPATCH_T(dtype, resize_nearest_bwd_kernel, dim3(grid_for_n(total)), dim3(256), 0, (hipStream_t)stream,
             (const uint16_t*)dy, (uint16_t*)dx, c, h_in, w_in, cs_in, h_out, w_out, cs_out, sy, sx, total);
  CGAN_CHECK_LAUNCH("resize_nearest_bwd");
  return CGAN_OK;
}

extern "C" int cgan_make_m_cond_bwd_nhwc(const void* dcond_nhwc, const void* depth_nhwc, const void* seg_nhwc,
                                         void* ddepth_nhwc, void* dseg_nhwc, int32_t dtype, int32_t n, int32_t h,
                                         int32_t w, int32_t seg_c, int32_t with_x, void* stream) {
  CGAN_REQUIRE(dcond_nhwc && depth_nhwc && seg_nhwc && ddepth_nhwc && dseg_nhwc, "make_m_cond_bwd: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "make_m_cond_bwd: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && h > 0 && w > 0 && seg_c > 0, "make_m_cond_bwd: bad shape");
  CGAN_REQUIRE((long)h * w < (1l << 30), "make_m_cond_bwd: map too large");
  const int ccs = ((1 + seg_c + (with_x ? 3 : 0)) + 3) / 4 * 4;
  const int scs = cgan_cs(seg_c);
  hipStream_t s = (hipStream_t)stream;
  DISPATCH_T(dtype, m_cond_bwd_depth_kernel, dim3(n), dim3(256), 0, s, (const uint16_t*)dcond_nhwc,
             (const uint16_t*)depth_nhwc, (uint16_t*)ddepth_nhwc, h * w, ccs);
  const long total = (long)n * h * w;
  DISPATCH_T(dtype, m_cond_bwd_seg_kernel, dim3(grid_for_n(total)), dim3(256), 0, s, (const uint16_t*)dcond_nhwc,
             (const uint16_t*)seg_nhwc, (uint16_t*)dseg_nhwc, seg_c, scs, ccs, total);
  CGAN_CHECK_LAUNCH("make_m_cond_bwd");
  return CGAN_OK;
}
