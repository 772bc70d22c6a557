// Input pipeline of apply_events: uint8 HWC photo -> aspect-preserving resize so that the smaller side
// is `to` -> centre crop to x to -> [-1, 1] fp32 NCHW, i.e. to_m1_p1(resize_and_crop(img)) of the reference
// (apply_events.py:179-195, 211-241).  The reference's resize is scikit-image 0.18.3 `resize(img, size,
// preserve_range=True, anti_aliasing=True)`: a float64 Gaussian pre-filter (scipy.ndimage.gaussian_filter, sigma =
// max(0, (scale - 1) / 2) per axis, truncate 4, mode 'mirror') followed by a bilinear warp with pixel-centre alignment
// and 'reflect' (= mirror) coordinates, then astype(uint8) truncation.  Everything here is float64 with the operation
// order of scipy's correlate1d (centre tap first, then symmetric pairs from the outermost inwards, no fused
// multiply-add) so that the truncation to uint8 sees the same values.  HBM-bound byte work; three small kernels.
#include "cgan_common.h"

namespace {

__device__ __forceinline__ int mirror(int i, int n) {   // d c b | a b c d | c b a  (no edge repeat); |overshoot| < n
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// axis-0 (rows) Gaussian of the uint8 image -> float64; one thread per element of a row-major [h][w*c] array
__global__ __launch_bounds__(256) void prep_gauss_rows_kernel(const uint8_t* __restrict__ img, double* __restrict__ out,
                                                              const double* __restrict__ wts, int radius, int h,
                                                              int row_elems) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)h * row_elems) return;
  const int r = (int)(i / row_elems), e = (int)(i - (long)r * row_elems);
  double tmp = (double)img[i] * (radius > 0 ? wts[radius] : 1.0);
  for (int jj = -radius; jj < 0; ++jj) {
    const double a = (double)img[(long)mirror(r + jj, h) * row_elems + e];
    const double b = (double)img[(long)mirror(r - jj, h) * row_elems + e];
    tmp += (a + b) * wts[radius + jj];
  }
  out[i] = tmp;
}

// axis-1 (columns) Gaussian, float64 -> float64
__global__ __launch_bounds__(256) void prep_gauss_cols_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                              const double* __restrict__ wts, int radius, int h, int w,
                                                              int c) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)h * w * c) return;
  const int ch = (int)(i % c);
  const long pix = i / c;
  const int x = (int)(pix % w);
  const long rowbase = (pix - x) * c;
  double tmp = in[i] * wts[radius];
  for (int jj = -radius; jj < 0; ++jj) {
    const double a = in[rowbase + (long)mirror(x + jj, w) * c + ch];
    const double b = in[rowbase + (long)mirror(x - jj, w) * c + ch];
    tmp += (a + b) * wts[radius + jj];
  }
  out[i] = tmp;
}

// bilinear warp of the cropped window + uint8 truncation + [-1, 1]; one thread per output pixel and channel
__global__ __launch_bounds__(256) void prep_warp_crop_kernel(const double* __restrict__ in, float* __restrict__ out, int h,
                                                             int w, int c, double fr, double fc, int top, int left,
                                                             int oh, int ow, int truncate) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= oh * ow * c) return;
  const int ch = i / (oh * ow);
  const int rem = i - ch * oh * ow;
  const int oy = rem / ow, ox = rem - oy * ow;
  // pixel-centre alignment: src = scale * (dst + 0.5) - 0.5  (skimage resize: the 0th pixel is at (0.5, 0.5))
  const double r = fr * ((double)(top + oy) + 0.5) - 0.5;
  const double cc = fc * ((double)(left + ox) + 0.5) - 0.5;
  const double fl_r = floor(r), fl_c = floor(cc);
  const int minr = (int)fl_r, minc = (int)fl_c;
  const int maxr = (int)ceil(r), maxc = (int)ceil(cc);
  const double dr = r - fl_r, dc = cc - fl_c;
  const int r0 = mirror(minr, h), r1 = mirror(maxr, h), c0 = mirror(minc, w), c1 = mirror(maxc, w);
  const double tl = in[((long)r0 * w + c0) * c + ch], tr = in[((long)r0 * w + c1) * c + ch];
  const double bl = in[((long)r1 * w + c0) * c + ch], br = in[((long)r1 * w + c1) * c + ch];
  const double topv = (1.0 - dc) * tl + dc * tr;
  const double botv = (1.0 - dc) * bl + dc * br;
  double v = (1.0 - dr) * topv + dr * botv;
  v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);       // clip=True: the filtered image never leaves [0, 255]
  // resize_and_crop: astype(uint8) truncation, then rc_img / 255.0 (float64); keep_ratio: the float image / 255
  const double u = (truncate ? (double)(int)v : v) / 255.0;
  out[i] = ((float)u - 0.5f) * 2.0f;                 // to_m1_p1: (img.astype(float32) - 0.5) * 2
}

}  // namespace

extern "C" int cgan_resize_crop_geometry(int32_t h, int32_t w, int32_t to, int32_t* rows, int32_t* cols, int32_t* top,
                                         int32_t* left) {
  CGAN_REQUIRE(h > 0 && w > 0 && to > 0 && rows && cols && top && left, "resize_crop_geometry: bad argument");
  // apply_events.py:224-238: int(to * w / h) with Python float division
  int R, C;
  if (h < w) {
    R = to;
    C = (int)((double)to * (double)w / (double)h);
  } else {
    R = (int)((double)to * (double)h / (double)w);
    C = to;
  }
  *rows = R;
  *cols = C;
  *top = (R - to) / 2;
  *left = (C - to) / 2;
  return CGAN_OK;
}

extern "C" size_t cgan_resize_crop_u8_workspace_bytes(int32_t h, int32_t w, int32_t c) {
  if (h <= 0 || w <= 0 || c <= 0) return 0;
  return (size_t)h * w * c * sizeof(double) * 2;
}

static int resize_impl(const void* img, int h, int w, int c, int R, int C, int top, int left, int oh, int ow, int truncate,
                       const double* weights_rows, int radius_rows, const double* weights_cols, int radius_cols,
                       float* out_chw, void* workspace, size_t workspace_bytes, void* stream, const char* what) {
  CGAN_REQUIRE(img && out_chw && workspace, "%s: null pointer", what);
  CGAN_REQUIRE(h > 0 && w > 0 && c > 0 && c <= 4 && R > 0 && C > 0 && oh > 0 && ow > 0, "%s: bad shape", what);
  CGAN_REQUIRE(top >= 0 && left >= 0 && top + oh <= R && left + ow <= C, "%s: output window outside the resized image", what);
  CGAN_REQUIRE(workspace_bytes >= cgan_resize_crop_u8_workspace_bytes(h, w, c), "%s: workspace too small", what);
  CGAN_REQUIRE(radius_rows >= 0 && radius_cols >= 0 && radius_rows < h && radius_cols < w,
               "%s: Gaussian radius must be smaller than the image", what);
  CGAN_REQUIRE((radius_rows == 0 || weights_rows) && (radius_cols == 0 || weights_cols), "%s: missing weights", what);
  hipStream_t s = (hipStream_t)stream;
  double* a = (double*)workspace;
  double* b = a + (size_t)h * w * c;
  const long n = (long)h * w * c;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(prep_gauss_rows_kernel, dim3(blocks), dim3(256), 0, s, (const uint8_t*)img, a, weights_rows,
                     radius_rows, h, w * c);
  const double* filtered = a;
  if (radius_cols > 0) {
    hipLaunchKernelGGL(prep_gauss_cols_kernel, dim3(blocks), dim3(256), 0, s, (const double*)a, b, weights_cols,
                       radius_cols, h, w, c);
    filtered = b;
  }
  const double fr = (double)h / (double)R, fc = (double)w / (double)C;
  hipLaunchKernelGGL(prep_warp_crop_kernel, dim3((unsigned)(((long)oh * ow * c + 255) / 256)), dim3(256), 0, s, filtered,
                     out_chw, h, w, c, fr, fc, top, left, oh, ow, truncate);
  CGAN_CHECK_LAUNCH(what);
  return CGAN_OK;
}

extern "C" int cgan_resize_crop_u8(const void* img_hwc_u8, int32_t h, int32_t w, int32_t c, int32_t to,
                                   const double* weights_rows, int32_t radius_rows, const double* weights_cols,
                                   int32_t radius_cols, float* out_chw, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  CGAN_REQUIRE(h > 0 && w > 0 && to > 0, "resize_crop_u8: bad shape");
  int32_t R, C, top, left;
  cgan_resize_crop_geometry(h, w, to, &R, &C, &top, &left);
  return resize_impl(img_hwc_u8, h, w, c, R, C, top, left, to, to, 1, weights_rows, radius_rows, weights_cols, radius_cols,
                     out_chw, workspace, workspace_bytes, stream, "resize_crop_u8");
}

extern "C" int cgan_resize_u8(const void* img_hwc_u8, int32_t h, int32_t w, int32_t c, int32_t rows, int32_t cols,
                              const double* weights_rows, int32_t radius_rows, const double* weights_cols,
                              int32_t radius_cols, float* out_chw, void* workspace, size_t workspace_bytes, void* stream) {
  return resize_impl(img_hwc_u8, h, w, c, rows, cols, 0, 0, rows, cols, 0, weights_rows, radius_rows, weights_cols,
                     radius_cols, out_chw, workspace, workspace_bytes, stream, "resize_u8");
}

// ---- validation metrics (climategan/eval_metrics.py:67-130 accuracy, mIOU): per-class counts of argmax(pred) ----------
namespace {

constexpr int METRIC_MAX_C = 64;

// counts[0][k] = #pixels predicted k, counts[1][k] = #pixels labelled k, counts[2][k] = #pixels both
template <typename T>
__global__ __launch_bounds__(256) void seg_counts_kernel(const void* __restrict__ pred, int layout, long hw, long npix,
                                                         int c, int cs, const float* __restrict__ labels,
                                                         unsigned long long* __restrict__ counts) {
  __shared__ unsigned int hist[3][METRIC_MAX_C];
  for (int i = threadIdx.x; i < 3 * METRIC_MAX_C; i += 256) (&hist[0][0])[i] = 0u;
  __syncthreads();
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    int best = 0;
    float bv;
    if (layout == 0) {                       // NHWC 16-bit, cs storage channels
      const uint16_t* row = reinterpret_cast<const uint16_t*>(pred) + p * cs;
      bv = f32_of_bits<T>(row[0]);
      for (int k = 1; k < c; ++k) {
        const float v = f32_of_bits<T>(row[k]);
        if (v > bv) { bv = v; best = k; }    // first maximum wins, as torch.argmax / np.argmax
      }
    } else {                                 // NCHW fp32
      const long n = p / hw, r = p - n * hw;
      const float* base = reinterpret_cast<const float*>(pred) + n * c * hw + r;
      bv = base[0];
      for (int k = 1; k < c; ++k) {
        const float v = base[(long)k * hw];
        if (v > bv) { bv = v; best = k; }
      }
    }
    const float lab = labels[p];
    atomicAdd(&hist[0][best], 1u);
    const int li = (int)lab;
    if (lab >= 0.f && lab < (float)c && (float)li == lab) {
      atomicAdd(&hist[1][li], 1u);
      if (li == best) atomicAdd(&hist[2][best], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * c; i += 256) {
    const unsigned int v = hist[i / c][i % c];
    if (v) atomicAdd(counts + i, (unsigned long long)v);
  }
}

}  // namespace

extern "C" int cgan_seg_counts(const void* pred, int32_t layout, int32_t dtype, int32_t n, int64_t hw, int32_t c,
                               const float* labels, unsigned long long* counts, void* stream) {
  CGAN_REQUIRE(pred && labels && counts, "seg_counts: null pointer");
  CGAN_REQUIRE(layout == 0 || layout == 1, "seg_counts: layout must be 0 (NHWC 16-bit) or 1 (NCHW fp32)");
  CGAN_REQUIRE(layout == 1 || dtype == CGAN_F16 || dtype == CGAN_BF16, "seg_counts: bad dtype %d", dtype);
  CGAN_REQUIRE(n > 0 && hw > 0 && c > 0 && c <= METRIC_MAX_C, "seg_counts: bad shape (at most %d classes)", METRIC_MAX_C);
  const long npix = (long)n * hw;
  long blocks = (npix + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipStream_t s = (hipStream_t)stream;
  if (layout == 0 && dtype == CGAN_BF16)
    hipLaunchKernelGGL(seg_counts_kernel<BF16>, dim3((unsigned)blocks), dim3(256), 0, s, pred, layout, (long)hw, npix, c,
                       cgan_cs(c), labels, counts);
  else
    hipLaunchKernelGGL(seg_counts_kernel<F16>, dim3((unsigned)blocks), dim3(256), 0, s, pred, layout, (long)hw, npix, c,
                       cgan_cs(c), labels, counts);
  CGAN_CHECK_LAUNCH("seg_counts");
  return CGAN_OK;
}
