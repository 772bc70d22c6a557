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
