// Spectral-norm power iteration (reference climategan/norms.py:100-112), one iteration per forward:
//   t = W^T u ; v = t / (|t| + eps) ; s = W v ; u = s / (|s| + eps) ; sigma = u . s
// W = w_bar viewed [rows][cols] fp32 row-major (rows = Cout, cols = Cin*kh*kw).  HBM-bound (W is read twice),
// deterministic (no atomics: fixed-order partial sums).
#include "cgan_common.h"

namespace {

constexpr int SN_ROWS_PER_CHUNK = 32;
constexpr float SN_EPS = 1e-12f;

// t_part[rc][k] = sum_{o in chunk rc} W[o][k] * u[o]
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                      float* __restrict__ t_part, int rows, int cols) {
  int k = blockIdx.x * 256 + threadIdx.x;
  int r0 = blockIdx.y * SN_ROWS_PER_CHUNK;
  int r1 = min(rows, r0 + SN_ROWS_PER_CHUNK);
  if (k >= cols) return;
  float acc = 0.f;
  for (int o = r0; o < r1; ++o) acc += w[(size_t)o * cols + k] * u[o];
  t_part[(size_t)blockIdx.y * cols + k] = acc;
}

__device__ __forceinline__ float block_sum(float v, float* sm) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  float tot = 0.f;
  int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) tot += sm[i];
  return tot;
}

// t[k] = sum_rc t_part[rc][k]; scal[0] = |t|^2      (single block)
__global__ __launch_bounds__(1024) void sn_reduce_t_kernel(const float* __restrict__ t_part, float* __restrict__ t,
                                                           float* __restrict__ scal, int cols, int rchunks) {
  __shared__ float sm[16];
  float sq = 0.f;
  for (int k = threadIdx.x; k < cols; k += blockDim.x) {
    float acc = 0.f;
    for (int rc = 0; rc < rchunks; ++rc) acc += t_part[(size_t)rc * cols + k];
    t[k] = acc;
    sq += acc * acc;
  }
  float tot = block_sum(sq, sm);
  if (threadIdx.x == 0) scal[0] = tot;
}

// r[o] = sum_k W[o][k] * t[k]     (one wave per row)
__global__ __launch_bounds__(256) void sn_w_t_kernel(const float* __restrict__ w, const float* __restrict__ t,
                                                     float* __restrict__ r, int rows, int cols) {
  int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (o >= rows) return;
  const float* wr = w + (size_t)o * cols;
  float acc = 0.f;
  for (int k = lane; k < cols; k += 64) acc += wr[k] * t[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) r[o] = acc;
}

// v = t/(|t|+eps); s = r/(|t|+eps); u = s/(|s|+eps); sigma = u.s      (single block)
__global__ __launch_bounds__(1024) void sn_finalize_kernel(const float* __restrict__ t, const float* __restrict__ r,
                                                           const float* __restrict__ scal, float* __restrict__ u,
                                                           float* __restrict__ v, float* __restrict__ sigma, int rows,
                                                           int cols) {
  __shared__ float sm[16];
  const float inv_v = 1.f / (sqrtf(scal[0]) + SN_EPS);
  for (int k = threadIdx.x; k < cols; k += blockDim.x) v[k] = t[k] * inv_v;
  float sq = 0.f;
  for (int o = threadIdx.x; o < rows; o += blockDim.x) {
    float s = r[o] * inv_v;
    sq += s * s;
  }
  float ns2 = block_sum(sq, sm);
  float inv_u = 1.f / (sqrtf(ns2) + SN_EPS);
  for (int o = threadIdx.x; o < rows; o += blockDim.x) u[o] = r[o] * inv_v * inv_u;
  if (threadIdx.x == 0) sigma[0] = ns2 * inv_u;
}

// ---- batched variants: blockIdx.z = layer, table of CganSnItem in device memory.  Same arithmetic, same
// summation order as the single-layer kernels (results are bit-identical), 4 launches for ALL layers.
__global__ __launch_bounds__(256) void sn_wt_u_batched(const CganSnItem* __restrict__ items) {
  const CganSnItem it = items[blockIdx.z];
  int k = blockIdx.x * 256 + threadIdx.x;
  int r0 = blockIdx.y * SN_ROWS_PER_CHUNK;
  if (r0 >= it.rows || k >= it.cols) return;
  int r1 = min(it.rows, r0 + SN_ROWS_PER_CHUNK);
  float acc = 0.f;
  // (eight rows in flight per thread; the products are still added row by row in order)
#pragma unroll 8
  for (int o = r0; o < r1; ++o) acc += it.w_bar[(size_t)o * it.cols + k] * it.u[o];
  it.workspace[(size_t)blockIdx.y * it.cols + k] = acc;
}

__global__ __launch_bounds__(1024) void sn_reduce_t_batched(const CganSnItem* __restrict__ items) {
  __shared__ float sm[16];
  const CganSnItem it = items[blockIdx.x];
  const int rchunks = (it.rows + SN_ROWS_PER_CHUNK - 1) / SN_ROWS_PER_CHUNK;
  float* t = it.workspace + (size_t)rchunks * it.cols;
  float* scal = t + it.cols + it.rows;
  float sq = 0.f;
  for (int k = threadIdx.x; k < it.cols; k += blockDim.x) {
    float acc = 0.f;
#pragma unroll 8
    for (int rc = 0; rc < rchunks; ++rc) acc += it.workspace[(size_t)rc * it.cols + k];
    t[k] = acc;
    sq += acc * acc;
  }
  float tot = block_sum(sq, sm);
  if (threadIdx.x == 0) scal[0] = tot;
}

__global__ __launch_bounds__(256) void sn_w_t_batched(const CganSnItem* __restrict__ items) {
  const CganSnItem it = items[blockIdx.y];
  int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  int lane = threadIdx.x & 63;
  if (o >= it.rows) return;
  const int rchunks = (it.rows + SN_ROWS_PER_CHUNK - 1) / SN_ROWS_PER_CHUNK;
  const float* t = it.workspace + (size_t)rchunks * it.cols;
  float* r = it.workspace + (size_t)rchunks * it.cols + it.cols;
  const float* wr = it.w_bar + (size_t)o * it.cols;
  float acc = 0.f;
#pragma unroll 8
  for (int k = lane; k < it.cols; k += 64) acc += wr[k] * t[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) r[o] = acc;
}

__global__ __launch_bounds__(1024) void sn_finalize_batched(const CganSnItem* __restrict__ items) {
  __shared__ float sm[16];
  const CganSnItem it = items[blockIdx.x];
  const int rchunks = (it.rows + SN_ROWS_PER_CHUNK - 1) / SN_ROWS_PER_CHUNK;
  const float* t = it.workspace + (size_t)rchunks * it.cols;
  const float* r = t + it.cols;
  const float* scal = r + it.rows;
  const float inv_v = 1.f / (sqrtf(scal[0]) + SN_EPS);
  for (int k = threadIdx.x; k < it.cols; k += blockDim.x) it.v[k] = t[k] * inv_v;
  float sq = 0.f;
  for (int o = threadIdx.x; o < it.rows; o += blockDim.x) {
    float s = r[o] * inv_v;
    sq += s * s;
  }
  float ns2 = block_sum(sq, sm);
  float inv_u = 1.f / (sqrtf(ns2) + SN_EPS);
  for (int o = threadIdx.x; o < it.rows; o += blockDim.x) it.u[o] = r[o] * inv_v * inv_u;
  if (threadIdx.x == 0) it.sigma[0] = ns2 * inv_u;
}

}  // namespace

extern "C" int cgan_spectral_norm_power_iter_batched(const CganSnItem* items_device, int32_t count, int32_t max_rows,
                                                     int32_t max_cols, void* stream) {
  CGAN_REQUIRE(items_device && count > 0 && max_rows > 0 && max_cols > 0, "spectral_norm_batched: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int rchunks = ceil_div(max_rows, SN_ROWS_PER_CHUNK);
  hipLaunchKernelGGL(sn_wt_u_batched, dim3(ceil_div(max_cols, 256), rchunks, count), dim3(256), 0, s, items_device);
  CGAN_CHECK_LAUNCH("spectral_norm_batched(W^T u)");
  hipLaunchKernelGGL(sn_reduce_t_batched, dim3(count), dim3(1024), 0, s, items_device);
  CGAN_CHECK_LAUNCH("spectral_norm_batched(reduce t)");
  hipLaunchKernelGGL(sn_w_t_batched, dim3(ceil_div(max_rows, 4), count), dim3(256), 0, s, items_device);
  CGAN_CHECK_LAUNCH("spectral_norm_batched(W t)");
  hipLaunchKernelGGL(sn_finalize_batched, dim3(count), dim3(1024), 0, s, items_device);
  CGAN_CHECK_LAUNCH("spectral_norm_batched(finalize)");
  return CGAN_OK;
}

extern "C" size_t cgan_spectral_norm_workspace_bytes(int32_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0) return 0;
  size_t rchunks = (size_t)ceil_div(rows, SN_ROWS_PER_CHUNK);
  return (rchunks * cols + cols + rows + 4) * sizeof(float);
}

extern "C" int cgan_spectral_norm_power_iter(const float* w_bar, float* u, float* v, float* sigma, int32_t rows,
                                             int32_t cols, void* workspace, size_t workspace_bytes, void* stream) {
  CGAN_REQUIRE(w_bar && u && v && sigma && workspace, "spectral_norm: null pointer");
  CGAN_REQUIRE(rows > 0 && cols > 0, "spectral_norm: bad shape");
  size_t need = cgan_spectral_norm_workspace_bytes(rows, cols);
  if (workspace_bytes < need) {
    cgan_set_error("spectral_norm: workspace %zu B < required %zu B", workspace_bytes, need);
    return CGAN_ERR_WORKSPACE;
  }
  const int rchunks = ceil_div(rows, SN_ROWS_PER_CHUNK);
  float* t_part = (float*)workspace;
  float* t = t_part + (size_t)rchunks * cols;
  float* r = t + cols;
  float* scal = r + rows;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sn_wt_u_kernel, dim3(ceil_div(cols, 256), rchunks), dim3(256), 0, s, w_bar, u, t_part, rows, cols);
  CGAN_CHECK_LAUNCH("spectral_norm(W^T u)");
  hipLaunchKernelGGL(sn_reduce_t_kernel, dim3(1), dim3(1024), 0, s, t_part, t, scal, cols, rchunks);
  CGAN_CHECK_LAUNCH("spectral_norm(reduce t)");
  hipLaunchKernelGGL(sn_w_t_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, s, w_bar, t, r, rows, cols);
  CGAN_CHECK_LAUNCH("spectral_norm(W t)");
  hipLaunchKernelGGL(sn_finalize_kernel, dim3(1), dim3(1024), 0, s, t, r, scal, u, v, sigma, rows, cols);
  CGAN_CHECK_LAUNCH("spectral_norm(finalize)");
  return CGAN_OK;
}
