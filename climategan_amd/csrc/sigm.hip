// SIGMLoss (climategan/losses.py:237-278; the MiDaS scale-and-shift-invariant depth loss with a 4-scale Sobel gradient
// matching term) and its gradient, on the depth decoder's 1-channel NHWC map vs an fp32 target.
//
//   t = median(all elements), s = mean |v - t|  for prediction and target;  R = (p - t_p)/s_p - (q - t_q)/s_q
//   loss = 0.5/np sum|R| + gm/np * B * sum_k sum(|sobel_x * R_k| + |sobel_y * R_k|),  R_k = R[::2^k, ::2^k], np = h*w
// (the factor B: the reference expands the 3x3 filters to B OUTPUT channels, losses.py:262-268, so every image's
// response is counted B times).  Median = exact order statistic by a 4-pass radix select on order-preserving keys.
// Gradient through the normalisation: with G = dL/dp', N = B h w, m = the median element,
//   dL/dp_j = G_j/s - [j==m] sum(G)/s - (sum_i G_i (p_i - t))/s^2 * (sign(p_j - t) - [j==m] sum_i sign(p_i - t))/N
#include "cgan_common.h"

namespace {

__device__ __forceinline__ unsigned fkey(float f) {           // order-preserving float -> uint
  unsigned b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

struct SelState {     // one per tensor
  unsigned prefix;    // key bits decided so far
  unsigned mask;      // which bits are decided
  long long k;        // rank still to resolve inside the prefix bucket
  unsigned hist[256];
};

template <typename T, bool PRED>
__device__ __forceinline__ float load_v(const void* p, long i) {
  if (PRED) return f32_of_bits<T>(((const uint16_t*)p)[i * 8]);   // channel 0 of an NHWC map stored with 8 channels
  return ((const float*)p)[i];
}

__global__ void sel_init_kernel(SelState* st, long long k) {
  if (threadIdx.x < 256) st->hist[threadIdx.x] = 0;
  if (threadIdx.x == 0) { st->prefix = 0; st->mask = 0; st->k = k; }
}
template <typename T, bool PRED>
__global__ __launch_bounds__(256) void sel_hist_kernel(const void* v, SelState* st, int shift, long n) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const unsigned prefix = st->prefix, mask = st->mask;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned key = fkey(load_v<T, PRED>(v, i));
    if ((key & mask) == prefix) atomicAdd(&h[(key >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void sel_pick_kernel(SelState* st, int shift) {
  if (threadIdx.x == 0) {
    long long k = st->k, cum = 0;
    int b = 0;
    for (; b < 256; ++b) {
      if (cum + (long long)st->hist[b] > k) break;
      cum += st->hist[b];
    }
    st->prefix |= (unsigned)b << shift;
    st->mask |= 255u << shift;
    st->k = k - cum;
  }
  __syncthreads();
  if (threadIdx.x < 256) st->hist[threadIdx.x] = 0;
}

// Deterministic reductions: a kernel's block b leaves its partial sums in parts[j][b] (plain stores, j < 3 quantities, at
// most PARTS_ROWS blocks); finish_sums_kernel (one block) adds the rows in a fixed order into stats[slot_j] (scaled,
// accumulating when asked: the Sobel sum runs over the scales).  The first version used fp32 atomics: s_p, s_q and the
// chain-rule sums feed the gradient, which then differed in its last bits from run to run.
constexpr int PARTS_ROWS = 1024;
__device__ __forceinline__ void block_partial(float v, float* __restrict__ row) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __shared__ float pp[4];      // (called up to three times per kernel with distinct rows: every call ends in a barrier)
  if ((threadIdx.x & 63) == 0) pp[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) row[blockIdx.x] = pp[0] + pp[1] + pp[2] + pp[3];
  __syncthreads();
}
__global__ __launch_bounds__(256) void finish_sums_kernel(const float* __restrict__ parts, int rows, float* __restrict__ stats,
                                                          int slot0, int slot1, int slot2, float scale, int accumulate) {
  __shared__ float wsum[4];
  const int slots[3] = {slot0, slot1, slot2};
  for (int j = 0; j < 3; ++j) {
    if (slots[j] < 0) continue;
    float acc = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) acc += parts[j * PARTS_ROWS + r];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      const float tot = (wsum[0] + wsum[1] + wsum[2] + wsum[3]) * scale;
      stats[slots[j]] = accumulate ? stats[slots[j]] + tot : tot;
    }
    __syncthreads();
  }
}

// stats: [0] t_p [1] s_p [2] t_q [3] s_q [4] sum|R| [5] sobel sum [6] sumG [7] sumG*(p-t) [8] sum sign(p-t)  [9] (int) median idx
template <typename T, bool PRED>
__global__ __launch_bounds__(256) void absdev_kernel(const void* v, const SelState* st, float* stats, float* parts, int slot,
                                                     long n) {
  const float t = fkey_inv(st->prefix);
  float acc = 0.f;
  int best = 0x7fffffff;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = load_v<T, PRED>(v, i);
    acc += fabsf(x - t);
    if (PRED && x == t && (int)i < best) best = (int)i;
  }
  if (PRED && best != 0x7fffffff) atomicMin(reinterpret_cast<int*>(&stats[9]), best);     // (integer: order-independent)
  if (blockIdx.x == 0 && threadIdx.x == 0) stats[slot] = t;
  block_partial(acc, parts);
}

template <typename T>
__global__ __launch_bounds__(256) void residual_kernel(const uint16_t* p, const float* q, const float* stats, float* parts,
                                                       float* R, long n) {
  const float tp = stats[0], sp = stats[1], tq = stats[2], sq = stats[3];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float r = (f32_of_bits<T>(p[i * 8]) - tp) / sp - (q[i] - tq) / sq;
    R[i] = r;
    acc += fabsf(r);
  }
  block_partial(acc, parts);
}

// Sobel responses of scale k at (y, x) of the strided map R[::st, ::st]
__device__ __forceinline__ void sobel_at(const float* __restrict__ base, int w, int st, int y, int x, float& rx, float& ry) {
  float v[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) v[a][c] = base[(long)((y + a) * st) * w + (x + c) * st];
  // sobelx = [[1,0,-1],[2,0,-2],[1,0,-1]], sobely = [[1,2,1],[0,0,0],[-1,-2,-1]]   (cross-correlation, losses.py:245-246)
  rx = v[0][0] - v[0][2] + 2.f * (v[1][0] - v[1][2]) + v[2][0] - v[2][2];
  ry = v[0][0] + 2.f * v[0][1] + v[0][2] - v[2][0] - 2.f * v[2][1] - v[2][2];
}

// one thread per valid Sobel position of scale k: |Rx| + |Ry| into per-block partial rows
__global__ __launch_bounds__(256) void sobel_kernel(const float* R, float* parts, int b, int h, int w, int k) {
  const int st = 1 << k, hk = h >> k, wk = w >> k;
  const int oh = hk - 2, ow = wk - 2;
  const long total = (oh > 0 && ow > 0) ? (long)b * oh * ow : 0;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % ow);
    const long r = i / ow;
    const int y = (int)(r % oh);
    const long n = r / oh;
    float rx, ry;
    sobel_at(R + n * (long)h * w, w, st, y, x, rx, ry);
    acc += fabsf(rx) + fabsf(ry);
  }
  block_partial(acc, parts);
}

// The gradient of that sum in GATHER form: one thread per pixel of the scale-k grid adds the sub-gradients of the <= 9
// windows that contain it (re-evaluating their responses) -- G[pixel] += ..., one writer per pixel, the scales in launch
// order: no atomics (the first version scattered with fp32 atomics).
__global__ __launch_bounds__(256) void sobel_grad_kernel(const float* R, float* G, int b, int h, int w, int k, float gscale) {
  const int st = 1 << k, hk = h >> k, wk = w >> k;
  const int oh = hk - 2, ow = wk - 2;
  if (oh <= 0 || ow <= 0) return;
  const long total = (long)b * hk * wk;
  const float wx[3][3] = {{1, 0, -1}, {2, 0, -2}, {1, 0, -1}}, wy[3][3] = {{1, 2, 1}, {0, 0, 0}, {-1, -2, -1}};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % wk);
    const long r = i / wk;
    const int Y = (int)(r % hk);
    const long n = r / hk;
    const float* base = R + n * (long)h * w;
    float g = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int y = Y - a, x = X - c;                  // the window whose tap (a, c) is this pixel
        if (y < 0 || y >= oh || x < 0 || x >= ow) continue;
        float rx, ry;
        sobel_at(base, w, st, y, x, rx, ry);
        const float sx = rx > 0.f ? gscale : (rx < 0.f ? -gscale : 0.f), sy = ry > 0.f ? gscale : (ry < 0.f ? -gscale : 0.f);
        g += wx[a][c] * sx + wy[a][c] * sy;
      }
    if (g != 0.f) G[n * (long)h * w + (long)(Y * st) * w + X * st] += g;
  }
}

// G += simse term; reductions for the chain rule through (p - t)/s
template <typename T>
__global__ __launch_bounds__(256) void grad_sums_kernel(const uint16_t* p, const float* R, float* G, const float* stats,
                                                        float* parts, float simse_w, long n) {
  const float tp = stats[0];
  float a = 0.f, b = 0.f, c = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float r = R[i];
    const float g = G[i] + (r > 0.f ? simse_w : (r < 0.f ? -simse_w : 0.f));
    G[i] = g;
    const float d = f32_of_bits<T>(p[i * 8]) - tp;
    a += g;
    b += g * d;
    c += d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  }
  block_partial(a, parts);
  block_partial(b, parts + PARTS_ROWS);
  block_partial(c, parts + 2 * PARTS_ROWS);
}
template <typename T>
__global__ __launch_bounds__(256) void grad_final_kernel(const uint16_t* p, const float* G, const float* stats,
                                                         uint16_t* dp, float weight, long n) {
  const float tp = stats[0], sp = stats[1];
  const float sumG = stats[6], sumGd = stats[7], sumSign = stats[8];
  const int med = *reinterpret_cast<const int*>(&stats[9]);
  const float coef = sumGd / (sp * sp) / (float)n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = f32_of_bits<T>(p[i * 8]) - tp;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    float g = G[i] / sp - coef * sg;
    if ((int)i == med) g += -sumG / sp + coef * sumSign;
    u32x4 o = (u32x4){pack2<T>(weight * g, 0.f), 0u, 0u, 0u};
    reinterpret_cast<u32x4*>(dp)[i] = o;
  }
}
__global__ void sigm_loss_kernel(const float* stats, float* loss, float weight, float inv_np, float gm, float bsz) {
  if (threadIdx.x == 0) atomicAdd(loss, weight * (0.5f * inv_np * stats[4] + gm * inv_np * bsz * stats[5]));
}

inline int g1(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g)); }

template <typename T>
int run(const void* pred, const float* target, float* loss, void* dpred, int b, int h, int w, float gmweight, int scales,
        float weight, void* workspace, hipStream_t s) {
  const long n = (long)b * h * w;
  char* wp = (char*)workspace;
  SelState* st_p = (SelState*)wp; wp += 2048;
  SelState* st_q = (SelState*)wp; wp += 2048;
  float* stats = (float*)wp; wp += 256;
  float* parts = (float*)wp; wp += 3 * PARTS_ROWS * sizeof(float);
  float* R = (float*)wp; wp += ((size_t)n * 4 + 255) / 256 * 256;
  float* G = (float*)wp;
  hipError_t e = hipMemsetAsync(stats, 0, 64, s);
  if (e == hipSuccess && dpred) e = hipMemsetAsync(G, 0, (size_t)n * 4, s);
  if (e != hipSuccess) {
    cgan_set_error("sigm: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return CGAN_ERR_HIP;
  }
  const long long k = (n - 1) / 2;           // torch.median: the lower median
  hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(256), 0, s, st_p, k);
  hipLaunchKernelGGL(sel_init_kernel, dim3(1), dim3(256), 0, s, st_q, k);
  for (int shift = 24; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL((sel_hist_kernel<T, true>), dim3(g1(n)), dim3(256), 0, s, pred, st_p, shift, n);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(256), 0, s, st_p, shift);
    hipLaunchKernelGGL((sel_hist_kernel<T, false>), dim3(g1(n)), dim3(256), 0, s, (const void*)target, st_q, shift, n);
    hipLaunchKernelGGL(sel_pick_kernel, dim3(1), dim3(256), 0, s, st_q, shift);
  }
  e = hipMemsetAsync(&stats[9], 0x7f, 4, s);     // "no index yet" for the atomicMin below (0x7f7f7f7f)
  if (e != hipSuccess) {
    cgan_set_error("sigm: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return CGAN_ERR_HIP;
  }
#define FINISH(rows, s0, s1, s2, scale, accum) \
  hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(256), 0, s, (const float*)parts, rows, stats, s0, s1, s2, scale, accum)
  const int gn = g1(n);
  hipLaunchKernelGGL((absdev_kernel<T, true>), dim3(gn), dim3(256), 0, s, pred, (const SelState*)st_p, stats, parts, 0, n);
  FINISH(gn, 1, -1, -1, 1.f / (float)n, 0);
  hipLaunchKernelGGL((absdev_kernel<T, false>), dim3(gn), dim3(256), 0, s, (const void*)target, (const SelState*)st_q,
                     stats, parts, 2, n);
  FINISH(gn, 3, -1, -1, 1.f / (float)n, 0);
  hipLaunchKernelGGL(residual_kernel<T>, dim3(gn), dim3(256), 0, s, (const uint16_t*)pred, target, (const float*)stats, parts,
                     R, n);
  FINISH(gn, 4, -1, -1, 1.f, 0);
  const float inv_np = 1.f / (float)((long)h * w);
  for (int kk = 0; kk < scales; ++kk) {
    const int gk = g1(n >> (2 * kk));
    hipLaunchKernelGGL(sobel_kernel, dim3(gk), dim3(256), 0, s, (const float*)R, parts, b, h, w, kk);
    FINISH(gk, 5, -1, -1, 1.f, 1);                                   // (stats[5] starts at 0: the memset above)
    if (dpred)
      hipLaunchKernelGGL(sobel_grad_kernel, dim3(gk), dim3(256), 0, s, (const float*)R, G, b, h, w, kk,
                         gmweight * inv_np * (float)b);
  }
  hipLaunchKernelGGL(sigm_loss_kernel, dim3(1), dim3(64), 0, s, (const float*)stats, loss, weight, inv_np, gmweight,
                     (float)b);
  if (dpred) {
    hipLaunchKernelGGL(grad_sums_kernel<T>, dim3(gn), dim3(256), 0, s, (const uint16_t*)pred, (const float*)R, G,
                       (const float*)stats, parts, 0.5f * inv_np, n);
    FINISH(gn, 6, 7, 8, 1.f, 0);
    hipLaunchKernelGGL(grad_final_kernel<T>, dim3(gn), dim3(256), 0, s, (const uint16_t*)pred, (const float*)G,
                       (const float*)stats, (uint16_t*)dpred, weight, n);
  }
#undef FINISH
  return CGAN_OK;
}

}  // namespace

extern "C" size_t cgan_sigm_loss_workspace_bytes(int32_t b, int32_t h, int32_t w) {
  if (b <= 0 || h <= 0 || w <= 0) return 0;
  const size_t n = (size_t)b * h * w;
  return 2048 * 2 + 256 + 3 * PARTS_ROWS * sizeof(float) + 2 * ((n * 4 + 255) / 256 * 256);
}

extern "C" int cgan_sigm_loss_nhwc(const void* pred, const float* target, int32_t dtype, int32_t b, int32_t h, int32_t w,
                                   float gmweight, int32_t scales, float weight, float* loss_accum, void* dpred,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  CGAN_REQUIRE(pred && target && loss_accum && workspace, "sigm_loss: null pointer");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "sigm_loss: bad dtype %d", dtype);
  CGAN_REQUIRE(b > 0 && h > 2 && w > 2 && scales >= 1 && scales <= 8, "sigm_loss: bad shape");
  CGAN_REQUIRE(workspace_bytes >= cgan_sigm_loss_workspace_bytes(b, h, w), "sigm_loss: workspace too small");
  CGAN_REQUIRE(sizeof(SelState) <= 2048, "sigm_loss: internal");
  hipStream_t s = (hipStream_t)stream;
  int rc = dtype == CGAN_F16 ? run<F16>(pred, target, loss_accum, dpred, b, h, w, gmweight, scales, weight, workspace, s)
                             : run<BF16>(pred, target, loss_accum, dpred, b, h, w, gmweight, scales, weight, workspace, s);
  if (rc != CGAN_OK) return rc;
  CGAN_CHECK_LAUNCH("sigm_loss");
  return CGAN_OK;
}
