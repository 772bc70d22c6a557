// In-place pre-merge of a convolution epilogue's per-chunk BatchNorm statistics rows (cgan_batchnorm_train_stats_from_partials,
// norm_stats.hip).  Own translation unit, listed in the Makefile's NOVEC: the SLP vectoriser forms op_sel-modified packed-fp32
// instructions from chan_merge's pairs, whose results change on gfx950 while another stream runs MFMA / LDS-DMA kernels
// (R5 DESIGN 4.6) -- the two-stream train step stopped being bit-reproducible with them (tests/test_gpu_determinism.py).
#include "cgan_common.h"

namespace {

struct MeanM2 {
  float n, mean, m2;
};
__device__ __forceinline__ MeanM2 chan_merge(MeanM2 a, MeanM2 b) {      // as in norm_stats.hip
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  float n = a.n + b.n;
  float d = b.mean - a.mean;
  float w = b.n * __frcp_rn(n);
  MeanM2 r;
  r.n = n;
  r.mean = a.mean + d * w;
  r.m2 = a.m2 + b.m2 + d * d * (a.n * w);
  return r;
}

// A convolution's statistics epilogue writes one (mean, M2) row per 64 pixels:
// at 64 images of 160 x 160 that is 12 800 rows per group, which the finalize kernel below walks with ONE wave per channel,
// four cache lines in flight (33 us on average, up to 145 us, x 292 launches per train step at 32 per domain).  Here a
// block owns `len` consecutive rows x 8 channels (a row's 8 pairs = one 64-byte line, fully used), merges them in a fixed
// order -- 32 row lanes, each its rows in order, then the lanes in order -- and writes the result over the FIRST row of its
// own range: the block reads nothing another block writes and writes nothing another block reads, so the list is shortened
// in place; the finalize kernel then walks every len-th row with chunk size ppb * len.
__global__ __launch_bounds__(256) void stats_premerge_kernel(float* __restrict__ partial, int hw, int cs, int chunks,
                                                             int ppb, int len) {
  __shared__ float red[32][8][3];
  const int chl = threadIdx.x & 7, kl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + chl;
  const int k_lo = blockIdx.y * len, k_hi = min(chunks, k_lo + len);
  float* rows = partial + ((size_t)blockIdx.z * chunks * cs + c) * 2;
  MeanM2 acc = {0.f, 0.f, 0.f};
  for (int k0 = k_lo + kl; k0 < k_hi; k0 += 128) {
    float2 v[4];
    int cnt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 32 * j;
      cnt[j] = 0;
      v[j] = make_float2(0.f, 0.f);
      if (k < k_hi) {
        const int p0 = k * ppb;
        cnt[j] = min(hw, p0 + ppb) - p0;
        v[j] = *reinterpret_cast<const float2*>(rows + (size_t)k * cs * 2);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (cnt[j] > 0) {
        MeanM2 b = {(float)cnt[j], v[j].x, v[j].y};
        acc = chan_merge(acc, b);
      }
  }
  red[kl][chl][0] = acc.n; red[kl][chl][1] = acc.mean; red[kl][chl][2] = acc.m2;
  __syncthreads();                                   // every read of this block's rows is done
  if (kl != 0) return;
  acc = (MeanM2){0.f, 0.f, 0.f};
#pragma unroll 4
  for (int l = 0; l < 32; ++l) acc = chan_merge(acc, (MeanM2){red[l][chl][0], red[l][chl][1], red[l][chl][2]});
  *reinterpret_cast<float2*>(rows + (size_t)k_lo * cs * 2) = make_float2(acc.mean, acc.m2);
}

}  // namespace

int stats_premerge_launch(float* partial, int hw, int cs, int chunks, int ppb, int len, int groups, hipStream_t s) {
  hipLaunchKernelGGL(stats_premerge_kernel, dim3(cs / 8, ceil_div(chunks, len), groups), dim3(256), 0, s, partial, hw, cs,
                     chunks, ppb, len);
  CGAN_CHECK_LAUNCH("batchnorm_train_stats_from_partials(premerge)");
  return CGAN_OK;
}
