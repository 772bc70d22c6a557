// Instance-norm statistics over H*W per (n, c) on NHWC 16-bit tensors, and the elementwise normalise+act.
//
// HBM-bound: one pass over x.  Thread t of a block owns one 8-channel group (16-byte vector loads, lanes
// sweep consecutive channel groups then consecutive pixels => fully coalesced rows) and a strided subset
// of the block's pixel range; per-thread (sum, sumsq) over <= a few hundred pixels are converted to
// (mean, M2) and merged with Chan's parallel formula across threads (LDS) and across blocks (finalize
// kernel), so the variance never suffers the E[x^2] - E[x]^2 cancellation over 409 600 pixels.
#include "cgan_common.h"
#include <type_traits>

// Pre-merge of long partial lists (round 6): stats_premerge.hip (its own translation unit: it is compiled without the
// vectorisers, see the Makefile's NOVEC note -- the SLP vectoriser turns chan_merge's pairs into op_sel-modified packed-fp32
// instructions, which tests/test_build_isa.py refuses).
int stats_premerge_launch(float* partial, int hw, int cs, int chunks, int ppb, int len, int groups, hipStream_t s);

namespace {

constexpr int STATS_THREADS = 256;
constexpr int STATS_MAX_PIX_PER_BLOCK = 2048;  // pixels per block (per channel-group block), upper bound

struct MeanM2 {
  float n, mean, m2;
};
__device__ __forceinline__ MeanM2 chan_merge(MeanM2 a, MeanM2 b) {
  if (b.n == 0.f) return a;
  if (a.n == 0.f) return b;
  float n = a.n + b.n;
  float d = b.mean - a.mean;
  float w = b.n * __frcp_rn(n);   // counts are small integers: the 1-ulp reciprocal is ample
  MeanM2 r;
  r.n = n;
  r.mean = a.mean + d * w;
  r.m2 = a.m2 + b.m2 + d * d * (a.n * w);
  return r;
}

// partial: [n][chunks][cs][2] (mean, m2); counts are implied by the chunk extents
template <typename T>
__global__ __launch_bounds__(STATS_THREADS) void instnorm_partial_kernel(const uint16_t* __restrict__ x,
                                                                        float* __restrict__ partial, int hw, int cs,
                                                                        int chunks, int ppb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [PL][cgb*8][2]
  const int cg_total = cs / 8;
  const int cgb = cg_total < STATS_THREADS ? cg_total : STATS_THREADS;  // channel groups handled per block
  const int PL = STATS_THREADS / cgb;                                   // pixel lanes
  const int n = blockIdx.z;
  const int cg0 = blockIdx.y * cgb;
  const int chunk = blockIdx.x;
  const int p0 = chunk * ppb;
  const int p1 = min(hw, p0 + ppb);
  const int t = threadIdx.x;
  const int cgl = t % cgb, pl = t / cgb;
  const int cg = cg0 + cgl;
  const bool active = (pl < PL) && (cg < cg_total);

  float s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
  float cnt = 0.f;
  if (active) {
    const uint16_t* base = x + (size_t)n * hw * cs + cg * 8;
    // (four loads in flight per lane: with one, 16 waves per CU kept 16 KB on their way and the pass ran at 1.7 TB/s; the
    // sums are still formed pixel by pixel in the same order)
#pragma unroll 4
    for (int p = p0 + pl; p < p1; p += PL) {
      u32x4 v = *reinterpret_cast<const u32x4*>(base + (size_t)p * cs);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a, b;
        unpack2<T>(v[e], a, b);
        s[2 * e] += a; q[2 * e] += a * a;
        s[2 * e + 1] += b; q[2 * e + 1] += b * b;
      }
      cnt += 1.f;
    }
  }
  // per-thread (mean, m2) -> LDS; thread-local counts differ by at most one, carried in a parallel array
  float* smean = sm;                       // [PL][cgb*8]
  float* sm2 = sm + PL * cgb * 8;          // [PL][cgb*8]
  float* scnt = sm + 2 * PL * cgb * 8;     // [PL]
  if (pl < PL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float mean = cnt > 0.f ? s[e] / cnt : 0.f;
      float m2 = cnt > 0.f ? fmaxf(q[e] - s[e] * mean, 0.f) : 0.f;
      smean[(pl * cgb + cgl) * 8 + e] = mean;
      sm2[(pl * cgb + cgl) * 8 + e] = m2;
    }
    if (cgl == 0) scnt[pl] = cnt;
  }
  __syncthreads();
  // one thread per channel merges the PL pixel lanes
  for (int c = t; c < cgb * 8; c += STATS_THREADS) {
    if (cg0 * 8 + c >= cs) continue;
    MeanM2 acc = {0.f, 0.f, 0.f};
    for (int l = 0; l < PL; ++l) {
      MeanM2 b = {scnt[l], smean[l * cgb * 8 + c], sm2[l * cgb * 8 + c]};
      acc = chan_merge(acc, b);
    }
    float* o = partial + (((size_t)n * chunks + chunk) * cs + cg0 * 8 + c) * 2;
    o[0] = acc.mean;
    o[1] = acc.m2;
  }
}

// Optional training-mode BatchNorm epilogue of the finalize step (n_total == 1: the batch viewed as one image): the
// (mean', rstd') pair cgan_norm_act_apply consumes, the running statistics and the step counter, exactly as
// bn_train_prepare_kernel (train_ops.hip) computes them -- one launch less per BatchNorm forward.
struct BnEpilogue {
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  float* mean_out;     // null: plain instance-norm statistics
  float* rstd_out;
  long long* num_batches_tracked;
  float momentum;
  int c;
  CGAN_DEV_ONLY(int jitter_ppm; unsigned jitter_seed;)     // dev: the sensitivity experiment of DESIGN 4.13
};

constexpr int PREMERGE_MIN_CHUNKS = 512;   // shorter lists go straight to the finalize kernel
constexpr int PREMERGE_ROWS = 256;         // rows a block merges (8 per thread)

// one wave per (n, c): lanes stride over the chunk partials, then a 6-step shuffle tree of Chan merges
__global__ __launch_bounds__(256) void instnorm_finalize_kernel(const float* __restrict__ partial,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                int n_total, int hw, int cs, int chunks, int ppb,
                                                                float eps, BnEpilogue bn, int rows_per_group,
                                                                int row_stride) {
  const int idx0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // instance norm: one wave per (n, c).  Batch norm with several GROUPS (n_total > 1: the batch is a concatenation of
  // groups that the reference normalises in separate forward calls, e.g. the real and the simulated domain batch): one
  // wave per channel walks the groups IN ORDER, because the running statistics are updated once per group, sequentially.
  const bool grouped_bn = bn.mean_out != nullptr && n_total > 1;
  if (idx0 >= (grouped_bn ? cs : n_total * cs)) return;
  const int n_first = grouped_bn ? 0 : idx0 / cs, n_last = grouped_bn ? n_total : n_first + 1;
  const int c = grouped_bn ? idx0 : idx0 - n_first * cs;
  // four partial rows in flight per lane (the rows of one lane are cs * 8 bytes apart: every load is its own cache
  // line, so the latency has to be overlapped); merged in the same order as a plain loop
  auto load_rows = [&](int n, int k0, float2 (&v)[4], int (&cnt)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 64 * j;
      cnt[j] = 0;
      v[j] = make_float2(0.f, 0.f);
      if (k < chunks) {
        const int p0 = k * ppb;
        cnt[j] = min(hw, p0 + ppb) - p0;
        v[j] = *reinterpret_cast<const float2*>(partial + (((size_t)n * rows_per_group + (size_t)k * row_stride) * cs + c) * 2);
      }
    }
  };
  // The kernel is a chain of memory round trips (per group: the rows, then the epilogue's operands), 292 launches per joint
  // train step whatever the batch: everything that does not depend on a result is requested up front -- the first rows of
  // group n + 1 before group n is merged, and the epilogue's per-channel operands (the running statistics travel through
  // registers from group to group: same arithmetic, same order) before the first row arrives.
  float2 vn[4];
  int cn[4];
  load_rows(n_first, lane, vn, cn);
  float bn_g = 1.f, bn_b = 0.f, run_m = 0.f, run_v = 0.f;
  if (bn.mean_out && c < bn.c) {
    if (bn.gamma) bn_g = bn.gamma[c];
    if (bn.beta) bn_b = bn.beta[c];
    if (bn.running_mean) {
      run_m = bn.running_mean[c];
      run_v = bn.running_var[c];
    }
  }
  for (int n = n_first; n < n_last; ++n) {
  const int idx = n * cs + c;
  MeanM2 acc = {0.f, 0.f, 0.f};
  for (int k0 = lane; k0 < chunks; k0 += 256) {
    float2 v[4];
    int cnt[4];
    if (k0 == lane) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = vn[j]; cnt[j] = cn[j]; }
      if (n + 1 < n_last) load_rows(n + 1, lane, vn, cn);
    } else {
      load_rows(n, k0, v, cnt);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (cnt[j] > 0) {
        MeanM2 b = {(float)cnt[j], v[j].x, v[j].y};
        acc = chan_merge(acc, b);
      }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    MeanM2 b;
    b.n = __shfl_down(acc.n, off, 64);
    b.mean = __shfl_down(acc.mean, off, 64);
    b.m2 = __shfl_down(acc.m2, off, 64);
    acc = chan_merge(acc, b);
  }
  if (lane == 0) {
    const float mu = acc.mean;
    float rs = rsqrtf(acc.m2 / (float)hw + eps);
#ifdef CGAN_DEV
    if (bn.mean_out && bn.jitter_ppm) {       // batch rstd times (1 + u ppm), u uniform in [-1, 1] per (layer call, group, channel)
      unsigned hsh = (unsigned)idx * 2654435761u ^ bn.jitter_seed * 40503u ^ (unsigned)(size_t)bn.mean_out;
      hsh ^= hsh >> 15; hsh *= 2246822519u; hsh ^= hsh >> 13;
      rs *= 1.f + (float)bn.jitter_ppm * 1e-6f * ((float)(hsh & 0xffff) / 32767.5f - 1.f);
    }
#endif
    mean[idx] = mu;
    rstd[idx] = rs;
    if (bn.mean_out) {
      if (c == 0 && bn.num_batches_tracked) *bn.num_batches_tracked += 1;     // once per group = per reference forward
      float m = 0.f, r = 0.f;
      if (c < bn.c) {
        r = rs * bn_g;
        m = mu - ((bn.beta && r != 0.f) ? bn_b / r : 0.f);
        if (bn.running_mean) {
          const float count = (float)hw;
          const float var_b = 1.f / (rs * rs) - eps;
          const float var_u = count > 1.f ? var_b * count / (count - 1.f) : var_b;
          run_m = (1.f - bn.momentum) * run_m + bn.momentum * mu;
          run_v = (1.f - bn.momentum) * run_v + bn.momentum * var_u;
          bn.running_mean[c] = run_m;
          bn.running_var[c] = run_v;
        }
      }
      bn.mean_out[idx] = m;
      bn.rstd_out[idx] = r;
    }
  }
  }
}

// A thread keeps its 8 channels (mean / rstd in registers) and walks the pixels of image blockIdx.y: no per-element
// index division and no per-element statistic loads (the first version spent more on those than on the data).
template <typename T>
__global__ __launch_bounds__(256) void norm_act_apply_kernel(const uint16_t* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const uint16_t* __restrict__ res,
                                                             uint16_t* __restrict__ y, int hw, int cs, int c, int act,
                                                             float slope) {
  const int cg_total = cs / 8;
  const int tpp = cg_total < 256 ? cg_total : 256;   // threads per pixel
  const int rows = 256 / tpp;
  const int cgl = threadIdx.x % tpp, prow = threadIdx.x / tpp;
  if (prow >= rows) return;
  const int n = blockIdx.y;
  const uint16_t* xn = x + (size_t)n * hw * cs;
  uint16_t* yn = y + (size_t)n * hw * cs;
  const uint16_t* rn = res ? res + (size_t)n * hw * cs : nullptr;
  for (int cg = cgl; cg < cg_total; cg += tpp) {
    float m[8], r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      m[e] = mean[(size_t)n * cs + cg * 8 + e];
      r[e] = rstd[(size_t)n * cs + cg * 8 + e];
    }
    // what does not depend on the pixel is decided once per thread: pad channels in this group, a residual, the activation
    // (act_apply_n: wave-uniform) -- the loop body is load, 8 x (unpack, subtract, fma), activation, pack, store
    const bool pad = cg * 8 + 8 > c;
    auto body = [&](auto res_tag) {
      constexpr bool RES = decltype(res_tag)::value;
#pragma unroll 2
      for (int p = blockIdx.x * rows + prow; p < hw; p += gridDim.x * rows) {
        const size_t off = (size_t)p * cs + cg * 8;
        const u32x4 v = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(xn + off));
        u32x4 rv = {0u, 0u, 0u, 0u};
        if (RES) rv = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(rn + off));
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float a, b;
          unpack2<T>(v[e], a, b);
          f[2 * e] = (a - m[2 * e]) * r[2 * e];
          f[2 * e + 1] = (b - m[2 * e + 1]) * r[2 * e + 1];
          if (RES) {
            float ra, rb;
            unpack2<T>(rv[e], ra, rb);
            f[2 * e] += ra;                    // (fused by the compiler into the multiply above: same rounding as before)
            f[2 * e + 1] += rb;
          }
        }
        act_apply_n(f, act, slope);
        if (pad) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (cg * 8 + e >= c) f[e] = 0.f;
        }
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2<T>(f[2 * e], f[2 * e + 1]);
        CGAN_ST_STREAM(o, reinterpret_cast<u32x4*>(yn + off));
      }
    };
    if (rn) body(std::true_type{});
    else body(std::false_type{});
  }
}

// pixels per block: as large as possible (fewer partials to merge) while the grid still fills the chip
int pix_per_block(const CganNormStatsDesc* d) {
  const int cg_total = cgan_cs(d->c) / 8;
  const int cgb = cg_total < STATS_THREADS ? cg_total : STATS_THREADS;
  const long cgblocks = ceil_div(cg_total, cgb);
  int ppb = STATS_MAX_PIX_PER_BLOCK;
  while (ppb > 64 && (long)ceil_div(d->hw, ppb) * cgblocks * d->n < 1024) ppb >>= 1;
  return ppb;
}

int check(const CganNormStatsDesc* d) {
  CGAN_REQUIRE(d != nullptr, "instnorm: null descriptor");
  CGAN_REQUIRE(d->dtype == CGAN_F16 || d->dtype == CGAN_BF16, "instnorm: bad dtype %d", d->dtype);
  CGAN_REQUIRE(d->n > 0 && d->hw > 0 && d->c > 0, "instnorm: bad shape");
  return CGAN_OK;
}

}  // namespace

#ifdef CGAN_DEV
// dev: perturb every training-mode BatchNorm's batch rstd by up to +-ppm (uniform, hashed per layer call / group / channel):
// how far do the step's gradients move for an error of a given size in the statistics (R5 DESIGN 4.13)
static int g_bn_jitter_ppm = 0;
static unsigned g_bn_jitter_seed = 0;
extern "C" void cgan_debug_set_bn_jitter(int ppm, int seed) { g_bn_jitter_ppm = ppm; g_bn_jitter_seed = (unsigned)seed; }
#endif

extern "C" size_t cgan_instnorm_stats_workspace_bytes(const CganNormStatsDesc* d) {
  if (check(d) != CGAN_OK) return 0;
  int chunks = ceil_div(d->hw, 64);  // worst case of pix_per_block()
  return (size_t)d->n * chunks * cgan_cs(d->c) * 2 * sizeof(float);
}

static int stats_impl(const void* x, float* mean, float* rstd, const CganNormStatsDesc* d, void* workspace,
                      size_t workspace_bytes, void* stream, const BnEpilogue& bn) {
  int rc = check(d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(x && mean && rstd && workspace, "instnorm_stats: null pointer");
  size_t need = cgan_instnorm_stats_workspace_bytes(d);
  if (workspace_bytes < need) {
    cgan_set_error("instnorm_stats: workspace %zu B < required %zu B", workspace_bytes, need);
    return CGAN_ERR_WORKSPACE;
  }
  const int cs = cgan_cs(d->c);
  const int cg_total = cs / 8;
  const int cgb = cg_total < STATS_THREADS ? cg_total : STATS_THREADS;
  const int PL = STATS_THREADS / cgb;
  const int ppb = pix_per_block(d);
  const int chunks = ceil_div(d->hw, ppb);
  dim3 grid(chunks, ceil_div(cg_total, cgb), d->n);
  size_t smem = ((size_t)2 * PL * cgb * 8 + PL) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == CGAN_F16)
    hipLaunchKernelGGL(instnorm_partial_kernel<F16>, grid, dim3(STATS_THREADS), smem, s, (const uint16_t*)x,
                       (float*)workspace, d->hw, cs, chunks, ppb);
  else
    hipLaunchKernelGGL(instnorm_partial_kernel<BF16>, grid, dim3(STATS_THREADS), smem, s, (const uint16_t*)x,
                       (float*)workspace, d->hw, cs, chunks, ppb);
  CGAN_CHECK_LAUNCH("instnorm_stats(partial)");
  int total = (bn.mean_out != nullptr && d->n > 1) ? cs : d->n * cs;      // grouped batch norm: one wave per channel
  hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(ceil_div(total, 4)), dim3(256), 0, s, (const float*)workspace,
                     mean, rstd, d->n, d->hw, cs, chunks, ppb, d->eps, bn, chunks, 1);
  CGAN_CHECK_LAUNCH("instnorm_stats(finalize)");
  return CGAN_OK;
}

extern "C" int cgan_instnorm_stats(const void* x, float* mean, float* rstd, const CganNormStatsDesc* d, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  BnEpilogue none = {};
  return stats_impl(x, mean, rstd, d, workspace, workspace_bytes, stream, none);
}

extern "C" int cgan_batchnorm_train_stats(const void* x, const float* gamma, const float* beta, float momentum,
                                          float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                          float* batch_mean, float* batch_rstd, float* mean_out, float* rstd_out,
                                          const CganNormStatsDesc* d, void* workspace, size_t workspace_bytes,
                                          void* stream) {
  // d->n = number of GROUPS normalised independently (1: the whole batch, as nn.BatchNorm2d; G > 1: G equal slices of
  // the batch, each what the reference would have passed through the layer in its own forward call), d->hw = pixels per
  // group (n/G * h * w).  Outputs are [G][cs]; running statistics are updated group after group.
  CGAN_REQUIRE(d != nullptr && d->n >= 1 && d->n <= 16, "batchnorm_train_stats: 1..16 groups of n/G*h*w pixels each");
  CGAN_REQUIRE(mean_out && rstd_out, "batchnorm_train_stats: null pointer");
  CGAN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "batchnorm_train_stats: running stats go together");
  BnEpilogue bn = {gamma, beta, running_mean, running_var, mean_out, rstd_out, (long long*)num_batches_tracked, momentum,
                   d->c};
  CGAN_DEV_ONLY(bn.jitter_ppm = g_bn_jitter_ppm; bn.jitter_seed = g_bn_jitter_seed;)
  return stats_impl(x, batch_mean, batch_rstd, d, workspace, workspace_bytes, stream, bn);
}

extern "C" int cgan_batchnorm_train_stats_from_partials(float* partial, int32_t chunk_pixels, const float* gamma,
                                                        const float* beta, float momentum, float* running_mean,
                                                        float* running_var, int64_t* num_batches_tracked,
                                                        float* batch_mean, float* batch_rstd, float* mean_out,
                                                        float* rstd_out, const CganNormStatsDesc* d, void* stream) {
  // the finalize half of cgan_batchnorm_train_stats on per-chunk (mean, M2) rows that a convolution's epilogue produced
  // (cgan_conv2d_nhwc_fwd_stats): [group][chunk][cs][2], chunk = chunk_pixels consecutive pixels, d->hw (pixels per group)
  // a whole number of chunks.  The rows are CONSUMED: long lists are shortened in place first (stats_premerge_kernel)
  int rc = check(d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(partial && batch_mean && batch_rstd && mean_out && rstd_out, "batchnorm_train_stats_from_partials: null pointer");
  CGAN_REQUIRE(d->n >= 1 && d->n <= 16, "batchnorm_train_stats_from_partials: 1..16 groups");
  CGAN_REQUIRE(chunk_pixels > 0 && d->hw % chunk_pixels == 0, "batchnorm_train_stats_from_partials: %d pixels per group are "
               "not a whole number of %d-pixel chunks", d->hw, chunk_pixels);
  CGAN_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "batchnorm_train_stats_from_partials: running stats go together");
  const int cs = cgan_cs(d->c);
  BnEpilogue bn = {gamma, beta, running_mean, running_var, mean_out, rstd_out, (long long*)num_batches_tracked, momentum, d->c};
  CGAN_DEV_ONLY(bn.jitter_ppm = g_bn_jitter_ppm; bn.jitter_seed = g_bn_jitter_seed;)
  const int total = d->n > 1 ? cs : d->n * cs;
  const int rows = d->hw / chunk_pixels;          // per group
  int chunks = rows, ppb = chunk_pixels, stride = 1;
  if (rows >= PREMERGE_MIN_CHUNKS) {
    const int len = PREMERGE_ROWS;
    const int merged = ceil_div(rows, len);
    rc = stats_premerge_launch(partial, d->hw, cs, rows, chunk_pixels, len, d->n, (hipStream_t)stream);
    if (rc != CGAN_OK) return rc;
    chunks = merged; ppb = chunk_pixels * len; stride = len;
  }
  hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(ceil_div(total, 4)), dim3(256), 0, (hipStream_t)stream, partial,
                     batch_mean, batch_rstd, d->n, d->hw, cs, chunks, ppb, d->eps, bn, rows, stride);
  CGAN_CHECK_LAUNCH("batchnorm_train_stats_from_partials");
  return CGAN_OK;
}

extern "C" int cgan_norm_add_act_apply(const void* x, const float* mean, const float* rstd, const void* residual,
                                       void* y, const CganNormStatsDesc* d, int32_t act, float act_slope,
                                       void* stream) {
  int rc = check(d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(x && mean && rstd && y, "norm_act_apply: null pointer");
  const int cs = cgan_cs(d->c);
  const int cg_total = cs / 8;
  const int tpp = cg_total < 256 ? cg_total : 256;
  const int rows = 256 / tpp;
  // ~4 pixels per thread, at most ~4096 workgroups over the batch
  long bx = ((long)d->hw + (long)rows * 4 - 1) / ((long)rows * 4);
  const long cap = 4096 / d->n > 1 ? 4096 / d->n : 1;
  bx = bx < 1 ? 1 : (bx > cap ? cap : bx);
  CGAN_REQUIRE(d->n <= 65535, "norm_act_apply: batch too large");
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == CGAN_F16)
    hipLaunchKernelGGL(norm_act_apply_kernel<F16>, dim3((unsigned)bx, d->n), dim3(256), 0, s, (const uint16_t*)x, mean,
                       rstd, (const uint16_t*)residual, (uint16_t*)y, d->hw, cs, d->c, act, act_slope);
  else
    hipLaunchKernelGGL(norm_act_apply_kernel<BF16>, dim3((unsigned)bx, d->n), dim3(256), 0, s, (const uint16_t*)x, mean,
                       rstd, (const uint16_t*)residual, (uint16_t*)y, d->hw, cs, d->c, act, act_slope);
  CGAN_CHECK_LAUNCH("norm_act_apply");
  return CGAN_OK;
}

extern "C" int cgan_norm_act_apply(const void* x, const float* mean, const float* rstd, void* y,
                                   const CganNormStatsDesc* d, int32_t act, float act_slope, void* stream) {
  return cgan_norm_add_act_apply(x, mean, rstd, nullptr, y, d, act, act_slope, stream);
}
