// Interface of the LDS-tiled implicit-GEMM convolution kernel for wide layers (conv_gemm.hip), used by
// conv_mfma.hip's dispatcher.
#pragma once
#include "cgan_common.h"

struct ConvGemmArgs {
  const uint16_t* x;
  const u32x4* w;       // packed [ctile][ks = tap * (cin_s/32) + cc][lane]
  const float* bias;    // padded to ctiles*16, or null
  const uint16_t* res;  // residual or null
  const uint16_t* res2; // has_res == 3 only: the ReLU output whose sign masks the result (cgan_res_apply3), else unused
  uint16_t* y;
  int n, h_in, w_in, cin_s;
  int cout, cout_s, ctiles, ksteps;
  int kh, kw, stride, pad, dil, pad_mode;
  int h_out, w_out, npix;
  int act, has_res, res_ups;
  float slope;
  // optional: per-(pixel chunk, channel) (mean, M2) of the fp32 results before bias-free rounding, [chunk][cout_s][2],
  // chunk = conv_gemm_stats_chunk_pixels() consecutive output pixels (training-mode BatchNorm statistics without a
  // second pass over y); null = off
  float* stats;
};

// conv_gemm_ext.hip: the same tiling for strided data gradients by output parity classes and for split-K launches
struct ConvGemmCls {
  int koff;        // k-step offset of the class's block inside the packed operator (pack_dgrad_classes_kernel)
  int kh, kw;      // the class's taps per axis (0 x 0: the class only writes zeros)
  int offy, offx;  // first dy row / column an output (i, j) of the class reads: i + offy, j + offx
};
struct ConvGemmExtArgs {
  ConvGemmArgs a;
  int cls_s;       // > 0: parity-class mode of a stride-cls_s data gradient, blockIdx.y = class (a.h_out x a.w_out = dx extent)
  int ksplit;      // > 1: split-K, blockIdx.y = K slice of ks_per k-steps, fp32 partials to ws [slice][pixel][cout_s]
  int ks_per;
  float* ws;
  int pair;        // 1: split-precision output (cgan_conv2d_nhwc_fwd_pair): y / the residual are split maps of Split<T>::NB blocks
  ConvGemmCls cls[4];
};
bool conv_gemm_ext_shape_ok(const ConvGemmArgs& a);
int conv_gemm_splitk_plan(const ConvGemmArgs& a);        // K slices for a grid that cannot fill the chip (1 = none)
size_t conv_gemm_splitk_workspace_bytes(const ConvGemmArgs& a, int ksplit);
int conv_gemm_splitk_launch(const ConvGemmArgs& a, int ksplit, float* ws, int dtype, hipStream_t s);
int conv_gemm_cls_launch(const ConvGemmArgs& a, int cls_s, const ConvGemmCls* cls, int dtype, hipStream_t s);
// a plain (one class, one K range) launch whose epilogue stores the fp32 result as its 16-bit components (split-precision maps)
int conv_gemm_pair_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);

// true for convs whose channel counts make the 128x256 (cout x pixel) LDS tiling worthwhile
bool conv_gemm_applicable(const CganConvDesc* d);
int conv_gemm_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);
// true if the kernel conv_gemm_launch picks for ``a`` takes has_res == 3 (the shared store path: conv_gemm_staged_store)
bool conv_gemm_res2_ok(const ConvGemmArgs& a, int dtype);
// pixels per statistics chunk of the kernel conv_gemm_launch would run for ``a`` (0: that kernel writes no statistics,
// or npix is not a whole number of chunks)
int conv_gemm_stats_chunk_pixels(const ConvGemmArgs& a, int dtype);
// conv_gemm_big.hip: the 256 couts x 256 pixels / K = 64 kernel (eight waves, one workgroup per CU)
bool conv_gemm_big_ok(const ConvGemmArgs& a);
// conv1x1_direct.hip: 1x1 layers with <= 256 couts, activations straight into B-fragment registers, all couts per workgroup
bool conv1x1_allc_ok(const ConvGemmArgs& a);
int conv1x1_allc_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);
int conv_gemm_big_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);
bool conv_gemm_big_pair_ok(const ConvGemmArgs& a, int dtype);     // split-precision forward on the 256 x 256 tile (round 6)
int conv_gemm_big_pair_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);
// conv1x1_xres.hip: short-K (cin <= 256) 1x1 layers, activation tile resident in LDS, all couts per workgroup
bool conv1x1_xres_ok(const ConvGemmArgs& a);
int conv1x1_xres_launch(const ConvGemmArgs& a, int dtype, hipStream_t s);

#ifdef __HIPCC__
typedef float f32x2 __attribute__((ext_vector_type(2)));

// sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15), left in every lane of the row: four row rotations on the
// VALU's data-parallel-primitive path (__shfl_xor compiles to ds_bpermute_b32: an LDS instruction per step and value)
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

// Training-mode BatchNorm statistics from a wave's accumulator tile acc[WC][WP] (lane (j, g) holds channels
// ct*16 + 4g + {0..3} of pixel (tile, j)): one (mean, M2) pair per channel over the wave's WP x 16 pixels = statistics
// chunk ``chunk``, written to p.stats [chunk][cout_s][2].  The bias, if any, is folded into the accumulators first (the
// caller's store path must then skip it).  Shared by conv_gemm_kernel and conv_gemm_big_kernel.
template <typename T, int WC, int WP>
__device__ __forceinline__ void conv_gemm_stats_epilogue(f32x4 (&acc)[WC][WP], const ConvGemmArgs& p, int cout_base,
                                                         int chunk, int j, int g) {
  if (p.bias) {
    // statistics are of the values AS STORED = round(acc + bias): fold the bias into the accumulators here and let the
    // store path skip it (wave-uniform branch)
#pragma unroll
    for (int c = 0; c < WC; ++c) {
      const int chb = cout_base + c * 16 + 4 * g;
      const f32x4 b = chb < p.cout_s ? *reinterpret_cast<const f32x4*>(p.bias + chb) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < WP; ++t) acc[c][t] += b;
    }
  }
  // training-mode BatchNorm statistics from the accumulators: (mean, M2) of this wave's WP x 16 pixels per channel --
  // sum and sum of squares over the wave's pixel tiles, then over the 16 lanes (pixels) that share a channel quad; the
  // finalize kernel of norm_stats.hip merges the chunks with Chan's formula.  Every chunk is full (the dispatcher
  // checked npix against the chunk size), pixels past npix do not exist here.
  constexpr float inv_cnt = 1.f / (float)(WP * 16);
#pragma unroll
  for (int c = 0; c < WC; ++c) {
    // Statistics of the values AS STORED (rounded to the 16-bit type), not of the fp32 accumulators: a channel whose
    // spread is below the rounding step of its mean (post-ReLU inputs make such channels) is pure rounding noise in y,
    // and only the stored values' own variance normalises that noise to unit size -- with the accumulators' (true,
    // much smaller) variance it was amplified (batch variances up to 30 % apart in layer3 / layer4, the encoder's
    // gradient 2 % longer and 0.02 further from the reference's direction: tests/test_gpu_configs_640.py).
    // Two passes, pairs of channels on the packed-fp32 VALU path: the chunk mean first, then M2 = sum (v - mean)^2
    // (the one-pass form sum v^2 - (sum v)^2 / n cancels where the mean is large against the spread).
    f32x2 vr[WP][2];
#pragma unroll
    for (int t = 0; t < WP; ++t) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float r0, r1;
        unpack2<T>(pack2<T>(acc[c][t][2 * hh], acc[c][t][2 * hh + 1]), r0, r1);
        vr[t][hh] = (f32x2){r0, r1};
      }
    }
    f32x2 s0[2] = {vr[0][0], vr[0][1]};      // (no "0 + v": that is a packed add with an op_sel-modified constant, R5 DESIGN 4.6)
#pragma unroll
    for (int t = 1; t < WP; ++t) {
      s0[0] += vr[t][0];
      s0[1] += vr[t][1];
    }
    float sm[4], sq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) sm[r] = row16_sum(s0[r >> 1][r & 1]) * inv_cnt;     // the chunk mean, in every lane of the row
    const f32x2 m2[2] = {(f32x2){sm[0], sm[1]}, (f32x2){sm[2], sm[3]}};
    f32x2 s1[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const f32x2 dv = vr[0][hh] - m2[hh];
      s1[hh] = dv * dv;
    }
#pragma unroll
    for (int t = 1; t < WP; ++t)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x2 dv = vr[t][hh] - m2[hh];
        s1[hh] += dv * dv;
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) sq[r] = row16_sum(s1[r >> 1][r & 1]);
    const int ch = cout_base + c * 16 + 4 * g;
    if (j == 0 && ch < p.cout_s) {
      float o[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[2 * r] = sm[r];
        o[2 * r + 1] = sq[r];
      }
      float* dst = p.stats + ((size_t)chunk * p.cout_s + ch) * 2;
      *reinterpret_cast<f32x4*>(dst) = (f32x4){o[0], o[1], o[2], o[3]};
      *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){o[4], o[5], o[6], o[7]};
    }
  }
}

// Store path of the LDS-staged GEMM kernels (conv_gemm_kernel, conv_gemm_big_kernel): the wave's accumulator tile
// acc[WC][WP] (lane (j, g) holds channels ct*16 + 4g + {0..3} of pixel (tile, j)) goes through ``stg`` in fp32 (row = one
// pixel x the wave's WC*16 couts, + 16 B pad) PP pixel tiles at a time and leaves as 16-byte chunks with bias / residual /
// activation applied in fp32.  ``pix0``: first pixel of the wave's WP tiles; ``bias_ep``: the bias still to add (null if
// the statistics epilogue folded it into the accumulators).
//
// Round 6: every global READ of the store path is issued before the first store.  The per-chunk form (load the residual
// chunk, use it, store) compiled to ``global_load_dwordx4; s_waitcnt vmcnt(0)`` in front of every store -- vmcnt counts
// the stores too, so a wave went through WP*WC/2 fully serialized HBM round trips (16 for the 256 x 256 tile, one
// workgroup per CU: ~46 us of a 90-us workgroup on the bottleneck's 1x1 data gradients with the skip gradient added in
// the epilogue, 600 us where the same GEMM without the residual takes 310).  A lane's 8-channel chunk is the same in
// every iteration (64 % CH == 0): the bias is loaded once; the residual chunks of ALL passes (2*WC*WP registers, half the
// accumulators' count, free once the K loop's fragments are dead) are in flight together, then the stores stream out.
template <typename T, int WC, int WP, int PP, bool RES2 = false>
__device__ __forceinline__ void conv_gemm_staged_store(const f32x4 (&acc)[WC][WP], const ConvGemmArgs& p,
                                                       unsigned char* stg, int pix0, int cout_base, const float* bias_ep,
                                                       int lane, int j, int g) {
  constexpr int ROWB = WC * 64 + 16;
  constexpr int CH = WC * 2;                                  // 8-channel chunks per staged row
  constexpr int NIT = PP * 16 * CH / 64;                      // chunks per lane and pass
  constexpr int NPASS = WP / PP;
  constexpr int PSTEP = 64 / CH;                              // pixels between a lane's consecutive chunks
  static_assert(64 % CH == 0 && WP % PP == 0, "a lane keeps its channel chunk over the iterations");
  const int qc = lane % CH, pl0 = lane / CH;
  const int ch = cout_base + qc * 8;
  const bool ch_ok = ch < p.cout_s;
  const bool plain = !bias_ep && !p.has_res && p.act == CGAN_ACT_NONE && p.cout == p.cout_s;
  f32x4 bb0 = (f32x4){0.f, 0.f, 0.f, 0.f}, bb1 = bb0;
  if (bias_ep && ch_ok) {                                     // (padded to whole cout tiles: two 16-byte loads)
    bb0 = *reinterpret_cast<const f32x4*>(bias_ep + ch);
    bb1 = *reinterpret_cast<const f32x4*>(bias_ep + ch + 4);
  }
  u32x4 rv[NPASS][NIT];
  // has_res == 3 (RES2 instantiations only: the registers are not the common path's to spend): a second map (the mask's
  // source) per chunk.  Pass k + 1's chunks are requested once pass k's accumulators are staged (dead): everything is
  // still in flight before the one wait, except from the third pass on (tiles with one staging pass per pixel tile)
  u32x4 rm[RES2 ? (NPASS > 1 ? 2 : 1) : 1][RES2 ? NIT : 1];
  auto load_rm = [&](int pass, u32x4 (&dst)[RES2 ? NIT : 1]) {
#pragma unroll
    for (int it = 0; it < (RES2 ? NIT : 0); ++it) {
      const int pix = pix0 + pass * PP * 16 + it * PSTEP + pl0;
      dst[it] = (u32x4){0u, 0u, 0u, 0u};
      if (pix < p.npix && ch_ok) dst[it] = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(p.res2 + (size_t)pix * p.cout_s + ch));
    }
  };
  if (p.has_res) {                                            // wave-uniform
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pix = pix0 + pass * PP * 16 + it * PSTEP + pl0;
        rv[pass][it] = (u32x4){0u, 0u, 0u, 0u};
        if (pix < p.npix && ch_ok) {
          size_t rbase;
          if (p.res_ups) {
            const int ox = pix % p.w_out;
            const int r = pix / p.w_out;
            const int oy = r % p.h_out;
            const int nn = r / p.h_out;
            rbase = (((size_t)nn * (p.h_out >> 1) + (oy >> 1)) * (p.w_out >> 1) + (ox >> 1)) * p.cout_s;
          } else {
            rbase = (size_t)pix * p.cout_s;
          }
          rv[pass][it] = CGAN_LD_STREAM(reinterpret_cast<const u32x4*>(p.res + rbase + ch));
        }
      }
    if (RES2 && p.has_res == 3) load_rm(0, rm[0]);
    __builtin_amdgcn_sched_barrier(0);                        // the loads stay in front of the staging and the stores
  }
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
    for (int tt = 0; tt < PP; ++tt)
#pragma unroll
      for (int c = 0; c < WC; ++c)
        *reinterpret_cast<f32x4*>(stg + (tt * 16 + j) * ROWB + c * 64 + g * 16) = acc[c][pass * PP + tt];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (RES2 && NPASS > 1 && p.has_res == 3 && pass + 1 < NPASS) {
      __builtin_amdgcn_sched_barrier(0);
      load_rm(pass + 1, rm[(pass + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ONE wait for the bias / residual loads, as a builtin the compiler's wait-count pass sees, on every path: left to
    // itself it puts s_waitcnt vmcnt(0) in front of each chunk's first use of a prefetched register (the uses sit behind
    // wave-uniform branches, the merged scoreboard state is "maybe pending") -- and that waits for the previous STORE
    if (pass == 0 || (RES2 && p.has_res == 3 && pass >= 2)) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt / lgkmcnt untouched
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pl = it * PSTEP + pl0;
      const int pix = pix0 + pass * PP * 16 + pl;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32 + 16);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (!plain) {     // wave-uniform: the BatchNorm-followed convs (no bias / residual / activation / pad channels) skip all of it
        if (bias_ep) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] += bb0[r];
            v[4 + r] += bb1[r];
          }
        }
        if (RES2 && p.has_res == 3) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float r0, r1, m0, m1;
            unpack2<T>(rv[pass][it][e], r0, r1);
            unpack2<T>(rm[RES2 && NPASS > 1 ? (pass & 1) : 0][RES2 ? it : 0][e], m0, m1);
            v[2 * e] = cgan_res_apply3(v[2 * e], r0, m0);
            v[2 * e + 1] = cgan_res_apply3(v[2 * e + 1], r1, m1);
          }
        } else if (p.has_res) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float r0, r1;
            unpack2<T>(rv[pass][it][e], r0, r1);
            v[2 * e] = cgan_res_apply(v[2 * e], r0, p.has_res);
            v[2 * e + 1] = cgan_res_apply(v[2 * e + 1], r1, p.has_res);
          }
        }
        act_apply_n(v, p.act, p.slope);
        if (p.cout < p.cout_s) {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (ch + r >= p.cout) v[r] = 0.f;   // keep pad channels zero
        }
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
      if (pix < p.npix && ch_ok) CGAN_ST_STREAM(o, reinterpret_cast<u32x4*>(p.y + (size_t)pix * p.cout_s + ch));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // staged rows are consumed before the next pass overwrites
  }
}
#endif
