// conv2d NHWC forward as an implicit GEMM on MFMA 16x16x32 (f16 / bf16, fp32 accumulate), gfx950.
//
// GEMM view (per launch):   D[cout][pixel] = sum_k  Wp[cout][k] * X[k][pixel]
//   M = output channels (MFMA A operand = packed weights, rows)          -> 16-row "cout tiles"
//   N = output pixels n*h_out*w_out flattened (MFMA B operand, columns)  -> 16-pixel "pixel tiles"
//   K = taps * cin_s, ordered k = tap * cin_s + c (cin_s = storage channels, multiple of 8)
// Putting channels on M means a lane's 4 accumulator registers are 4 CONSECUTIVE channels of one pixel
// (D layout: col = lane&15, row = 4*(lane>>4)+r), i.e. one 8-byte NHWC store, and the epilogue (bias,
// residual, activation) works on contiguous channel vectors.
//
// Operand fetch: an 8-element K group (lane>>4 selects it) never straddles a tap because cin_s % 8 == 0,
// so every B fragment is ONE 16-byte NHWC load of 8 channels of the tap-shifted input pixel (zero / reflect
// padding and the folded x2 nearest upsample are pure index math).  A fragments are pre-packed in fragment
// order, so a wave reads 1 KiB contiguous per (cout tile, k-step).
//
// This is the general kernel (any kh/kw/stride/dilation/pad mode); operands come through L1/L2.
#include "cgan_common.h"
#include "conv3x3_lds.h"
#include "conv_gemm.h"

namespace {

struct ConvParams {
  const uint16_t* x;
  const u32x4* w;
  const float* bias;
  const uint16_t* res;
  const uint16_t* res2;   // has_res == 3 only (cgan_res_apply3): the ReLU output that masks the result
  uint16_t* y;
  int n, h_in, w_in, hx, wx, cin_s, cin_p, cg;   // cin_p: per-tap K extent (cin_s, or padded to 32 for 3x3)
  int cout, cout_s, ctiles;
  int kh, kw, stride, pad, dil, pad_mode;
  int h_out, w_out, npix;
  int kgroups, ksteps;
  int in_ups, act, has_res, res_ups;
  int in_zs;   // > 1: x is read through zero insertion (stride of the forward conv whose data gradient this is)
  int cls_s;   // > 1: data gradient of a stride-cls_s conv by output parity classes (blockIdx.z), see cls_axis
  int cls_pad; // pad' = k - 1 - pad of that data gradient
  float slope;
  float* stats; // optional per-chunk (mean, M2) output of the wide-layer GEMM kernel (cgan_conv2d_nhwc_fwd_stats)
  int pair;     // 1: split-precision conv (cgan_conv2d_nhwc_fwd_pair): x, y and the residual are split maps of Split<T>::NS
                // blocks per pixel (a value carried in two / three 16-bit numbers); the K extent cin_s = NB * csb
  int xs, csb;  // pixel stride of x in elements (cin_s; split maps: NS * csb) and channels per block of a split map
};

// Parity-class decomposition of the data gradient of a stride-s convolution (dilation 1).  dx[y] = sum over the
// flipped taps ky' with (y - pad' + ky') divisible by s of dy[(y - pad' + ky') / s] w'[ky'].  All outputs y = a (mod s)
// use the same taps ky' = r + s t, t < nt, and read dy at consecutive rows base + t: a dense stride-1 convolution with
// nt taps per axis instead of a k-tap convolution over a zero-inserted map that is (s^2 - 1) / s^2 zeros.
__host__ __device__ inline void cls_axis(int s, int k, int padp, int a, int& r, int& nt) {
  r = ((padp - a) % s + s) % s;
  nt = r < k ? (k - r + s - 1) / s : 0;
}
__host__ __device__ inline int cls_ksteps(int s, int kh, int kw, int padp, int cg, int cls) {
  int r, nty, ntx;
  cls_axis(s, kh, padp, cls / s, r, nty);
  cls_axis(s, kw, padp, cls % s, r, ntx);
  return (nty * ntx * cg + 3) / 4;
}

// K layout of the packed weights: k = tap * cin_p + c.  3x3 kernels pad the per-tap channel extent to a multiple
// of 32 (one MFMA k-step never straddles a tap), which is what the LDS-tiled 3x3 kernel (conv3x3_lds.hip) needs;
// every other kernel size is dense (cin_p = cin_s).
__host__ __device__ inline int conv_cin_p(int cin_s, int kh, int kw) {
  return (kh == 3 && kw == 3) ? ((cin_s + 31) & ~31) : cin_s;
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
  // nn.ReflectionPad2d: -1 -> 1, n -> n-2 (pad < n guaranteed by the host check)
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

template <typename T, int CT, int PT, bool PIPE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvParams p) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15;
  const int g = lane >> 4;

  // ---- parity-class mode (strided data gradient): this workgroup's class, its taps and its block of the weights
  int kw_eff = p.kw, ksteps = p.ksteps, kgroups = p.kgroups, npix = p.npix, wc = p.w_out, hc = p.h_out;
  int ca = 0, cb = 0, ry = 0, rx = 0;
  const u32x4* wbase = p.w;
  if (p.cls_s) {
    const int cs_ = p.cls_s, cls = blockIdx.z;
    ca = cls / cs_;
    cb = cls - ca * cs_;
    int koff = 0;
    for (int c = 0; c < cls; ++c) koff += cls_ksteps(cs_, p.kh, p.kw, p.cls_pad, p.cg, c);
    int nty, ntx;
    cls_axis(cs_, p.kh, p.cls_pad, ca, ry, nty);
    cls_axis(cs_, p.kw, p.cls_pad, cb, rx, ntx);
    kw_eff = ntx;
    kgroups = nty * ntx * p.cg;
    ksteps = (kgroups + 3) / 4;
    wbase = p.w + (size_t)p.ctiles * koff * 64;
    hc = p.h_out > ca ? (p.h_out - ca + cs_ - 1) / cs_ : 0;
    wc = p.w_out > cb ? (p.w_out - cb + cs_ - 1) / cs_ : 0;
    npix = p.n * hc * wc;
  }

  // ---- per-lane pixel coordinates for the PT pixel tiles of this wave
  int pn[PT], py0[PT], px0[PT], opix[PT];
  bool pvalid[PT];
  const int ptile0 = (blockIdx.x * 4 + wave) * PT;
  if (ptile0 * 16 >= npix) return;
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    int pix = (ptile0 + t) * 16 + j;
    pvalid[t] = pix < npix;
    int pc = pvalid[t] ? pix : 0;
    int ox = pc % wc;
    int r = pc / wc;
    int oy = r % hc;
    pn[t] = r / hc;
    if (p.cls_s) {
      oy = oy * p.cls_s + ca;
      ox = ox * p.cls_s + cb;
      py0[t] = (oy - p.cls_pad + ry) / p.cls_s;     // exact: first dy row this output row reads
      px0[t] = (ox - p.cls_pad + rx) / p.cls_s;
      opix[t] = pvalid[t] ? (pn[t] * p.h_out + oy) * p.w_out + ox : -1;
    } else {
      py0[t] = oy * p.stride - p.pad;
      px0[t] = ox * p.stride - p.pad;
      opix[t] = pvalid[t] ? pix : -1;
    }
  }

  f32x4 acc[CT][PT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int ctile0 = blockIdx.y * CT;

  // running (tap, channel-group) of this lane's K group: kg = ks*4 + g
  int c8 = g, ky = 0, kx = 0;
  while (c8 >= p.cg) {
    c8 -= p.cg;
    if (++kx == kw_eff) { kx = 0; ++ky; }
  }

  // operand fetch of one k-step (A: packed weights [ctile][ks][lane] x 16 B; B: 8 channels of the tap-shifted
  // input pixel), then advance this lane's K group by 4
  auto fetch = [&](int ks, u32x4 (&a)[CT], u32x4 (&b)[PT]) {
    // (k-steps past the end -- the pipelined loop runs to a multiple of its depth -- fetch NOTHING new: the weight address is
    // clamped and kvalid masks the activation loads, so the extra MFMAs multiply by zero activations)
    const bool kvalid = (ks * 4 + g) < kgroups;
    const int ksc = ks < ksteps ? ks : ksteps - 1;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      int ct = ctile0 + c;
      a[c] = (u32x4){0u, 0u, 0u, 0u};
      if (ct < p.ctiles) a[c] = wbase[((size_t)ct * ksteps + ksc) * 64 + lane];
    }
    const int dy = ky * p.dil, dx = kx * p.dil;
    unsigned coff = (unsigned)(c8 * 8);
    if (p.pair) {      // K-block b of a split map's NB * csb input channels = storage block xcomp(b) (cgan_common.h, Split<T>)
      const int b = (int)(((float)(c8 * 8) + 0.5f) * (1.0f / (float)p.csb));
      coff = (unsigned)(Split<T>::xcomp(b < Split<T>::NB ? b : 0) * p.csb + (c8 * 8 - b * p.csb));
    }
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      int iy = py0[t] + dy, ix = px0[t] + dx;
      bool ok = kvalid && pvalid[t] && (c8 * 8 < p.cin_s);
      if (p.pad_mode == CGAN_PAD_REFLECT) {
        iy = reflect_idx(iy, p.h_in);
        ix = reflect_idx(ix, p.w_in);
      } else {
        ok = ok && iy >= 0 && iy < p.h_in && ix >= 0 && ix < p.w_in;
      }
      b[t] = (u32x4){0u, 0u, 0u, 0u};
      if (p.in_zs > 1) {   // transposed conv: only every in_zs-th virtual row / column holds data
        ok = ok && (iy % p.in_zs) == 0 && (ix % p.in_zs) == 0;
        iy /= p.in_zs;
        ix /= p.in_zs;
      }
      if (ok) {
        if (p.in_ups) { iy >>= 1; ix >>= 1; }
        // 32-bit element offset (fill_params checks the tensor has < 2^32 elements): a 64-bit multiply chain here
        // costs more issue slots than the MFMAs it feeds
        const unsigned off = ((unsigned)(pn[t] * p.hx + iy) * (unsigned)p.wx + (unsigned)ix) * (unsigned)p.xs + coff;
        b[t] = *reinterpret_cast<const u32x4*>(p.x + off);
      }
    }
    c8 += 4;
    while (c8 >= p.cg) {
      c8 -= p.cg;
      if (++kx == kw_eff) { kx = 0; ++ky; }
    }
  };
  auto mma = [&](const u32x4 (&a)[CT], const u32x4 (&b)[PT]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int t = 0; t < PT; ++t)
        acc[c][t] = mfma16(as_vec8<T>(a[c]), as_vec8<T>(b[t]), acc[c][t]);
  };

  if (PIPE) {
    // software pipeline, 4 k-steps deep: the loads of k-steps ks+1 .. ks+3 are in flight while the MFMAs of k-step ks
    // execute.  Operands come from L1/L2 (~1 us under load), and a small grid -- few output pixels, long K: the
    // discriminators' 512 -> 512 4x4 layers at 19^2 / 9^2, the Painter's 640-channel blocks at 5^2 - 20^2 -- has nothing
    // else to hide that latency with: with ONE k-step in flight (rounds 1-3) such a layer spent ~0.8 us per k-step, 100-200
    // us for 2-10 us of work.  fetch() advances the lane's (tap, channel group) state, so it is called once per k-step, in
    // order; the slots are compile-time indices after unrolling (registers: 3 (CT + PT) x 4 more than the plain loop, on
    // grids that cannot fill the chip anyway).
    constexpr int D = 4;
    u32x4 ra[D][CT], rb[D][PT];
#pragma unroll
    for (int i = 0; i < D - 1; ++i) fetch(i, ra[i], rb[i]);
    // branch-free body (a fetch under a run-time condition makes the compiler wait for ALL outstanding loads at the join)
    for (int ks = 0; ks < ksteps; ks += D) {
#pragma unroll
      for (int jj = 0; jj < D; ++jj) {
        fetch(ks + jj + D - 1, ra[(jj + D - 1) % D], rb[(jj + D - 1) % D]);
        mma(ra[jj], rb[jj]);
      }
    }
  } else {
    for (int ks = 0; ks < ksteps; ++ks) {
      u32x4 a[CT], b[PT];
      fetch(ks, a, b);
      mma(a, b);
    }
  }

  // ---- epilogue: lane holds channels ct*16 + 4g + {0..3} of pixel (tile, j).  The bias quad of a cout tile is loaded once
  // (not per pixel tile); the pad-channel test runs only when the layer has pad channels (wave-uniform).
  const bool pad_c = p.cout < p.cout_s;
  f32x4 bias_q[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int ch = (ctile0 + c) * 16 + g * 4;       // (the bias vector is padded to whole cout tiles)
    bias_q[c] = (p.bias && ch < p.cout_s) ? *reinterpret_cast<const f32x4*>(p.bias + ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int pix = opix[t];
    if (pix < 0) continue;
    size_t rbase = 0;
    if (p.has_res) {
      if (p.res_ups) {
        int ox = pix % p.w_out;
        int r = pix / p.w_out;
        int oy = r % p.h_out;
        int nn = r / p.h_out;
        rbase = (((size_t)nn * (p.h_out >> 1) + (oy >> 1)) * (p.w_out >> 1) + (ox >> 1)) * p.cout_s;
      } else {
        rbase = (size_t)pix * p.cout_s;
      }
    }
    if (p.pair) {
      // split-precision epilogue: bias / residual / activation in fp32, then v -> its 16-bit components (c0 = round16(v),
      // c1 = round16(v - c0), ...), stored as the channel blocks the next conv multiplies by the matching weight blocks
      // (cgan_common.h, Split<T>)
      constexpr int NS = Split<T>::NS, NC = Split<T>::NC;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int ch = (ctile0 + c) * 16 + g * 4;
        if (ch >= p.cout_s) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[c][t][r] + bias_q[c][r];
        if (p.has_res) {
          const uint16_t* rp = p.res + rbase * NS + ch;
          float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = NC - 1; k >= 0; --k) {
            const u32x2 rv = *reinterpret_cast<const u32x2*>(rp + k * p.cout_s);
            float a0, a1, a2, a3;
            unpack2<T>(rv[0], a0, a1);
            unpack2<T>(rv[1], a2, a3);
            rs[0] += a0; rs[1] += a1; rs[2] += a2; rs[3] += a3;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rs[r];
        }
        act_apply_n(v, p.act, p.slope);
        if (pad_c) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (ch + r >= p.cout) v[r] = 0.f;
        }
        u32x2 comp[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
          comp[k][0] = pack2<T>(v[0], v[1]);
          comp[k][1] = pack2<T>(v[2], v[3]);
          float q0, q1, q2, q3;
          unpack2<T>(comp[k][0], q0, q1);
          unpack2<T>(comp[k][1], q2, q3);
          v[0] -= q0; v[1] -= q1; v[2] -= q2; v[3] -= q3;
        }
        uint16_t* yp = p.y + (size_t)pix * p.cout_s * NS + ch;
#pragma unroll
        for (int b = 0; b < NS; ++b) *reinterpret_cast<u32x2*>(yp + b * p.cout_s) = comp[b];
      }
      continue;
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      int ch = (ctile0 + c) * 16 + g * 4;
      if (ch >= p.cout_s) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[c][t][r] + bias_q[c][r];
      if (p.has_res) {
        u32x2 rv = *reinterpret_cast<const u32x2*>(p.res + rbase + ch);
        float r0, r1, r2, r3;
        unpack2<T>(rv[0], r0, r1);
        unpack2<T>(rv[1], r2, r3);
        v[0] = cgan_res_apply(v[0], r0, p.has_res); v[1] = cgan_res_apply(v[1], r1, p.has_res);
        v[2] = cgan_res_apply(v[2], r2, p.has_res); v[3] = cgan_res_apply(v[3], r3, p.has_res);
      }
      act_apply_n(v, p.act, p.slope);
      if (pad_c) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (ch + r >= p.cout) v[r] = 0.f;  // keep pad channels zero
      }
      u32x2 o;
      o[0] = pack2<T>(v[0], v[1]);
      o[1] = pack2<T>(v[2], v[3]);
      *reinterpret_cast<u32x2*>(p.y + (size_t)pix * p.cout_s + ch) = o;
    }
  }
}

// ---- weight packing: fp32 OIHW (optionally / sigma) -> 16-bit fragments [ctile][ks][lane][8]
template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                        const float* __restrict__ sigma, uint16_t* __restrict__ packed,
                                        float* __restrict__ bias_out, int cout, int cin, int cin_p, int kh, int kw,
                                        int ctiles, int ksteps, int tr) {
  // tr = 1: pack the data-gradient operator of the OIHW weight w[cin][cout][kh][kw] of the FORWARD conv (rows =
  // forward input channels, K channels = forward output channels, taps flipped); cout/cin here are the rows / K
  // channels of the packed operator in both modes.
  const int total = ctiles * ksteps * 64;
  const float sig = sigma ? sigma[0] : 1.f;   // w_bar / sigma as a true division (norms.py:112)
  const int taps = kh * kw;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    int lane = idx & 63;
    int ks = (idx >> 6) % ksteps;
    int ct = (idx >> 6) / ksteps;
    int co = ct * 16 + (lane & 15);
    int k0 = ks * 32 + (lane >> 4) * 8;
    uint16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int k = k0 + e;
      int tap = k / cin_p;
      int c = k - tap * cin_p;
      float v = 0.f;
      if (co < cout && tap < taps && c < cin)
        v = __fdiv_rn(tr ? w[((size_t)c * cout + co) * taps + (taps - 1 - tap)] : w[((size_t)co * cin + c) * taps + tap], sig);
      o[e] = bits_of<T>(v);
    }
    u32x4 pk;
    pk[0] = o[0] | ((uint32_t)o[1] << 16);
    pk[1] = o[2] | ((uint32_t)o[3] << 16);
    pk[2] = o[4] | ((uint32_t)o[5] << 16);
    pk[3] = o[6] | ((uint32_t)o[7] << 16);
    reinterpret_cast<u32x4*>(packed)[idx] = pk;
  }
  if (bias_out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ctiles * 16; i += gridDim.x * blockDim.x)
      bias_out[i] = (bias && i < cout) ? bias[i] : 0.f;
  }
}

// batched: blockIdx.y = layer.  A work unit = (16-row cout tile, tap, 32-channel chunk): its 16 x 32 weights are read
// as 16 contiguous runs of 32*taps floats (coalesced: the OIHW rows), transposed through LDS, and written as whole
// 16-byte fragment entries with the cout row varying fastest (256 contiguous bytes per 16 threads).  The first version
// gathered one float per thread from 36-byte-strided addresses (0.36 ms per Painter forward for 42 M weights).
template <typename T>
__global__ __launch_bounds__(256) void pack_conv_weight_batched_kernel(const CganPackItem* __restrict__ items) {
  constexpr int CB = 32;                       // channels per unit
  __shared__ float tile[16][CB * 16 + 17];     // [cout row][c * 16 + tap] (tap passes) or the raw layouts below; + pad: bank spread
  float* tile_flat = &tile[0][0];              // 16 x 529 floats >= 32 x 257 (the data-gradient operator's raw layout)
  const CganPackItem it = items[blockIdx.y];
  // rows / K channels of the packed operator: the forward weight's (c_out, c_in), or -- data-gradient operator -- swapped
  const int n_rows = it.transposed ? it.c_in : it.c_out, n_k = it.transposed ? it.c_out : it.c_in;
  const int cin_p = conv_cin_p((n_k + 7) & ~7, it.kh, it.kw);
  const int cout_s = (n_rows + 7) & ~7;
  const int ctiles = (cout_s + 15) / 16;
  const int taps = it.kh * it.kw;
  const int ksteps = (taps * (cin_p / 8) + 3) / 4;
  const float sig = it.sigma ? it.sigma[0] : 1.f;   // w_bar / sigma as a true division (norms.py:112)
  const int cchunks = (cin_p + CB - 1) / CB;
  const int tap_passes = (taps + 15) / 16;     // 7x7 stems: 49 taps in 4 passes of <= 16
  const int units = ctiles * cchunks * tap_passes;
  u32x4* out = reinterpret_cast<u32x4*>(it.packed);
  for (int u = blockIdx.x; u < units; u += gridDim.x) {
    const int tp = u % tap_passes;
    const int cc = (u / tap_passes) % cchunks;
    const int ct = u / (tap_passes * cchunks);
    const int c0 = cc * CB, t0 = tp * 16;
    const int nt = min(16, taps - t0);         // taps staged in this pass
    __syncthreads();
    const bool plain = it.sigma == nullptr;
    if (taps <= 16) {
      // ---- RAW staging (all layers but the 7x7 stem): the unit's source floats are copied to LDS exactly as they lie in memory
      // (16-byte loads where the rows are 16-byte aligned, no per-element (channel, tap) bookkeeping), and the store phase
      // below picks element (row, channel, tap) out of the raw rows.  Forward operator: 16 rows (couts) of up to CB * taps
      // contiguous floats, raw[row][c_local * taps + tap]; data-gradient operator: per K channel 16 * taps contiguous floats,
      // raw[c_local][row * taps + forward tap].  (The first staged version walked (channel, tap) per element with carries
      // and 4-byte loads: the full pack of the generator's 105 M weights took 340-450 us against 140 us of HBM time.)
      const int nc = min(CB, n_k - c0);                       // K channels of this unit that exist
      if (!it.transposed) {
        constexpr int PITCH = CB * 16 + 1;
        const int row = threadIdx.x >> 4, l = threadIdx.x & 15;
        const int co = ct * 16 + row;
        const int len = nc * taps;
        if (co < it.c_out && nc > 0) {
          const float* src = it.w_oihw + ((size_t)co * it.c_in + c0) * taps;
          float* dst = tile_flat + row * PITCH;
          if (((((size_t)co * it.c_in + c0) * taps) & 3) == 0 && (reinterpret_cast<size_t>(it.w_oihw) & 15) == 0) {
            const int len4 = len & ~3;
            for (int k = l * 4; k < len4; k += 64) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(src + k);
              dst[k] = v[0]; dst[k + 1] = v[1]; dst[k + 2] = v[2]; dst[k + 3] = v[3];
            }
            for (int k = len4 + l; k < len; k += 16) dst[k] = src[k];
          } else {
            for (int k = l; k < len; k += 16) dst[k] = src[k];
          }
        }
      } else {
        constexpr int PITCH = 16 * 16 + 1;
        const int c = threadIdx.x >> 3, l = threadIdx.x & 7;          // 32 K channels x 8 lanes
        const int k = c0 + c;
        const int nr = min(16, n_rows - ct * 16);                     // rows of this tile that exist
        const int len = nr * taps;
        if (k < n_k && nr > 0) {
          const float* src = it.w_oihw + ((size_t)k * n_rows + ct * 16) * taps;
          float* dst = tile_flat + c * PITCH;
          if (((((size_t)k * n_rows + ct * 16) * taps) & 3) == 0 && (reinterpret_cast<size_t>(it.w_oihw) & 15) == 0) {
            const int len4 = len & ~3;
            for (int q = l * 4; q < len4; q += 32) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(src + q);
              dst[q] = v[0]; dst[q + 1] = v[1]; dst[q + 2] = v[2]; dst[q + 3] = v[3];
            }
            for (int q = len4 + l; q < len; q += 8) dst[q] = src[q];
          } else {
            for (int q = l; q < len; q += 8) dst[q] = src[q];
          }
        }
      }
      __syncthreads();
      const int groups = min(CB, cin_p - c0) / 8;
      for (int idx = threadIdx.x; idx < taps * groups * 16; idx += blockDim.x) {
        const int row = idx & 15;
        const int gq = (idx >> 4) % groups;
        const int tap = (idx >> 4) / groups;
        const int k = tap * cin_p + c0 + gq * 8;
        const int ks = k >> 5, g = (k & 31) >> 3;
        const bool row_ok = ct * 16 + row < n_rows;
        uint16_t o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int cl = gq * 8 + e;
          float v = 0.f;
          if (row_ok && c0 + cl < n_k) {
            v = it.transposed ? tile_flat[cl * (16 * 16 + 1) + row * taps + (taps - 1 - tap)]
                              : tile_flat[row * (CB * 16 + 1) + cl * taps + tap];
            if (!plain) v = __fdiv_rn(v, sig);
          }
          o[e] = bits_of<T>(v);
        }
        u32x4 pk;
        pk[0] = o[0] | ((uint32_t)o[1] << 16);
        pk[1] = o[2] | ((uint32_t)o[3] << 16);
        pk[2] = o[4] | ((uint32_t)o[5] << 16);
        pk[3] = o[6] | ((uint32_t)o[7] << 16);
        out[((size_t)ct * ksteps + ks) * 64 + g * 16 + row] = pk;
      }
    } else {
    // ---- load (more than 16 taps, staged in passes of 16): rows of the OIHW tensor, contiguous in (c, tap).  (c, tap) of a thread's elements advance by constants
    // with a carry -- the first version divided twice per element and always divided by sigma: the pack of the
    // generator's 105 M weights ran at a quarter of the HBM rate on instruction count alone
    if (!it.transposed) {
      // thread = (row, 16 lanes along the row's CB * taps contiguous floats)
      const int row = threadIdx.x >> 4, l = threadIdx.x & 15;
      const int co = ct * 16 + row;
      const int dc = 16 / taps, dt = 16 - dc * taps;
      int c = l / taps, tap = l - c * taps;
      const float* src = it.w_oihw + ((size_t)co * it.c_in + c0) * taps;
      for (int r = l; r < CB * taps; r += 16) {
        if (tap >= t0 && tap < t0 + nt) {
          float v = 0.f;
          if (co < it.c_out && c0 + c < it.c_in) {
            v = src[r];
            if (!plain) v = __fdiv_rn(v, sig);
          }
          tile[row][c * 16 + (tap - t0)] = v;
        }
        tap += dt;
        c += dc;
        if (tap >= taps) { tap -= taps; ++c; }
      }
    } else {
      // operator element (row r, K channel k, tap t) = w[k][r][taps - 1 - t]: for one k the 16 rows x taps of a unit are
      // contiguous in the forward tensor.  thread = (k, 8 lanes along those 16 * taps floats)
      const int c = threadIdx.x >> 3, l = threadIdx.x & 7;
      const int k = c0 + c;
      const int dr = 8 / taps, dt = 8 - dr * taps;
      int row = l / taps, ft = l - row * taps;            // ft: tap index in the forward tensor (flipped below)
      const float* src = it.w_oihw + ((size_t)k * n_rows + ct * 16) * taps;
      for (int r = l; r < 16 * taps; r += 8) {
        const int tap = taps - 1 - ft;
        if (tap >= t0 && tap < t0 + nt) {
          float v = 0.f;
          if (ct * 16 + row < n_rows && k < n_k) {
            v = src[r];
            if (!plain) v = __fdiv_rn(v, sig);
          }
          tile[row][c * 16 + (tap - t0)] = v;
        }
        ft += dt;
        row += dr;
        if (ft >= taps) { ft -= taps; ++row; }
      }
    }
    __syncthreads();
    // ---- store: one 16-byte fragment entry per (tap, 8-channel group, row)
    const int groups = min(CB, cin_p - c0) / 8;
    for (int idx = threadIdx.x; idx < nt * groups * 16; idx += blockDim.x) {
      const int row = idx & 15;
      const int gq = (idx >> 4) % groups;
      const int tl = (idx >> 4) / groups;
      const int k = (t0 + tl) * cin_p + c0 + gq * 8;
      const int ks = k >> 5, g = (k & 31) >> 3;
      uint16_t o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = bits_of<T>(tile[row][(gq * 8 + e) * 16 + tl]);
      u32x4 pk;
      pk[0] = o[0] | ((uint32_t)o[1] << 16);
      pk[1] = o[2] | ((uint32_t)o[3] << 16);
      pk[2] = o[4] | ((uint32_t)o[5] << 16);
      pk[3] = o[6] | ((uint32_t)o[7] << 16);
      out[((size_t)ct * ksteps + ks) * 64 + g * 16 + row] = pk;
    }
    }
    // ---- the K groups past taps * cin_p that pad the last k-step are zero
    if (cc == 0 && tp == 0) {
      const int kg_used = taps * (cin_p / 8), kg_all = ksteps * 4;
      for (int idx = threadIdx.x; idx < (kg_all - kg_used) * 16; idx += blockDim.x) {
        const int row = idx & 15, kg = kg_used + (idx >> 4);
        out[((size_t)ct * ksteps + (kg >> 2)) * 64 + (kg & 3) * 16 + row] = (u32x4){0u, 0u, 0u, 0u};
      }
    }
  }
  if (it.bias_out)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ctiles * 16; i += gridDim.x * blockDim.x)
      it.bias_out[i] = (it.bias && i < it.c_out) ? it.bias[i] : 0.f;
}

// Packed weights of the parity-class data gradient: one [ctile][ks][lane][8] block per class (blockIdx.y), K order
// (t_y, t_x, channel) over the class's taps ky' = ry + s t_y, kx' = rx + s t_x of the flipped, channel-transposed
// operator.  w is the FORWARD OIHW weight [c_out_fwd = K channels][c_in_fwd = rows][kh][kw].
template <typename T>
__global__ void pack_dgrad_classes_kernel(const float* __restrict__ w, const float* __restrict__ sigma,
                                          uint16_t* __restrict__ packed, int rows, int kch, int cin_p, int kh, int kw,
                                          int ctiles, int cs_, int padp) {
  const int cls = blockIdx.y, cg = cin_p / 8;
  int koff = 0;
  for (int c = 0; c < cls; ++c) koff += cls_ksteps(cs_, kh, kw, padp, cg, c);
  int ry, rx, nty, ntx;
  cls_axis(cs_, kh, padp, cls / cs_, ry, nty);
  cls_axis(cs_, kw, padp, cls % cs_, rx, ntx);
  const int ksteps = (nty * ntx * cg + 3) / 4, taps = kh * kw;
  const float sig = sigma ? sigma[0] : 1.f;
  u32x4* out = reinterpret_cast<u32x4*>(packed) + (size_t)ctiles * koff * 64;
  const int total = ctiles * ksteps * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, ks = (idx >> 6) % ksteps, ct = (idx >> 6) / ksteps;
    const int row = ct * 16 + (lane & 15);
    const int k0 = ks * 32 + (lane >> 4) * 8;
    uint16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e, tp = k / cin_p, c = k - tp * cin_p;
      float v = 0.f;
      if (row < rows && tp < nty * ntx && c < kch) {
        const int ty = tp / ntx, tx = tp - ty * ntx;
        const int tap = (ry + cs_ * ty) * kw + rx + cs_ * tx;          // flipped tap index
        v = __fdiv_rn(w[((size_t)c * rows + row) * taps + (taps - 1 - tap)], sig);
      }
      o[e] = bits_of<T>(v);
    }
    u32x4 pk;
    pk[0] = o[0] | ((uint32_t)o[1] << 16);
    pk[1] = o[2] | ((uint32_t)o[3] << 16);
    pk[2] = o[4] | ((uint32_t)o[5] << 16);
    pk[3] = o[6] | ((uint32_t)o[7] << 16);
    out[idx] = pk;
  }
}

int fill_params(ConvParams& p, const CganConvDesc* d) {
  CGAN_REQUIRE(d != nullptr, "conv2d: null descriptor");
  CGAN_REQUIRE(d->dtype == CGAN_F16 || d->dtype == CGAN_BF16, "conv2d: bad dtype %d", d->dtype);
  CGAN_REQUIRE(d->n > 0 && d->h_in > 0 && d->w_in > 0 && d->c_in > 0 && d->c_out > 0, "conv2d: bad shape");
  CGAN_REQUIRE(d->kh > 0 && d->kw > 0 && d->stride > 0 && d->dilation > 0 && d->pad >= 0, "conv2d: bad kernel params");
  int eh = (d->h_in + 2 * d->pad - d->dilation * (d->kh - 1) - 1) / d->stride + 1;
  int ew = (d->w_in + 2 * d->pad - d->dilation * (d->kw - 1) - 1) / d->stride + 1;
  CGAN_REQUIRE(eh == d->h_out && ew == d->w_out, "conv2d: h_out/w_out (%d,%d) inconsistent, expected (%d,%d)",
               d->h_out, d->w_out, eh, ew);
  CGAN_REQUIRE(d->pad_mode == CGAN_PAD_ZERO || d->pad_mode == CGAN_PAD_REFLECT, "conv2d: Unsupported padding type: %d",
               d->pad_mode);
  if (d->pad_mode == CGAN_PAD_REFLECT)
    CGAN_REQUIRE(d->pad < d->h_in && d->pad < d->w_in, "conv2d: reflect padding %d must be < input size", d->pad);
  if (d->in_upsample) CGAN_REQUIRE((d->h_in % 2) == 0 && (d->w_in % 2) == 0, "conv2d: in_upsample needs even h_in/w_in");
  if (d->has_residual && d->residual_upsample)
    CGAN_REQUIRE((d->h_out % 2) == 0 && (d->w_out % 2) == 0, "conv2d: residual_upsample needs even h_out/w_out");
  CGAN_REQUIRE(d->act >= CGAN_ACT_NONE && d->act <= CGAN_ACT_SIGMOID, "conv2d: Unsupported activation: %d", d->act);
  p.n = d->n; p.h_in = d->h_in; p.w_in = d->w_in;
  p.hx = d->in_upsample ? d->h_in / 2 : d->h_in;
  p.wx = d->in_upsample ? d->w_in / 2 : d->w_in;
  p.cin_s = cgan_cs(d->c_in); p.cin_p = conv_cin_p(p.cin_s, d->kh, d->kw); p.cg = p.cin_p / 8;
  p.xs = p.cin_s; p.csb = p.cin_s;
  p.cout = d->c_out; p.cout_s = cgan_cs(d->c_out); p.ctiles = ceil_div(p.cout_s, 16);
  p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad; p.dil = d->dilation; p.pad_mode = d->pad_mode;
  p.h_out = d->h_out; p.w_out = d->w_out;
  long npix = (long)d->n * d->h_out * d->w_out;
  CGAN_REQUIRE(npix < (1L << 31) - 64, "conv2d: too many output pixels");
  CGAN_REQUIRE((double)d->n * d->h_in * d->w_in * cgan_cs(d->c_in) < 4294967296.0, "conv2d: input tensor has 2^32 elements or more");
  p.npix = (int)npix;
  p.kgroups = d->kh * d->kw * p.cg; p.ksteps = ceil_div(p.kgroups, 4);
  p.in_ups = d->in_upsample; p.act = d->act; p.slope = d->act_slope; p.in_zs = 1; p.cls_s = 0; p.cls_pad = 0;
  p.has_res = d->has_residual; p.res_ups = d->residual_upsample;
  p.stats = nullptr;
  p.res2 = nullptr;
  p.pair = 0;
  return CGAN_OK;
}

// pixels the grid has to cover: all of them, or (parity-class mode) those of the largest class, per blockIdx.z
int grid_pixels(const ConvParams& p) {
  if (!p.cls_s) return p.npix;
  return p.n * ceil_div(p.h_out, p.cls_s) * ceil_div(p.w_out, p.cls_s);
}

template <typename T, int CT>
void launch_ct(const ConvParams& p, hipStream_t s) {
  const int ptiles = ceil_div(grid_pixels(p), 16);
  const int gy = ceil_div(p.ctiles, CT);
  const int gz = p.cls_s ? p.cls_s * p.cls_s : 1;
  // enough blocks to fill 256 CUs: shrink the per-wave pixel register tile for small problems
  if ((long)ceil_div(ptiles, 16) * gy * gz >= 512) {
    hipLaunchKernelGGL((conv_mfma_kernel<T, CT, 4, false>), dim3(ceil_div(ptiles, 16), gy, gz), dim3(256), 0, s, p);
  } else if ((long)ceil_div(ptiles, 8) * gy * gz >= 512) {
    hipLaunchKernelGGL((conv_mfma_kernel<T, CT, 2, false>), dim3(ceil_div(ptiles, 8), gy, gz), dim3(256), 0, s, p);
  } else if ((long)ceil_div(ptiles, 4) * gy * gz >= 2048) {
    hipLaunchKernelGGL((conv_mfma_kernel<T, CT, 1, false>), dim3(ceil_div(ptiles, 4), gy, gz), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((conv_mfma_kernel<T, CT, 1, true>), dim3(ceil_div(ptiles, 4), gy, gz), dim3(256), 0, s, p);
  }
}

template <typename T>
void launch(const ConvParams& p, hipStream_t s) {
  // channel tiles per workgroup: as many as divide ctiles (more reuse of the gathered B fragments), unless the
  // grid would then be too small to fill the chip (few pixels, many channels: the 5x5..20x20 Painter layers)
  const long pblocks = (long)ceil_div(ceil_div(grid_pixels(p), 16), 4) * (p.cls_s ? p.cls_s * p.cls_s : 1);
  int ct = 1;
  if (p.ctiles >= 4 && p.ctiles % 4 == 0) ct = 4;
  else if (p.ctiles % 3 == 0) ct = 3;
  else if (p.ctiles % 2 == 0) ct = 2;
  else if (p.ctiles == 5) ct = 3;
  else if (p.ctiles > 1) ct = 4;
  while (ct > 1 && pblocks * ceil_div(p.ctiles, ct) < 512) --ct;
  switch (ct) {
    case 4: launch_ct<T, 4>(p, s); break;
    case 3: launch_ct<T, 3>(p, s); break;
    case 2: launch_ct<T, 2>(p, s); break;
    default: launch_ct<T, 1>(p, s); break;
  }
}

}  // namespace

// development knob: 1 = always use the general gather kernel, 2 = never use the wide-layer GEMM kernel, 3 = the GEMM
// kernel also where the spatially tiled 3x3 kernel would be preferred, 4 = automatic without round 5's parity-class /
// split-K launches of the LDS-tiled GEMM (A/B measurements, parity tests of every kernel)
CGAN_KNOB(int, g_conv_force, 0);
CGAN_DEV_ONLY(extern "C" void cgan_debug_set_conv_kernel(int v) { g_conv_force = v; })
// 1: strided data gradients read dy through zero insertion (the first implementation) instead of by parity classes
CGAN_KNOB(int, g_dgrad_zero_insert, 0);
CGAN_DEV_ONLY(extern "C" void cgan_debug_set_dgrad_zero_insert(int v) { g_dgrad_zero_insert = v; })

extern "C" size_t cgan_conv2d_packed_weight_bytes(const CganConvDesc* d) {
  ConvParams p;
  if (fill_params(p, d) != CGAN_OK) return 0;
  return (size_t)p.ctiles * p.ksteps * 64 * 16;
}

extern "C" int cgan_conv2d_pack_weight(const float* w_oihw, const float* bias, const float* sigma, void* packed,
                                       float* bias_out, const CganConvDesc* d, void* stream) {
  ConvParams p;
  int rc = fill_params(p, d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(w_oihw && packed, "conv2d_pack_weight: null pointer");
  const int total = p.ctiles * p.ksteps * 64;
  const int blocks = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == CGAN_F16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<F16>, dim3(blocks), dim3(256), 0, s, w_oihw, bias, sigma,
                       (uint16_t*)packed, bias_out, d->c_out, d->c_in, p.cin_p, d->kh, d->kw, p.ctiles, p.ksteps, 0);
  else
    hipLaunchKernelGGL(pack_conv_weight_kernel<BF16>, dim3(blocks), dim3(256), 0, s, w_oihw, bias, sigma,
                       (uint16_t*)packed, bias_out, d->c_out, d->c_in, p.cin_p, d->kh, d->kw, p.ctiles, p.ksteps, 0);
  CGAN_CHECK_LAUNCH("conv2d_pack_weight");
  return CGAN_OK;
}

extern "C" int cgan_conv2d_pack_weight_batched(const CganPackItem* items_device, int32_t count, int32_t dtype,
                                               int32_t max_fragments, void* stream) {
  CGAN_REQUIRE(items_device && count > 0 && max_fragments > 0, "conv2d_pack_weight_batched: bad arguments");
  CGAN_REQUIRE(dtype == CGAN_F16 || dtype == CGAN_BF16, "conv2d_pack_weight_batched: bad dtype %d", dtype);
  const int blocks = ceil_div(max_fragments, 256) < 512 ? ceil_div(max_fragments, 256) : 512;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(pack_conv_weight_batched_kernel<F16>, dim3(blocks, count), dim3(256), 0, s, items_device);
  else
    hipLaunchKernelGGL(pack_conv_weight_batched_kernel<BF16>, dim3(blocks, count), dim3(256), 0, s, items_device);
  CGAN_CHECK_LAUNCH("conv2d_pack_weight_batched");
  return CGAN_OK;
}

// ------------------------------------------------------------------------------------------------
// split-K workspaces (round 5): one caller-owned fp32 scratch buffer per stream, registered once
// (cgan_conv2d_bind_workspace).  Launches on one stream are ordered, so consecutive convolutions share their stream's buffer;
// two streams never share one.  Without a binding (or with one that is too small) the small-grid layers stay on the general
// kernel -- slower, same results up to the fp32 summation order.
// ------------------------------------------------------------------------------------------------
#include <mutex>
namespace {
// keyed by (device, stream): the NULL stream has the same handle on every device of a process
struct WsSlot { int device; void* stream; void* ws; size_t bytes; };
constexpr int WS_SLOTS = 64;
WsSlot g_ws[WS_SLOTS];
int g_ws_n = 0;
int g_ws_next = 0;      // replacement cursor once the table is full (oldest binding first)
std::mutex g_ws_mu;
int ws_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return dev;
}
WsSlot ws_lookup(void* stream) {
  const int dev = ws_device();
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (int i = 0; i < g_ws_n; ++i)
    if (g_ws[i].stream == stream && g_ws[i].device == dev) return g_ws[i];
  return WsSlot{dev, stream, nullptr, 0};
}
size_t ws_max_bytes() {       // the largest workspace bound on the current device
  const int dev = ws_device();
  std::lock_guard<std::mutex> lk(g_ws_mu);
  size_t m = 0;
  for (int i = 0; i < g_ws_n; ++i)
    if (g_ws[i].device == dev && g_ws[i].bytes > m) m = g_ws[i].bytes;
  return m;
}
}  // namespace

// workspace == NULL, bytes == 0 removes the binding of (current device, stream).  A full table replaces its OLDEST binding
// (the caller that bound it loses split-K on that stream -- the general kernel runs instead -- nothing fails).
extern "C" int cgan_conv2d_bind_workspace(void* stream, void* workspace, size_t bytes) {
  CGAN_REQUIRE((workspace != nullptr) == (bytes > 0), "conv2d_bind_workspace: workspace and bytes must both be set or both be zero");
  CGAN_REQUIRE((reinterpret_cast<size_t>(workspace) & 15) == 0, "conv2d_bind_workspace: workspace must be 16-byte aligned");
  const int dev = ws_device();
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (int i = 0; i < g_ws_n; ++i)
    if (g_ws[i].stream == stream && g_ws[i].device == dev) {
      if (workspace == nullptr) {               // unbind: close the gap
        g_ws[i] = g_ws[g_ws_n - 1];
        --g_ws_n;
        if (g_ws_next >= g_ws_n) g_ws_next = 0;
      } else {
        g_ws[i].ws = workspace; g_ws[i].bytes = bytes;
      }
      return CGAN_OK;
    }
  if (workspace == nullptr) return CGAN_OK;     // nothing bound: nothing to remove
  if (g_ws_n < WS_SLOTS) {
    g_ws[g_ws_n++] = WsSlot{dev, stream, workspace, bytes};
  } else {
    g_ws[g_ws_next] = WsSlot{dev, stream, workspace, bytes};
    g_ws_next = (g_ws_next + 1) % WS_SLOTS;
  }
  return CGAN_OK;
}

static ConvGemmArgs gemm_args(const ConvParams& p) {
  ConvGemmArgs a;
  a.x = p.x; a.w = p.w; a.bias = p.bias; a.res = p.res; a.res2 = p.res2; a.y = p.y;
  a.n = p.n; a.h_in = p.h_in; a.w_in = p.w_in; a.cin_s = p.cin_s;
  a.cout = p.cout; a.cout_s = p.cout_s; a.ctiles = p.ctiles; a.ksteps = p.ksteps;
  a.kh = p.kh; a.kw = p.kw; a.stride = p.stride; a.pad = p.pad; a.dil = p.dil; a.pad_mode = p.pad_mode;
  a.h_out = p.h_out; a.w_out = p.w_out; a.npix = p.npix;
  a.act = p.act; a.has_res = p.has_res; a.res_ups = p.res_ups; a.slope = p.slope;
  a.stats = p.stats;
  return a;
}

// K slices if this launch (which the size thresholds keep off the plain wide-layer kernel) runs as a split-K launch of the
// LDS-tiled GEMM with ``ws_bytes`` of workspace; 1 = it does not
static int splitk_for(const ConvParams& p, size_t ws_bytes) {
  if (g_conv_force != 0 || p.in_zs != 1 || p.in_ups || p.cls_s || p.pair || p.stats || p.pad < 0) return 1;
  if (p.cin_p != p.cin_s) return 1;      // (3x3 layers whose per-tap channel extent is padded to 32: cin_s % 32 != 0 anyway)
  const ConvGemmArgs a = gemm_args(p);
  const int ks = conv_gemm_splitk_plan(a);
  if (ks <= 1 || conv_gemm_splitk_workspace_bytes(a, ks) > ws_bytes) return 1;
  return ks;
}

// launches the selection below leaves on the general kernel, or on the spatially tiled 3x3 kernel only because the wide-layer
// kernel's size thresholds refused them (>= 256 input channels: the tiled kernel is not the preferred one there)
static bool splitk_candidate(int kind, const ConvParams& p) {
  return kind == CGAN_CONV_KERNEL_GENERAL || (kind == CGAN_CONV_KERNEL_LDS3X3 && p.cin_s >= 256);
}

// kernel selection shared by the forward and the stride-1 data-gradient entry points
static int select_conv_kernel(const ConvParams& p, const CganConvDesc* d) {
  // narrow 3x3 / stride-1 layers (< 256 channels in) are faster in the spatially tiled 3x3 kernel (halo reuse in LDS)
  const bool prefer_3x3 = conv3x3_lds_applicable(d) && p.cin_s < 256 && g_conv_force != 3;
  if ((g_conv_force == 0 || g_conv_force == 3) && p.in_zs == 1 && !prefer_3x3 && conv_gemm_applicable(d)) return CGAN_CONV_KERNEL_GEMM;
  if (g_conv_force != 1 && p.in_zs == 1 && conv3x3_lds_applicable(d)) return CGAN_CONV_KERNEL_LDS3X3;
  // first-layer convs (8 storage channels in, k != 3): the halo-tiled kernel of the same family (conv_smallcin_kernel)
  if (g_conv_force == 0 && p.in_zs == 1 && !p.pair && conv_smallcin_applicable(d)) return CGAN_CONV_KERNEL_LDS3X3;
  return CGAN_CONV_KERNEL_GENERAL;
}

static int dispatch_conv(ConvParams& p, const CganConvDesc* d, hipStream_t s, const char* what) {
  const int kind = select_conv_kernel(p, d);
  if (kind == CGAN_CONV_KERNEL_GEMM) {
    const ConvGemmArgs a = gemm_args(p);
    int rc2 = conv_gemm_launch(a, d->dtype, s);
    if (rc2 != CGAN_OK) return rc2;
    CGAN_CHECK_LAUNCH(what);
    return CGAN_OK;
  }
  if (splitk_candidate(kind, p)) {
    // few output pixels, long K: K slices of the LDS-tiled GEMM + an ordered reduce (conv_gemm_ext.hip)
    const WsSlot slot = ws_lookup((void*)s);
    const int ks = splitk_for(p, slot.bytes);
    if (ks > 1) {
      int rc2 = conv_gemm_splitk_launch(gemm_args(p), ks, (float*)slot.ws, d->dtype, s);
      if (rc2 != CGAN_OK) return rc2;
      CGAN_CHECK_LAUNCH(what);
      return CGAN_OK;
    }
  }
  if (kind == CGAN_CONV_KERNEL_LDS3X3 && !conv3x3_lds_applicable(d)) {        // the first-layer kernel
    Conv3x3LdsArgs a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.res = p.res; a.y = p.y;
    a.n = p.n; a.h = p.h_out; a.w_ = p.w_out; a.hi = p.h_in; a.wi = p.w_in; a.pad = p.pad; a.reflect = 0;
    a.shuffle = 0; a.shuffle_w = 0; a.k = p.kh; a.stride = p.stride;
    a.hx = p.hx; a.wx = p.wx; a.cin = d->c_in; a.cin_s = p.cin_s; a.cin_p = p.cin_p;
    a.cout = p.cout; a.cout_s = p.cout_s; a.ctiles = p.ctiles; a.ksteps = p.ksteps;
    a.in_ups = 0; a.act = p.act; a.has_res = 0; a.res_ups = 0; a.slope = p.slope;
    int rc2 = conv_smallcin_launch(a, d->dtype, s);
    if (rc2 != CGAN_OK) return rc2;
    CGAN_CHECK_LAUNCH(what);
    return CGAN_OK;
  }
  if (kind == CGAN_CONV_KERNEL_LDS3X3) {
    Conv3x3LdsArgs a;
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.res = p.res; a.y = p.y;
    a.n = p.n; a.h = p.h_out; a.w_ = p.w_out; a.hi = p.h_in; a.wi = p.w_in; a.pad = p.pad;
    a.reflect = p.pad_mode == CGAN_PAD_REFLECT;
    a.shuffle = 0; a.shuffle_w = 0; a.k = 0; a.stride = 0;
    a.hx = p.hx; a.wx = p.wx; a.cin = d->c_in; a.cin_s = p.cin_s; a.cin_p = p.cin_p;
    a.cout = p.cout; a.cout_s = p.cout_s; a.ctiles = p.ctiles; a.ksteps = p.ksteps;
    a.in_ups = p.in_ups; a.act = p.act; a.has_res = p.has_res; a.res_ups = p.res_ups; a.slope = p.slope;
    int rc2 = conv3x3_lds_launch(a, d->dtype, s);
    if (rc2 != CGAN_OK) return rc2;
    CGAN_CHECK_LAUNCH(what);
    return CGAN_OK;
  }
  if (d->dtype == CGAN_F16) launch<F16>(p, s);
  else launch<BF16>(p, s);
  CGAN_CHECK_LAUNCH(what);
  return CGAN_OK;
}

extern "C" int cgan_conv2d_nhwc_fwd(const void* x, const void* packed_w, const float* bias_padded, const void* residual,
                                    void* y, const CganConvDesc* d, void* stream) {
  ConvParams p;
  int rc = fill_params(p, d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(x && packed_w && y, "conv2d_nhwc_fwd: null pointer");
  CGAN_REQUIRE(!d->has_bias || bias_padded, "conv2d_nhwc_fwd: has_bias but bias is null");
  CGAN_REQUIRE(!d->has_residual || residual, "conv2d_nhwc_fwd: has_residual but residual is null");
  p.x = (const uint16_t*)x; p.w = (const u32x4*)packed_w; p.bias = d->has_bias ? bias_padded : nullptr;
  p.res = (const uint16_t*)residual; p.y = (uint16_t*)y;
  return dispatch_conv(p, d, (hipStream_t)stream, "conv2d_nhwc_fwd");
}

// Split-precision forward (round 4): see the header.  The general gather kernel, or (round 5) the LDS-tiled GEMM for wide layers.
CGAN_KNOB(int, g_pair_big, 1);      // dev: 0 = never the 256 x 256 tile for split convs (same-box A/B, kernel-variant tests)
CGAN_DEV_ONLY(extern "C" void cgan_debug_set_pair_big(int v) { g_pair_big = v; })
extern "C" int cgan_conv2d_nhwc_fwd_pair(const void* x3, const void* packed_w3, const float* bias_padded,
                                         const void* residual3, void* y3, const CganConvDesc* d, void* stream) {
  ConvParams p;
  int rc = fill_params(p, d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(x3 && packed_w3 && y3, "conv2d_nhwc_fwd_pair: null pointer");
  CGAN_REQUIRE(!d->has_bias || bias_padded, "conv2d_nhwc_fwd_pair: has_bias but bias is null");
  CGAN_REQUIRE(!d->has_residual || residual3, "conv2d_nhwc_fwd_pair: has_residual but residual is null");
  const int nb = cgan_split_blocks(d->dtype);
  CGAN_REQUIRE((d->c_in % (8 * nb)) == 0, "conv2d_nhwc_fwd_pair: c_in must be the %d * round_up(C, 8) storage channels of a split map", nb);
  CGAN_REQUIRE((double)p.npix * p.cout_s * cgan_split_store_blocks(d->dtype) * 2.0 < 4294967295.0, "conv2d_nhwc_fwd_pair: output map of 4 GiB or more");
  p.x = (const uint16_t*)x3; p.w = (const u32x4*)packed_w3; p.bias = d->has_bias ? bias_padded : nullptr;
  p.res = (const uint16_t*)residual3; p.y = (uint16_t*)y3;
  p.pair = 1;
  p.csb = p.cin_s / nb;                                   // the input map stores each component once: NS blocks of csb
  p.xs = cgan_split_store_blocks(d->dtype) * p.csb;
  hipStream_t s = (hipStream_t)stream;
  // wide layers on the LDS-tiled GEMM (round 5): whole 32-channel k-steps per tap, zero padding, no folded upsample, enough
  // pixels for its 128 / 256-pixel block tiles; everything else stays on the gather kernel
  if (g_conv_force == 0 && p.in_zs == 1 && !p.in_ups && p.cin_p == p.cin_s && p.pad >= 0 && p.npix >= 1024 && p.ksteps >= 4) {
    const ConvGemmArgs a = gemm_args(p);
    // >= 192 couts and whole 64-channel K stages: the 256 x 256 / eight-wave tile (round 6: 0.30 -> 0.4+ of MFMA on the
    // ResNet layer3 / layer4 / ASPP convs, which are most of a split-precision Masker)
    if (g_pair_big && conv_gemm_big_pair_ok(a, d->dtype)) {
      rc = conv_gemm_big_pair_launch(a, d->dtype, s);
      if (rc != CGAN_OK) return rc;
      CGAN_CHECK_LAUNCH("conv2d_nhwc_fwd_pair(gemm 256)");
      return CGAN_OK;
    }
    if (conv_gemm_ext_shape_ok(a) && (double)p.npix * p.cout_s * cgan_split_store_blocks(d->dtype) < 2147483647.0) {
      rc = conv_gemm_pair_launch(a, d->dtype, s);
      if (rc != CGAN_OK) return rc;
      CGAN_CHECK_LAUNCH("conv2d_nhwc_fwd_pair(gemm)");
      return CGAN_OK;
    }
  }
  if (d->dtype == CGAN_F16) launch<F16>(p, s);
  else launch<BF16>(p, s);
  CGAN_CHECK_LAUNCH("conv2d_nhwc_fwd_pair");
  return CGAN_OK;
}

// Forward with training-mode BatchNorm statistics from the kernel's epilogue (round 3): see the header.
static const float k_bias_sentinel = 0.f;    // never dereferenced
static int stats_chunk_pixels(ConvParams& p, const CganConvDesc* d) {
  if (d->has_residual || d->act != CGAN_ACT_NONE || d->in_upsample) return 0;
  if (select_conv_kernel(p, d) != CGAN_CONV_KERNEL_GEMM) return 0;
  ConvGemmArgs a;
  a.x = nullptr; a.w = nullptr; a.bias = d->has_bias ? &k_bias_sentinel : nullptr;   // choose() keys on it: the query must pick what the launch picks
  a.res = nullptr; a.res2 = nullptr; a.y = nullptr; a.stats = nullptr;
  a.n = p.n; a.h_in = p.h_in; a.w_in = p.w_in; a.cin_s = p.cin_s;
  a.cout = p.cout; a.cout_s = p.cout_s; a.ctiles = p.ctiles; a.ksteps = p.ksteps;
  a.kh = p.kh; a.kw = p.kw; a.stride = p.stride; a.pad = p.pad; a.dil = p.dil; a.pad_mode = p.pad_mode;
  a.h_out = p.h_out; a.w_out = p.w_out; a.npix = p.npix;
  a.act = p.act; a.has_res = p.has_res; a.res_ups = p.res_ups; a.slope = p.slope;
  return conv_gemm_stats_chunk_pixels(a, d->dtype);
}

extern "C" int32_t cgan_conv2d_stats_chunk_pixels(const CganConvDesc* d) {
  ConvParams p;
  if (fill_params(p, d) != CGAN_OK) return 0;
  return stats_chunk_pixels(p, d);
}

extern "C" int cgan_conv2d_nhwc_fwd_stats(const void* x, const void* packed_w, const float* bias_padded, void* y,
                                          float* partial, size_t partial_bytes, const CganConvDesc* d, void* stream) {
  ConvParams p;
  int rc = fill_params(p, d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(x && packed_w && y && partial, "conv2d_nhwc_fwd_stats: null pointer");
  CGAN_REQUIRE(!d->has_bias || bias_padded, "conv2d_nhwc_fwd_stats: has_bias but bias is null");
  const int ppb = stats_chunk_pixels(p, d);
  CGAN_REQUIRE(ppb > 0, "conv2d_nhwc_fwd_stats: this descriptor's kernel writes no statistics "
                        "(cgan_conv2d_stats_chunk_pixels returned 0)");
  const size_t need = (size_t)(p.npix / ppb) * p.cout_s * 2 * sizeof(float);
  if (partial_bytes < need) {
    cgan_set_error("conv2d_nhwc_fwd_stats: partial buffer %zu B < required %zu B", partial_bytes, need);
    return CGAN_ERR_WORKSPACE;
  }
  p.x = (const uint16_t*)x; p.w = (const u32x4*)packed_w; p.bias = d->has_bias ? bias_padded : nullptr;
  p.res = nullptr; p.y = (uint16_t*)y; p.stats = partial;
  return dispatch_conv(p, d, (hipStream_t)stream, "conv2d_nhwc_fwd_stats");
}

// ------------------------------------------------------------------------------------------------
// backward-data: dx = conv_transpose(dy, w) expressed as a stride-1 convolution of dy (read through zero insertion
// when the forward stride is > 1) with the channel-transposed, tap-flipped weights, pad' = dil (k-1) - pad.
// ------------------------------------------------------------------------------------------------
static int dgrad_desc(const CganConvDesc* f, CganConvDesc* t) {
  CGAN_REQUIRE(f != nullptr, "conv2d bwd_data: null descriptor");
  CGAN_REQUIRE(f->pad_mode == CGAN_PAD_ZERO, "conv2d bwd_data: only zero padding has a backward path");
  CGAN_REQUIRE(f->stride >= 1 && f->dilation >= 1 && f->kh > 0 && f->kw > 0, "conv2d bwd_data: bad kernel params");
  *t = *f;
  t->c_in = f->c_out; t->c_out = f->c_in;
  t->h_in = (f->h_out - 1) * f->stride + 1;      // virtual (zero-inserted) extent of dy
  t->w_in = (f->w_out - 1) * f->stride + 1;
  t->stride = 1;
  t->pad = f->dilation * (f->kh - 1) - f->pad;
  CGAN_REQUIRE(f->kh == f->kw, "conv2d bwd_data: square kernels only");
  t->h_out = t->h_in + 2 * t->pad - f->dilation * (f->kh - 1);
  t->w_out = t->w_in + 2 * t->pad - f->dilation * (f->kw - 1);
  CGAN_REQUIRE(t->h_out <= f->h_in && t->w_out <= f->w_in && t->h_out > 0 && t->w_out > 0,
               "conv2d bwd_data: inconsistent forward descriptor");
  t->in_upsample = 0; t->act = CGAN_ACT_NONE; t->has_bias = 0; t->has_residual = 0; t->residual_upsample = 0;
  return CGAN_OK;
}

static int dgrad_params(ConvParams& p, const CganConvDesc* f, CganConvDesc* t) {
  int rc = dgrad_desc(f, t);
  if (rc != CGAN_OK) return rc;
  const int pad_t = t->pad;
  if (pad_t < 0) t->pad = 0;                       // fill_params wants pad >= 0 (only ASPP's padded 1x1 gets here)
  if (pad_t < 0) { t->h_out = t->h_in; t->w_out = t->w_in; }
  rc = fill_params(p, t);
  if (rc != CGAN_OK) return rc;
  p.pad = pad_t;
  // rows / columns of the forward input that no output window reached (floor in the output-size formula) still get
  // written: every tap lands outside the virtual extent there, so they come out zero
  p.h_out = f->h_in; p.w_out = f->w_in;
  p.npix = f->n * f->h_in * f->w_in;
  p.hx = f->h_out; p.wx = f->w_out;                // stored extent of dy
  p.in_zs = f->stride;
  if (f->stride > 1 && f->dilation == 1 && pad_t >= 0 && g_dgrad_zero_insert == 0) {
    // parity classes: coordinates are those of the stored dy, no zero insertion
    p.cls_s = f->stride; p.cls_pad = pad_t; p.in_zs = 1;
    p.h_in = p.hx; p.w_in = p.wx;
  }
  t->h_out = f->h_in; t->w_out = f->w_in;
  return CGAN_OK;
}

// ------------------------------------------------------------------------------------------------
// Sub-pixel form of the data gradient of a 4x4 / stride-2 / pad-1 convolution with <= 4 input channels (the first layer of
// the PatchGAN / ADVENT discriminators, reference discriminator.py:100-120, whose input gradient the Painter's GAN and
// feature-matching terms need; round 5).  dx[2i + a][2j + b][c] = sum over the 3x3 neighbourhood (ty, tx) of dy[i + ty - 1]
// [j + tx - 1][co] * w[co][c][a + 3 - 2 ty][b + 3 - 2 tx] (taps outside 0..3 do not exist): ONE 3x3 / stride-1 / pad-1
// convolution of dy with 16 output channels (a, b, c) -- a full MFMA row tile instead of the parity classes' 4 live rows of
// 16 -- on the spatially tiled 3x3 kernel, whose epilogue scatters the channels back to pixels (depth to space).  The
// general kernel took 259 us for the 8 x 640 x 640 gradient (HBM time of its 157 MB: ~35 us).
// ------------------------------------------------------------------------------------------------
static bool dgrad_subpixel(const CganConvDesc* f) {
  return g_conv_force == 0 && f->kh == 4 && f->kw == 4 && f->stride == 2 && f->pad == 1 && f->dilation == 1 &&
         f->pad_mode == CGAN_PAD_ZERO && f->c_in <= 4 && !f->in_upsample && (cgan_cs(f->c_out) % 32) == 0 &&
         (f->h_in % 2) == 0 && (f->w_in % 2) == 0 && (long)f->h_out * f->w_out >= 1024;
}

// packed operator of that 3x3 convolution in the tiled kernel's layout [ctile 0][ks = tap * nq + q][lane][8]: row (a, b, c),
// K = the forward conv's output channels; w = the forward OIHW weight [c_out][c_in][4][4] (/ sigma)
template <typename T>
__global__ void pack_dgrad_subpixel_kernel(const float* __restrict__ w, const float* __restrict__ sigma,
                                           uint16_t* __restrict__ packed, int c_out, int c_in, int cin_p) {
  const int nq = cin_p / 32, total = 9 * nq * 64;
  const float sig = sigma ? sigma[0] : 1.f;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int lane = idx & 63, ks = idx >> 6;
    const int tap = ks / nq, q = ks - tap * nq;
    const int ty = tap / 3, tx = tap - ty * 3;
    const int row = lane & 15, a = row >> 3, b = (row >> 2) & 1, c = row & 3;
    const int ky = a + 3 - 2 * ty, kx = b + 3 - 2 * tx;
    const int k0 = q * 32 + (lane >> 4) * 8;
    uint16_t o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int co = k0 + e;
      float v = 0.f;
      if (co < c_out && c < c_in && ky >= 0 && ky < 4 && kx >= 0 && kx < 4)
        v = __fdiv_rn(w[(((size_t)co * c_in + c) * 4 + ky) * 4 + kx], sig);
      o[e] = bits_of<T>(v);
    }
    u32x4 pk;
    pk[0] = o[0] | ((uint32_t)o[1] << 16);
    pk[1] = o[2] | ((uint32_t)o[3] << 16);
    pk[2] = o[4] | ((uint32_t)o[5] << 16);
    pk[3] = o[6] | ((uint32_t)o[7] << 16);
    reinterpret_cast<u32x4*>(packed)[idx] = pk;
  }
}

static size_t dgrad_packed_fragments(const ConvParams& p) {
  if (!p.cls_s) return (size_t)p.ctiles * p.ksteps * 64;
  size_t ks = 0;
  for (int c = 0; c < p.cls_s * p.cls_s; ++c) ks += cls_ksteps(p.cls_s, p.kh, p.kw, p.cls_pad, p.cg, c);
  return (size_t)p.ctiles * ks * 64;
}

extern "C" size_t cgan_conv2d_dgrad_packed_weight_bytes(const CganConvDesc* fwd) {
  ConvParams p;
  CganConvDesc t;
  if (dgrad_params(p, fwd, &t) != CGAN_OK) return 0;
  if (dgrad_subpixel(fwd)) return (size_t)9 * (((cgan_cs(fwd->c_out) + 31) & ~31) / 32) * 64 * 16;
  const size_t fr = dgrad_packed_fragments(p);
  return (fr ? fr : 64) * 16;
}

extern "C" int cgan_conv2d_pack_weight_dgrad(const float* w_oihw, const float* sigma, void* packed,
                                             const CganConvDesc* fwd, void* stream) {
  ConvParams p;
  CganConvDesc t;
  int rc = dgrad_params(p, fwd, &t);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(w_oihw && packed, "conv2d_pack_weight_dgrad: null pointer");
  if (dgrad_subpixel(fwd)) {
    const int cin_p = (cgan_cs(fwd->c_out) + 31) & ~31;
    const int tot = 9 * (cin_p / 32) * 64;
    hipStream_t st = (hipStream_t)stream;
    if (fwd->dtype == CGAN_F16)
      hipLaunchKernelGGL(pack_dgrad_subpixel_kernel<F16>, dim3(ceil_div(tot, 256)), dim3(256), 0, st, w_oihw, sigma,
                         (uint16_t*)packed, fwd->c_out, fwd->c_in, cin_p);
    else
      hipLaunchKernelGGL(pack_dgrad_subpixel_kernel<BF16>, dim3(ceil_div(tot, 256)), dim3(256), 0, st, w_oihw, sigma,
                         (uint16_t*)packed, fwd->c_out, fwd->c_in, cin_p);
    CGAN_CHECK_LAUNCH("conv2d_pack_weight_dgrad(sub-pixel)");
    return CGAN_OK;
  }
  if (p.cls_s) {
    int max_ks = 1;
    for (int c = 0; c < p.cls_s * p.cls_s; ++c) {
      const int k = cls_ksteps(p.cls_s, p.kh, p.kw, p.cls_pad, p.cg, c);
      max_ks = k > max_ks ? k : max_ks;
    }
    const int tot = p.ctiles * max_ks * 64;
    const dim3 grid(ceil_div(tot, 256) < 1024 ? ceil_div(tot, 256) : 1024, p.cls_s * p.cls_s);
    hipStream_t st = (hipStream_t)stream;
    if (fwd->dtype == CGAN_F16)
      hipLaunchKernelGGL(pack_dgrad_classes_kernel<F16>, grid, dim3(256), 0, st, w_oihw, sigma, (uint16_t*)packed, t.c_out,
                         t.c_in, p.cin_p, t.kh, t.kw, p.ctiles, p.cls_s, p.cls_pad);
    else
      hipLaunchKernelGGL(pack_dgrad_classes_kernel<BF16>, grid, dim3(256), 0, st, w_oihw, sigma, (uint16_t*)packed, t.c_out,
                         t.c_in, p.cin_p, t.kh, t.kw, p.ctiles, p.cls_s, p.cls_pad);
    CGAN_CHECK_LAUNCH("conv2d_pack_weight_dgrad(classes)");
    return CGAN_OK;
  }
  const int total = p.ctiles * p.ksteps * 64;
  const int blocks = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
  hipStream_t s = (hipStream_t)stream;
  // rows = forward c_in, K channels = forward c_out
  if (fwd->dtype == CGAN_F16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<F16>, dim3(blocks), dim3(256), 0, s, w_oihw, (const float*)nullptr, sigma,
                       (uint16_t*)packed, (float*)nullptr, t.c_out, t.c_in, p.cin_p, t.kh, t.kw, p.ctiles, p.ksteps, 1);
  else
    hipLaunchKernelGGL(pack_conv_weight_kernel<BF16>, dim3(blocks), dim3(256), 0, s, w_oihw, (const float*)nullptr, sigma,
                       (uint16_t*)packed, (float*)nullptr, t.c_out, t.c_in, p.cin_p, t.kh, t.kw, p.ctiles, p.ksteps, 1);
  CGAN_CHECK_LAUNCH("conv2d_pack_weight_dgrad");
  return CGAN_OK;
}

// parity-class data gradient on the LDS-tiled GEMM (conv_gemm_ext.hip): the class table, or false = stays on the general kernel
static bool cls_on_gemm(const ConvParams& p, ConvGemmCls (&cls)[4]) {
  if (!p.cls_s || p.cls_s > 2 || g_conv_force != 0 || p.cin_p != p.cin_s || p.pair) return false;
  if (!conv_gemm_ext_shape_ok(gemm_args(p)) || p.cout_s < 64) return false;
  int koff = 0;
  for (int c = 0; c < p.cls_s * p.cls_s; ++c) {
    const int ca = c / p.cls_s, cb = c % p.cls_s;
    int ry, rx, nty, ntx;
    cls_axis(p.cls_s, p.kh, p.cls_pad, ca, ry, nty);
    cls_axis(p.cls_s, p.kw, p.cls_pad, cb, rx, ntx);
    // (a - pad' + r) is a multiple of s by the choice of r: output row i s + a first reads dy row i + (a - pad' + r) / s
    cls[c] = ConvGemmCls{koff, nty, ntx, (ca - p.cls_pad + ry) / p.cls_s, (cb - p.cls_pad + rx) / p.cls_s};
    if (nty * ntx == 0) cls[c].kh = cls[c].kw = 0;
    koff += cls_ksteps(p.cls_s, p.kh, p.kw, p.cls_pad, p.cg, c);
  }
  return true;
}

// ws_bytes: the split-K workspace the launch in question would find (a split-K launch needs its stream's binding)
static int kernel_kind_impl(const CganConvDesc* d, int32_t bwd_data, size_t ws_bytes) {
  ConvParams p;
  if (!bwd_data) {
    int rc = fill_params(p, d);
    if (rc != CGAN_OK) return rc;
    const int kind = select_conv_kernel(p, d);
    return splitk_candidate(kind, p) && splitk_for(p, ws_bytes) > 1 ? CGAN_CONV_KERNEL_GEMM : kind;
  }
  CganConvDesc t;
  int rc = dgrad_params(p, d, &t);
  if (rc != CGAN_OK) return rc;
  const bool plain = d->stride == 1 && p.pad >= 0 && t.h_in + 2 * p.pad - t.dilation * (t.kh - 1) == t.h_out &&
                     t.w_in + 2 * p.pad - t.dilation * (t.kw - 1) == t.w_out;
  if (!plain) {
    if (dgrad_subpixel(d)) return CGAN_CONV_KERNEL_LDS3X3;
    ConvGemmCls cls[4];
    return cls_on_gemm(p, cls) ? CGAN_CONV_KERNEL_GEMM : CGAN_CONV_KERNEL_GENERAL;
  }
  t.pad = p.pad;
  const int kind = select_conv_kernel(p, &t);
  return splitk_candidate(kind, p) && splitk_for(p, ws_bytes) > 1 ? CGAN_CONV_KERNEL_GEMM : kind;
}

// the kernel a launch of this descriptor on ``stream`` (of the current device) runs: that stream's workspace decides split-K
extern "C" int cgan_conv2d_kernel_kind_on(const CganConvDesc* d, int32_t bwd_data, void* stream) {
  return kernel_kind_impl(d, bwd_data, ws_lookup(stream).bytes);
}
// (kept for callers without a stream at hand: answers for the largest workspace bound on the current device)
extern "C" int cgan_conv2d_kernel_kind(const CganConvDesc* d, int32_t bwd_data) {
  return kernel_kind_impl(d, bwd_data, ws_max_bytes());
}

static int bwd_data_impl(const void* dy, const void* packed_w_dgrad, const void* dx_add, void* dx, const CganConvDesc* fwd,
                         void* stream, int res_mode = 1, const void* relu_out2 = nullptr) {
  ConvParams p;
  CganConvDesc t;
  int rc = dgrad_params(p, fwd, &t);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(dy && packed_w_dgrad && dx, "conv2d_nhwc_bwd_data: null pointer");
  p.x = (const uint16_t*)dy; p.w = (const u32x4*)packed_w_dgrad; p.bias = nullptr; p.res = nullptr; p.y = (uint16_t*)dx;
  hipStream_t s = (hipStream_t)stream;
  const bool plain = fwd->stride == 1 && p.pad >= 0 && t.h_in + 2 * p.pad - t.dilation * (t.kh - 1) == t.h_out &&
                     t.w_in + 2 * p.pad - t.dilation * (t.kw - 1) == t.w_out;
  CGAN_REQUIRE(dx_add == nullptr || plain, "conv2d_nhwc_bwd_data_add: only for stride-1 'same' convolutions");
  if (dx_add) {             // the other gradient contribution of the same tensor rides in the epilogue's residual slot
    p.res = (const uint16_t*)dx_add;
    p.has_res = res_mode;         // 1: add the other contribution; 2: the ReLU derivative from the activation's output;
    p.res2 = (const uint16_t*)relu_out2;   // 3: both (the caller checked conv_gemm_res2_ok)
    p.res_ups = 0;
    t.has_residual = 1;
  }
  if (plain) {
    t.pad = p.pad;
    return dispatch_conv(p, &t, s, "conv2d_nhwc_bwd_data");
  }
  if (dgrad_subpixel(fwd)) {
    Conv3x3LdsArgs a;
    a.x = p.x; a.w = p.w; a.bias = nullptr; a.res = nullptr; a.y = p.y;
    a.n = fwd->n; a.h = fwd->h_out; a.w_ = fwd->w_out; a.hi = fwd->h_out; a.wi = fwd->w_out; a.pad = 1; a.reflect = 0;
    a.hx = fwd->h_out; a.wx = fwd->w_out;
    a.cin = fwd->c_out; a.cin_s = cgan_cs(fwd->c_out); a.cin_p = (a.cin_s + 31) & ~31;
    a.cout = 16; a.cout_s = 16; a.ctiles = 1; a.ksteps = 9 * (a.cin_p / 32);
    a.in_ups = 0; a.act = CGAN_ACT_NONE; a.has_res = 0; a.res_ups = 0; a.slope = 0.f;
    a.shuffle = fwd->h_in; a.shuffle_w = fwd->w_in; a.k = 0; a.stride = 0;
    int rc2 = conv3x3_lds_launch(a, fwd->dtype, s);
    if (rc2 != CGAN_OK) return rc2;
    CGAN_CHECK_LAUNCH("conv2d_nhwc_bwd_data(sub-pixel)");
    return CGAN_OK;
  }
  ConvGemmCls cls[4];
  if (cls_on_gemm(p, cls)) {
    int rc2 = conv_gemm_cls_launch(gemm_args(p), p.cls_s, cls, fwd->dtype, s);
    if (rc2 != CGAN_OK) return rc2;
    CGAN_CHECK_LAUNCH("conv2d_nhwc_bwd_data");
    return CGAN_OK;
  }
  if (fwd->dtype == CGAN_F16) launch<F16>(p, s);
  else launch<BF16>(p, s);
  CGAN_CHECK_LAUNCH("conv2d_nhwc_bwd_data");
  return CGAN_OK;
}

extern "C" int cgan_conv2d_nhwc_bwd_data(const void* dy, const void* packed_w_dgrad, void* dx, const CganConvDesc* fwd,
                                         void* stream) {
  return bwd_data_impl(dy, packed_w_dgrad, nullptr, dx, fwd, stream);
}

extern "C" int cgan_conv2d_nhwc_bwd_data_relu(const void* dy, const void* packed_w_dgrad, const void* relu_out, void* dx,
                                              const CganConvDesc* fwd, void* stream) {
  CGAN_REQUIRE(relu_out != nullptr, "conv2d_nhwc_bwd_data_relu: null pointer");
  return bwd_data_impl(dy, packed_w_dgrad, relu_out, dx, fwd, stream, 2);
}

// dx = [relu_out > 0] * (data gradient + dx_add).  Fused (one launch, has_res = 3) where the dispatcher's kernel for this
// descriptor has the shared store path; elsewhere the add rides in the epilogue and the ReLU derivative is a pass of its own
// over dx (cgan_act_bwd, in place: element i is read and written by the same thread) -- same values either way.
extern "C" int cgan_conv2d_nhwc_bwd_data_add_relu(const void* dy, const void* packed_w_dgrad, const void* dx_add,
                                                  const void* relu_out, void* dx, const CganConvDesc* fwd, void* stream) {
  CGAN_REQUIRE(dx_add != nullptr && relu_out != nullptr, "conv2d_nhwc_bwd_data_add_relu: null pointer");
  ConvParams p;
  CganConvDesc t;
  int rc = dgrad_params(p, fwd, &t);
  if (rc != CGAN_OK) return rc;
  const bool plain = fwd->stride == 1 && p.pad >= 0 && t.h_in + 2 * p.pad - t.dilation * (t.kh - 1) == t.h_out &&
                     t.w_in + 2 * p.pad - t.dilation * (t.kw - 1) == t.w_out;
  CGAN_REQUIRE(plain, "conv2d_nhwc_bwd_data_add_relu: only for stride-1 'same' convolutions");
  p.res = (const uint16_t*)dx_add; p.has_res = 1; p.res_ups = 0;
  t.has_residual = 1; t.pad = p.pad;
  if (select_conv_kernel(p, &t) == CGAN_CONV_KERNEL_GEMM && conv_gemm_res2_ok(gemm_args(p), fwd->dtype))
    return bwd_data_impl(dy, packed_w_dgrad, dx_add, dx, fwd, stream, 3, relu_out);
  rc = bwd_data_impl(dy, packed_w_dgrad, dx_add, dx, fwd, stream);
  if (rc != CGAN_OK) return rc;
  return cgan_act_bwd(relu_out, dx, dx, fwd->dtype, CGAN_ACT_RELU, 0.f,
                      (int64_t)fwd->n * fwd->h_in * fwd->w_in * cgan_cs(fwd->c_in), stream);
}

extern "C" int cgan_conv2d_nhwc_bwd_data_add(const void* dy, const void* packed_w_dgrad, const void* dx_add, void* dx,
                                             const CganConvDesc* fwd, void* stream) {
  CGAN_REQUIRE(dx_add != nullptr, "conv2d_nhwc_bwd_data_add: null pointer");
  return bwd_data_impl(dy, packed_w_dgrad, dx_add, dx, fwd, stream);
}
