// The LDS-tiled implicit GEMM of conv_gemm.hip (same operand layout, same 3-stage LDS-DMA ring, same fragment reads) for the
// two kinds of launch the plain kernel does not take (round 5; until then both ran on the general gather kernel of
// conv_mfma.hip at 0.02-0.11 of the MFMA peak):
//
//  (1) PARITY CLASSES: the data gradient of a stride-s convolution (PatchGAN / ADVENT 4x4 s2, ResNet 3x3 s2, 1x1 s2
//      shortcuts; reference discriminator.py:100-163, 327-349, resnet101_v3.py:30-50).  All outputs (y, x) = (a, b) mod s use
//      the same ceil(k/s)^2 flipped taps and read dy at consecutive positions: per class a dense stride-1 convolution with
//      its own tap count, its own offset (the class's "padding") and its own block of the packed operator
//      (pack_dgrad_classes_kernel, conv_mfma.hip).  blockIdx.y = class; the epilogue scatters the class's pixels to
//      (i s + a, j s + b) of dx.  A class without taps (1x1 s2: three of four) writes zeros.
//  (2) SPLIT-K for grids that cannot fill the chip: few output pixels, long K -- the Painter's 640-channel 3x3 layers at
//      5^2 .. 20^2 (painter.py:149-160), the discriminators' 512 -> 512 4x4 layers at 20^2 / 10^2, the strided PatchGAN convs
//      at 40^2 / 20^2, the SPADE gamma||beta data gradients at 5^2 .. 20^2.  blockIdx.y = K slice; each slice leaves its
//      fp32 partial tile in a workspace [slice][pixel][cout_s] (coalesced through the same LDS staging as a normal store) and
//      conv_splitk_reduce_kernel sums the slices IN ORDER (deterministic) and applies bias / residual / activation / the
//      pad-channel zeroing in the plain kernel's operation order.
//
// The plain kernel stays untouched: its hot loop has no class / slice state.
#include "conv_gemm.h"
#include <type_traits>

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_ext_zeros[4];

template <int I, int N, typename F>
__device__ __forceinline__ void ext_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ext_static_for<I + 1, N>(f);
  }
}

template <typename T, int WAVES_C, int WC, int WP>
__global__ __launch_bounds__(256, 2) void conv_gemm_ext_kernel(ConvGemmExtArgs q, int npb, int ncb) {
  constexpr int NS = 3;
  constexpr int WAVES_P = 4 / WAVES_C;
  constexpr int CT_BLK = WAVES_C * WC;
  constexpr int PT_BLK = WAVES_P * WP;
  constexpr int STAGE_BYTES = (CT_BLK + PT_BLK) * 1024;
  constexpr int W_PER_WAVE = CT_BLK / 4;
  constexpr int P_PER_WAVE = PT_BLK / 4;
  constexpr int DMA_PER_WAVE = W_PER_WAVE + P_PER_WAVE;
  static_assert(CT_BLK % 4 == 0 && PT_BLK % 4 == 0, "tiles must split evenly over the 4 waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const ConvGemmArgs& p = q.a;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int g = lane >> 4;
  const int wc = wave % WAVES_C, wp = wave / WAVES_C;

  // ---- this workgroup's slice of the problem (wave-uniform): a parity class or a K range
  const int yi = blockIdx.y;
  const int ccn = p.cin_s >> 5;   // k-steps per tap
  int kh = p.kh, kw = p.kw, ksteps = p.ksteps, kpitch = p.ksteps, ks0 = 0;
  int hc = p.h_out, wcls = p.w_out, npix = p.npix, offy = -p.pad, offx = -p.pad, stride = p.stride;
  int ca = 0, cb = 0;
  const u32x4* wbase = p.w;
  if (q.cls_s) {
    const ConvGemmCls c = q.cls[yi];
    kh = c.kh; kw = c.kw;
    ksteps = kpitch = c.kh * c.kw * ccn;
    wbase = p.w + (size_t)p.ctiles * c.koff * 64;
    ca = yi / q.cls_s;
    cb = yi - ca * q.cls_s;
    hc = p.h_out > ca ? (p.h_out - ca + q.cls_s - 1) / q.cls_s : 0;
    wcls = p.w_out > cb ? (p.w_out - cb + q.cls_s - 1) / q.cls_s : 0;
    npix = p.n * hc * wcls;
    offy = c.offy; offx = c.offx;
    stride = 1;
  } else if (q.ksplit > 1) {
    ks0 = yi * q.ks_per;
    ksteps = min(q.ks_per, p.ksteps - ks0);
  }

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cblk = slot % ncb;
  const int pblk = xcd * ((npb + 7) >> 3) + slot / ncb;
  if (pblk >= npb || pblk * PT_BLK * 16 >= npix) return;

  // split maps (q.pair): the map stores each component once (Split<T>::NS blocks of csb channels per pixel); K-block b of the
  // conv's NB * csb input channels is storage block xcomp(b): the pixel stride is the stored one and a lane's 8-channel group
  // is looked up per stage (pair_coff)
  const int csb = q.pair ? p.cin_s / Split<T>::NB : p.cin_s;
  const int xstride = q.pair ? Split<T>::NS * csb : p.cin_s;
  const float inv_csb = 1.0f / (float)csb;
  auto pair_coff = [&](int c) {       // channel c of the K extent -> channel of the stored pixel (c, csb multiples of 8)
    const int b = (int)(((float)c + 0.5f) * inv_csb);
    return Split<T>::xcomp(b) * csb + (c - b * csb);
  };
  int pbase[P_PER_WAVE], py0[P_PER_WAVE], px0[P_PER_WAVE];
#pragma unroll
  for (int m = 0; m < P_PER_WAVE; ++m) {
    int pix = (pblk * PT_BLK + wave + 4 * m) * 16 + j;
    bool v = pix < npix;
    int pc = v ? pix : 0;
    int ox = pc % wcls;
    int r = pc / wcls;
    int oy = r % hc;
    int nn = r / hc;
    pbase[m] = nn * p.h_in * p.w_in * xstride + (q.pair ? 0 : g * 8);
    py0[m] = v ? oy * stride + offy : -(1 << 28);
    px0[m] = ox * stride + offx;
  }
  const long zero_off = reinterpret_cast<const unsigned char*>(g_ext_zeros) - reinterpret_cast<const unsigned char*>(p.x);

  // (tap, channel chunk, ring slot) of the NEXT stage to issue; K order = channel chunk outer, tap inner
  const int taps = kh * kw;
  int i_cc = ks0 / taps;
  int i_ky = (ks0 - i_cc * taps) / kw;
  int i_kx = ks0 - i_cc * taps - i_ky * kw;
  int i_buf = 0;
  auto issue_piece = [&](int piece) {
    unsigned char* buf = smem + i_buf * STAGE_BYTES;
    if (piece < W_PER_WAVE) {
      const int i = wave + 4 * piece;
      const int ct = min(cblk * CT_BLK + i, p.ctiles - 1);
      const u32x4* src = wbase + ((size_t)ct * kpitch + (i_ky * kw + i_kx) * ccn + i_cc) * 64 + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + i * 1024), 16, 0, 0);
    } else {
      const int m = piece - W_PER_WAVE;
      const int i = wave + 4 * m;
      const int iy = py0[m] + i_ky * p.dil, ix = px0[m] + i_kx * p.dil;
      const bool ok = (unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in;
      const int coff = q.pair ? pair_coff(i_cc * 32 + g * 8) : i_cc * 32;
      const long off = ok ? (long)(pbase[m] + (iy * p.w_in + ix) * xstride + coff) * 2 : zero_off;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + off;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + (CT_BLK + i) * 1024), 16, 0, 0);
    }
  };
  auto issue_advance = [&]() {
    const int kx1 = i_kx + 1;
    const bool wx = kx1 == kw;
    i_kx = wx ? 0 : kx1;
    const int ky1 = i_ky + (wx ? 1 : 0);
    const bool wy = ky1 == kh;
    i_ky = wy ? 0 : ky1;
    const int cc1 = i_cc + (wy ? 1 : 0);
    i_cc = cc1 == ccn ? 0 : cc1;          // stages past the end wrap to valid addresses and are never read
    i_buf = (i_buf + 1 == NS) ? 0 : i_buf + 1;
  };
  auto issue = [&]() {
#pragma unroll
    for (int qq = 0; qq < DMA_PER_WAVE; ++qq) issue_piece(qq);
    issue_advance();
  };

  f32x4 acc[WC][WP];
#pragma unroll
  for (int c = 0; c < WC; ++c)
#pragma unroll
    for (int t = 0; t < WP; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (ksteps > 0) {       // (a class without taps has no operator block to read from)
    issue();
    issue();
  }
  int r_buf = 0;
  for (int ks = 0; ks < ksteps; ++ks) {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * DMA_PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* buf = smem + r_buf * STAGE_BYTES;
    u32x4 a[WC], b[WP];
#pragma unroll
    for (int t = 0; t < WP; ++t)
      b[t] = *reinterpret_cast<const u32x4*>(buf + (CT_BLK + wp * WP + t) * 1024 + lane * 16);
#pragma unroll
    for (int c = 0; c < WC; ++c) a[c] = *reinterpret_cast<const u32x4*>(buf + (wc * WC + c) * 1024 + lane * 16);
    constexpr int NM = WC * WP;
    constexpr int GAP = NM / (DMA_PER_WAVE + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int c = i / WP, t = i % WP;
      acc[c][t] = mfma16(as_vec8<T>(a[c]), as_vec8<T>(b[t]), acc[c][t]);
      if ((i + 1) % GAP == 0 && (i + 1) / GAP <= DMA_PER_WAVE) {
        issue_piece((i + 1) / GAP - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    issue_advance();
    r_buf = (r_buf + 1 == NS) ? 0 : r_buf + 1;
  }

  // ---- epilogue: staged through LDS (fp32 rows of one pixel x the wave's WC*16 couts) and written as whole 8-channel chunks:
  // a K slice leaves its raw fp32 partials in the workspace, a parity class (or a plain launch) the finished 16-bit values
  constexpr int ROWB = WC * 64 + 16;
  constexpr int PP = (4 * 4 * 16 * ROWB <= NS * STAGE_BYTES && WP % 4 == 0) ? 4
                     : (4 * 2 * 16 * ROWB <= NS * STAGE_BYTES ? 2 : 1);
  constexpr int CH = WC * 2;
  static_assert(4 * PP * 16 * ROWB <= NS * STAGE_BYTES, "epilogue staging does not fit");
  static_assert(WP % PP == 0, "WP must be a multiple of PP");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  unsigned char* stg = smem + wave * (PP * 16 * ROWB);
  const int cout_base = (cblk * CT_BLK + wc * WC) * 16;
  const bool split = q.ksplit > 1;
  // (round 6, as conv_gemm_staged_store: a lane's 8-channel chunk is the same in every iteration -- the bias is loaded once;
  // a pass's residual components (split-precision launches) are requested together once the pass is staged; one wait per
  // pass instead of one in front of every store)
  constexpr int NIT = PP * 16 * CH / 64;
  constexpr int PSTEP = 64 / CH;
  constexpr int NSB = Split<T>::NS, NCC = Split<T>::NC;
  static_assert(64 % CH == 0, "a lane keeps its channel chunk over the iterations");
  const int qc = lane % CH, pl0 = lane / CH;
  const int ch = cout_base + qc * 8;
  const bool ch_ok = ch < p.cout_s;
  f32x4 bb0 = (f32x4){0.f, 0.f, 0.f, 0.f}, bb1 = bb0;
  if (p.bias && !split && ch_ok) {
    bb0 = *reinterpret_cast<const f32x4*>(p.bias + ch);
    bb1 = *reinterpret_cast<const f32x4*>(p.bias + ch + 4);
  }
  // (a lambda per pass with a compile-time pass index: written as a loop with ``continue`` the compiler did not unroll it for
  // the 128 x 256 tile and kept the accumulators -- indexed by the pass -- in scratch: 528 bytes per lane)
  auto do_pass = [&](auto pass_tag) {
    constexpr int pass = decltype(pass_tag)::value;
#pragma unroll
    for (int tt = 0; tt < PP; ++tt)
#pragma unroll
      for (int c = 0; c < WC; ++c)
        *reinterpret_cast<f32x4*>(stg + (tt * 16 + j) * ROWB + c * 64 + g * 16) = acc[c][pass * PP + tt];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int pix_base = (pblk * PT_BLK + wp * WP + pass * PP) * 16;
    if (split) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pl = it * PSTEP + pl0;
        const int pix = pix_base + pl;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32 + 16);
        if (pix >= npix || !ch_ok) continue;
        float* dst = q.ws + ((size_t)yi * p.npix + pix) * p.cout_s + ch;
        *reinterpret_cast<f32x4*>(dst) = v0;
        *reinterpret_cast<f32x4*>(dst + 4) = v1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      return;
    }
    if (q.pair) {
      // split-precision epilogue (as conv_mfma_kernel's): residual = the sum of its components, activation in fp32, then
      // v -> c0 = round16(v), c1 = round16(v - c0), ... stored as the channel blocks the next conv multiplies (Split<T>)
      u32x4 rs[NIT][NCC];
      if (p.has_res) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int pix = pix_base + it * PSTEP + pl0;
          const bool ok = pix < npix && ch_ok;
          size_t rbase = (size_t)(ok ? pix : 0);
          if (p.res_ups && ok) {
            const int ox = pix % p.w_out;
            const int r = pix / p.w_out;
            const int oy = r % p.h_out;
            const int nn = r / p.h_out;
            rbase = ((size_t)nn * (p.h_out >> 1) + (oy >> 1)) * (p.w_out >> 1) + (ox >> 1);
          }
          const uint16_t* rp = p.res + rbase * p.cout_s * NSB + ch;
#pragma unroll
          for (int k = 0; k < NCC; ++k) {
            rs[it][k] = (u32x4){0u, 0u, 0u, 0u};
            if (ok) rs[it][k] = *reinterpret_cast<const u32x4*>(rp + k * p.cout_s);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0): the bias and this pass's residual, once (conv_gemm.h)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pl = it * PSTEP + pl0;
        const int pix = pix_base + pl;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32 + 16);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] += bb0[r];
            v[4 + r] += bb1[r];
          }
        }
        if (p.has_res) {
          float rsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = NCC - 1; k >= 0; --k) {         // smallest component first
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float r0, r1;
              unpack2<T>(rs[it][k][e], r0, r1);
              rsum[2 * e] += r0;
              rsum[2 * e + 1] += r1;
            }
          }
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += rsum[r];
        }
        act_apply_n(v, p.act, p.slope);
        if (p.cout < p.cout_s) {
#pragma unroll
          for (int r = 0; r < 8; ++r)
            if (ch + r >= p.cout) v[r] = 0.f;
        }
        uint16_t* yp = p.y + (size_t)pix * p.cout_s * NSB + ch;
#pragma unroll
        for (int k = 0; k < NCC; ++k) {
          u32x4 comp;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            comp[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
            float q0, q1;
            unpack2<T>(comp[e], q0, q1);
            v[2 * e] -= q0;
            v[2 * e + 1] -= q1;
          }
          if (pix < npix && ch_ok) *reinterpret_cast<u32x4*>(yp + k * p.cout_s) = comp;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      return;
    }
    if (pass == 0) __builtin_amdgcn_s_waitcnt(0x0F70);        // the bias
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pl = it * PSTEP + pl0;
      const int pix = pix_base + pl;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32 + 16);
      if (pix >= npix || !ch_ok) continue;
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] += bb0[r];
          v[4 + r] += bb1[r];
        }
      }
      act_apply_n(v, p.act, p.slope);
      if (p.cout < p.cout_s) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
          if (ch + r >= p.cout) v[r] = 0.f;
      }
      size_t opix = (size_t)pix;
      if (q.cls_s) {
        const int jx = pix % wcls;
        const int r = pix / wcls;
        const int iy = r % hc;
        const int nn = r / hc;
        opix = ((size_t)nn * p.h_out + iy * q.cls_s + ca) * p.w_out + jx * q.cls_s + cb;
      }
      u32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
      CGAN_ST_STREAM(o, reinterpret_cast<u32x4*>(p.y + opix * p.cout_s + ch));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  ext_static_for<0, WP / PP>(do_pass);
}

// sum of the K slices (in slice order: deterministic) + the plain kernel's epilogue, one thread per (pixel, 8-channel chunk)
template <typename T>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvGemmArgs p, const float* __restrict__ ws, int ksplit) {
  const int chunks = p.cout_s >> 3;
  const int total = p.npix * chunks;
  const size_t slice = (size_t)p.npix * p.cout_s;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
    const int pix = idx / chunks, ch = (idx - pix * chunks) * 8;
    const float* src = ws + (size_t)pix * p.cout_s + ch;
    f32x4 s0 = *reinterpret_cast<const f32x4*>(src), s1 = *reinterpret_cast<const f32x4*>(src + 4);
    for (int k = 1; k < ksplit; ++k) {
      s0 += *reinterpret_cast<const f32x4*>(src + k * slice);
      s1 += *reinterpret_cast<const f32x4*>(src + k * slice + 4);
    }
    float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
    if (p.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + ch), b1 = *reinterpret_cast<const f32x4*>(p.bias + ch + 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] += b0[r];
        v[4 + r] += b1[r];
      }
    }
    if (p.has_res) {
      size_t rbase;
      if (p.res_ups) {
        int ox = pix % p.w_out;
        int r = pix / p.w_out;
        int oy = r % p.h_out;
        int nn = r / p.h_out;
        rbase = (((size_t)nn * (p.h_out >> 1) + (oy >> 1)) * (p.w_out >> 1) + (ox >> 1)) * p.cout_s;
      } else {
        rbase = (size_t)pix * p.cout_s;
      }
      const u32x4 rv = *reinterpret_cast<const u32x4*>(p.res + rbase + ch);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float r0, r1;
        unpack2<T>(rv[e], r0, r1);
        v[2 * e] = cgan_res_apply(v[2 * e], r0, p.has_res);
        v[2 * e + 1] = cgan_res_apply(v[2 * e + 1], r1, p.has_res);
      }
    }
    act_apply_n(v, p.act, p.slope);
    if (p.cout < p.cout_s) {
#pragma unroll
      for (int r = 0; r < 8; ++r)
        if (ch + r >= p.cout) v[r] = 0.f;
    }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
    *reinterpret_cast<u32x4*>(p.y + (size_t)pix * p.cout_s + ch) = o;
  }
}

template <typename T, int WAVES_C, int WC, int WP>
int launch_tile(const ConvGemmExtArgs& q, int grid_pixels, int ny, hipStream_t s) {
  constexpr int WAVES_P = 4 / WAVES_C;
  constexpr int CT_BLK = WAVES_C * WC, PT_BLK = WAVES_P * WP;
  constexpr size_t smem = (size_t)3 * (CT_BLK + PT_BLK) * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_ext_kernel<T, WAVES_C, WC, WP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("conv_gemm_ext: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  const int npb = ceil_div(ceil_div(grid_pixels, 16), PT_BLK);
  const int ncb = ceil_div(q.a.ctiles, CT_BLK);
  const int grid = ceil_div(npb, 8) * 8 * ncb;
  hipLaunchKernelGGL((conv_gemm_ext_kernel<T, WAVES_C, WC, WP>), dim3(grid, ny), dim3(256), smem, s, q, npb, ncb);
  return CGAN_OK;
}

// block tile by shape: 64 couts x 256 pixels for <= 64 output channels, 128 x 256 while that still gives every CU a couple of
// workgroups (all y slices counted), else 128 x 128
int tile_cfg(int ctiles, int grid_pixels, int ny) {
  if (ctiles <= 4) return 1;
  const int ptiles = ceil_div(grid_pixels, 16);
  if ((long)ceil_div(ptiles, 16) * ceil_div(ctiles, 8) * ny >= 512) return 3;
  return 4;
}

template <typename T>
int launch_any(const ConvGemmExtArgs& q, int grid_pixels, int ny, hipStream_t s) {
  switch (tile_cfg(q.a.ctiles, grid_pixels, ny)) {
    case 1: return launch_tile<T, 1, 4, 4>(q, grid_pixels, ny, s);
    case 3: return launch_tile<T, 2, 4, 8>(q, grid_pixels, ny, s);
    default: return launch_tile<T, 2, 4, 4>(q, grid_pixels, ny, s);
  }
}

}  // namespace

// channel / addressing preconditions shared by both launch kinds (the plain kernel's, without its size thresholds and
// without its 64-cout floor: a split-K launch of a c_out = 1 PatchGAN head, discriminator.py:163, multiplies 63 dead rows
// and still takes a fifth of the gather kernel's time -- its K = 8192 is what needs the parallelism)
bool conv_gemm_ext_shape_ok(const ConvGemmArgs& a) {
  return (a.cin_s % 32) == 0 && a.pad_mode != CGAN_PAD_REFLECT &&
         (long)a.n * a.h_in * a.w_in * a.cin_s < (1L << 31) - (1L << 20) && (long)a.npix * a.cout_s < (1L << 31);
}

// K slices for a launch whose (cout block x pixel block) grid cannot fill the chip: enough slices for ~2 workgroups per CU,
// each at least 8 k-steps long; 1 = do not split
int conv_gemm_splitk_plan(const ConvGemmArgs& a) {
  if (!conv_gemm_ext_shape_ok(a) || a.ksteps < 16) return 1;
  const int ptiles = ceil_div(a.npix, 16);
  const int blocks = a.ctiles <= 4 ? ceil_div(ptiles, 16) : ceil_div(ptiles, 8) * ceil_div(a.ctiles, 8);
  if (blocks >= 256) return 1;
  int ksplit = ceil_div(512, blocks);
  if (ksplit > a.ksteps / 8) ksplit = a.ksteps / 8;
  if (ksplit > 64) ksplit = 64;
  if (ksplit < 2) return 1;
  const int per = ceil_div(a.ksteps, ksplit);
  return ceil_div(a.ksteps, per);
}

size_t conv_gemm_splitk_workspace_bytes(const ConvGemmArgs& a, int ksplit) {
  return (size_t)ksplit * a.npix * a.cout_s * sizeof(float);
}

int conv_gemm_splitk_launch(const ConvGemmArgs& a, int ksplit, float* ws, int dtype, hipStream_t s) {
  ConvGemmExtArgs q;
  q.a = a;
  q.cls_s = 0;
  q.ksplit = ksplit;
  q.ks_per = ceil_div(a.ksteps, ksplit);
  q.ws = ws;
  q.pair = 0;
  for (int c = 0; c < 4; ++c) q.cls[c] = ConvGemmCls{0, 0, 0, 0, 0};
  int rc = dtype == CGAN_F16 ? launch_any<F16>(q, a.npix, ksplit, s) : launch_any<BF16>(q, a.npix, ksplit, s);
  if (rc != CGAN_OK) return rc;
  const int total = a.npix * (a.cout_s >> 3);
  const int blocks = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
  if (dtype == CGAN_F16)
    hipLaunchKernelGGL(conv_splitk_reduce_kernel<F16>, dim3(blocks), dim3(256), 0, s, a, ws, ksplit);
  else
    hipLaunchKernelGGL(conv_splitk_reduce_kernel<BF16>, dim3(blocks), dim3(256), 0, s, a, ws, ksplit);
  return CGAN_OK;
}

// ``a`` describes the data gradient as conv_mfma.hip's dgrad_params leaves it: x = dy (h_in x w_in = its stored extent),
// h_out x w_out = the forward input's extent, kh / kw = the forward kernel; cls[] filled by the caller per class
int conv_gemm_cls_launch(const ConvGemmArgs& a, int cls_s, const ConvGemmCls* cls, int dtype, hipStream_t s) {
  ConvGemmExtArgs q;
  q.a = a;
  q.cls_s = cls_s;
  q.ksplit = 1;
  q.ks_per = 0;
  q.ws = nullptr;
  q.pair = 0;
  for (int c = 0; c < 4; ++c) q.cls[c] = c < cls_s * cls_s ? cls[c] : ConvGemmCls{0, 0, 0, 0, 0};
  const int grid_pixels = a.n * ceil_div(a.h_out, cls_s) * ceil_div(a.w_out, cls_s);     // the largest class
  return dtype == CGAN_F16 ? launch_any<F16>(q, grid_pixels, cls_s * cls_s, s) : launch_any<BF16>(q, grid_pixels, cls_s * cls_s, s);
}

// Split-precision forward (cgan_conv2d_nhwc_fwd_pair) on this tiling: ``a`` describes the conv over the NB * round_up(C, 8)
// storage channels of the split input map; the epilogue takes bias / split residual / activation in fp32 and stores the
// components.  (Round 4 ran every split conv on the gather kernel: 20 images/s of apply_events in the fp32-grade mode.)
int conv_gemm_pair_launch(const ConvGemmArgs& a, int dtype, hipStream_t s) {
  ConvGemmExtArgs q;
  q.a = a;
  q.cls_s = 0;
  q.ksplit = 1;
  q.ks_per = 0;
  q.ws = nullptr;
  q.pair = 1;
  for (int c = 0; c < 4; ++c) q.cls[c] = ConvGemmCls{0, 0, 0, 0, 0};
  return dtype == CGAN_F16 ? launch_any<F16>(q, a.npix, 1, s) : launch_any<BF16>(q, a.npix, 1, s);
}
