// conv2d NHWC forward / data gradient for the WIDE layers with >= 256 output channels: 256 couts x 256 pixels per
// workgroup, K = 64 per pipeline stage, whole cache lines per pixel, one workgroup per CU (round 3).
//
// Why another tile.  Both operands of the implicit GEMM reach the MFMAs through LDS, filled by LDS-DMA from L2 / HBM.
// With a (BC couts x BP pixels) block tile the L2 -> LDS traffic of a layer is
//     npix * K * 2 B * (cout / BC)   +   cout * K * 2 B * (npix / BP)
// i.e. every operand is re-fetched once per block of the OTHER dimension.  For the ResNet bottleneck's 1x1 layers at
// 8 x 80 x 80 (256 -> 1024: 26 MB in, 105 MB out) the 128 x 128 tile of conv_gemm.hip moves 420 MB through the fill path
// for 131 MB of HBM traffic, the 256 x 128 tile 315 MB for the 1024 -> 256 layer: those layers ran at 2.0-2.4 TB/s of
// algorithmic traffic (25 % of the HBM roofline, round-2 bench) because the FILL, not HBM and not the MFMAs, was the
// limit.  256 x 256 halves the fill per FLOP (210 MB for both layers) -- the largest tile whose accumulators fit: 256 KiB
// of fp32, i.e. ALL 512 registers of four waves (one per SIMD: 8 x 8 accumulator tiles = 256 registers per lane, the rest
// for two fragment sets) or 128 per wave with eight waves.
//
// Pipeline (K = 64 channels of one tap per stage; K order = (64-channel chunk, tap, half), as conv_gemm_k64_kernel):
//   * LDS = 160 KiB exactly: a 2-deep ring of weight stages (256 couts x 64 k x 2 B = 32 KiB each, MFMA fragment order:
//     weights are pre-packed that way) + a 3-deep ring of pixel stages (256 pixels x 128 B = 32 KiB each).  Weights are
//     L2-resident (one stage of lead is enough), the activations of a 1x1 layer stream from HBM (two stages of lead);
//   * a pixel piece (one LDS-DMA wave instruction, 1 KiB) is 8 pixels x 128 B = 8 whole cache lines; lane l = 8 q + s fetches
//     16-byte chunk s ^ (4 h + ((q >> 1) & 3)) of pixel 8 h + q of its 16-pixel MFMA tile, so that the B-fragment
//     ds_read_b128 of either k-half is conflict-free (same image as conv_gemm_k64_kernel, checked exhaustively there);
//   * every wave is producer AND consumer: per k-half WC * WP MFMAs from one fragment set with, slotted between them, the
//     fragment reads of the next half (4-wave variant: second register set) and this wave's share of the LDS-DMA pieces
//     of stage s + 2 (pixel pieces in the first half, weight pieces in the second);
//   * ONE barrier per stage, in the middle of it: once a wave holds the fragments of both halves of stage s in registers
//     the stage's slots are free; it then waits for its own pieces of stage s + 1 (counted vmcnt: the pixel pieces of
//     stage s + 2 stay in flight), and the barrier publishes both facts.  LDS-DMA data is visible to another wave's
//     ds_read exactly under that sequence (issuing wave's vmcnt, then a barrier the reader has passed).
// Epilogue: the operand rings are dead by then; accumulators are staged through them in fp32 and leave as coalesced
// 16-byte stores with bias / residual / activation applied in fp32, the code path of conv_gemm_kernel.
#include "conv_gemm.h"
#include <type_traits>

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_big_zeros[4];   // what the lanes of a padded tap fetch

// PAIR (round 6): the split-precision forward (cgan_conv2d_nhwc_fwd_pair) on this tile.  x is a split map that stores every
// component once (Split<T>::NS blocks of csb channels per pixel); the K extent p.cin_s = NB * csb, K-block b = storage block
// xcomp(b) -- a lane's 8-channel group of a stage is looked up when the stage is issued; the epilogue takes bias / split
// residual / activation in fp32 and stores the components (as conv_gemm_ext_kernel's pair epilogue does).
template <typename T, int NW, bool PAIR = false, bool RES2 = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void conv_gemm_big_kernel(ConvGemmArgs p, int npb, int ncb) {
  constexpr int WC = NW == 16 ? 4 : 8;        // cout tiles per wave (128 couts; sixteen waves: 64)
  constexpr int WP = NW == 4 ? 8 : 4;         // pixel tiles per wave (128 / 64 pixels)
  constexpr bool DOUBLE = NW == 4;            // second fragment register set (512 registers per lane with one wave per SIMD)
  constexpr int CT_BLK = 16, PT_BLK = 16;     // 256 x 256
  constexpr int STAGE = 32 * 1024;            // one operand, one stage
  constexpr int NSW = 2, NSX = 3;
  constexpr int X_BASE = NSW * STAGE;
  constexpr int NPC = 32 / NW;                // 1-KiB pieces per wave, operand and stage
  constexpr int NM = WC * WP, NR = WC + WP;
  static_assert(NSW * STAGE + NSX * STAGE == 160 * 1024, "the rings take all of the CU's LDS");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cblk = slot % ncb;
  const int pblk = xcd * ((npb + 7) >> 3) + slot / ncb;
  if (pblk >= npb) return;

  // ------------------------------------------------------------------------------------------------ producer side
  // this wave's pieces: id = wave + NW * m of the 32 per operand and stage; weights: (cout tile id >> 1, k-half id & 1),
  // pixels: (pixel tile id >> 1, 8-pixel half id & 1) -- the half is the same for all pieces of a wave (NW is even)
  const int half = wave & 1;
  const int ccn = p.cin_s >> 5;               // 32-channel k-steps per tap (packed-weight index unit)
  const int cc2n = p.cin_s >> 6;              // 64-channel chunks
  const int n_st = p.kh * p.kw * cc2n;        // stages
  const long zero_off = reinterpret_cast<const unsigned char*>(g_big_zeros) - reinterpret_cast<const unsigned char*>(p.x);
  const u32x4* wsrc[NPC];
  int poff[NPC];
  unsigned vmask[NPC];
  const int csb = PAIR ? p.cin_s / Split<T>::NB : p.cin_s;         // channels per block of a split map
  const int xs = PAIR ? Split<T>::NS * csb : p.cin_s;              // stored channels per pixel of x
  const float inv_csb = 1.0f / (float)csb;
  const int kchunk = (lane & 7) ^ (4 * half + (((lane >> 3) >> 1) & 3));
  {
    const int q = lane >> 3;
#pragma unroll
    for (int m = 0; m < NPC; ++m) {
      const int t = (wave >> 1) + (NW / 2) * m;                      // cout tile / pixel tile of the block
      wsrc[m] = p.w + (size_t)min(cblk * CT_BLK + t, p.ctiles - 1) * p.ksteps * 64 + half * 64 + lane;
      const int pix = (pblk * PT_BLK + t) * 16 + 8 * half + q;
      const bool v = pix < p.npix;
      const int pc = v ? pix : 0;
      const int ox = pc % p.w_out;
      const int r = pc / p.w_out;
      const int oy = r % p.h_out;
      const int nn = r / p.h_out;
      const int py0 = oy * p.stride - p.pad, px0 = ox * p.stride - p.pad;
      poff[m] = ((nn * p.h_in * p.w_in + py0 * p.w_in + px0) * xs + (PAIR ? 0 : kchunk * 8)) * 2;
      unsigned mk = 0;
      for (int ky = 0; ky < p.kh; ++ky)
        for (int kx = 0; kx < p.kw; ++kx) {
          const bool ok = v && (unsigned)(py0 + ky * p.dil) < (unsigned)p.h_in && (unsigned)(px0 + kx * p.dil) < (unsigned)p.w_in;
          mk |= (ok ? 1u : 0u) << (ky * p.kw + kx);
        }
      vmask[m] = mk;
    }
  }
  // (tap, chunk) of the next stage to issue, advanced with wave-uniform scalar arithmetic; pixel and weight pieces of a
  // stage are issued half a stage apart, so each stream keeps its own cursor
  struct Cursor {
    int ky, kx, cc2, slot;
  };
  Cursor cx = {0, 0, 0, 0}, cw = {0, 0, 0, 0};
  auto advance = [&](Cursor& c, int nslots) {
    if (++c.kx == p.kw) {
      c.kx = 0;
      if (++c.ky == p.kh) {
        c.ky = 0;
        ++c.cc2;
      }
    }
    c.slot = (c.slot + 1 == nslots) ? 0 : c.slot + 1;
  };
  auto issue_w_piece = [&](int m) {
    const size_t w_ks = (size_t)((cw.ky * p.kw + cw.kx) * ccn + 2 * cw.cc2) * 64;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[m] + w_ks),
                                     (__attribute__((address_space(3))) void*)(smem + cw.slot * STAGE + (wave + NW * m) * 1024),
                                     16, 0, 0);
  };
  auto issue_x_piece = [&](int m) {
    const int tap = cx.ky * p.kw + cx.kx;
    int coff = cx.cc2 * 64;
    if (PAIR) {                               // K channel -> channel of the stored pixel (this lane's 8-channel group)
      const int c = cx.cc2 * 64 + kchunk * 8;
      const int b = (int)(((float)c + 0.5f) * inv_csb);
      coff = Split<T>::xcomp(b) * csb + (c - b * csb);
    }
    const int tap_off = ((cx.ky * p.w_in + cx.kx) * p.dil * xs + coff) * 2;
    const long off = ((vmask[m] >> tap) & 1u) ? (long)(poff[m] + tap_off) : zero_off;
    const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + off;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + X_BASE + cx.slot * STAGE + (wave + NW * m) * 1024),
                                     16, 0, 0);
  };
  auto issue_w_stage = [&]() {
#pragma unroll
    for (int m = 0; m < NPC; ++m) issue_w_piece(m);
    advance(cw, NSW);
  };
  auto issue_x_stage = [&]() {
#pragma unroll
    for (int m = 0; m < NPC; ++m) issue_x_piece(m);
    advance(cx, NSX);
  };

  // ------------------------------------------------------------------------------------------------ consumer side
  constexpr int NWC = CT_BLK / WC;            // wave grid: NWC cout groups x NW / NWC pixel groups
  const int wc = wave % NWC, wp = wave / NWC;
  const int j16 = lane & 15, g = lane >> 4;
  f32x4 acc[WC][WP];
#pragma unroll
  for (int c = 0; c < WC; ++c)
#pragma unroll
    for (int t = 0; t < WP; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int a_off = wc * WC * 2048 + lane * 16;
  int b_off[2];
  {
    const int q = j16 & 7, h = j16 >> 3;
#pragma unroll
    for (int c = 0; c < 2; ++c)
      b_off[c] = X_BASE + wp * WP * 2048 + 1024 * h + 128 * q + 16 * ((4 * c + g) ^ (4 * h + ((q >> 1) & 3)));
  }
  int rw = 0, rx = 0;                         // ring slots of the stage being read
  auto fetch_one = [&](int sw, int sx, int c, int qi, u32x4* fa, u32x4* fb) {
    if (qi < WP)
      fb[qi] = *reinterpret_cast<const u32x4*>(smem + sx * STAGE + b_off[c] + qi * 2048);
    else
      fa[qi - WP] = *reinterpret_cast<const u32x4*>(smem + sw * STAGE + a_off + ((qi - WP) * 2 + c) * 1024);
  };
  u32x4 a0[WC], b0[WP];
  u32x4 a1[DOUBLE ? WC : 1], b1[DOUBLE ? WP : 1];

  // One k-half: NM MFMAs from (ca, cb); between them, optionally, the NR fragment reads of the following half into
  // (na, nb) from ring slots (sw, sx) / k-half nh, and this wave's NPC pieces of stage s + 2 (PIECES 1: pixels, 2: weights).
  auto half_step = [&](const u32x4* ca, const u32x4* cb, u32x4* na, u32x4* nb, bool fetch, int sw, int sx, int nh,
                       auto pieces_tag, bool more) {
    constexpr int PIECES = decltype(pieces_tag)::value;
    constexpr int PG = NM / NPC;              // MFMAs per piece
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int c = i / WP, t = i % WP;
      acc[c][t] = mfma16(as_vec8<T>(ca[c]), as_vec8<T>(cb[t]), acc[c][t]);
      bool fence = false;
      if (DOUBLE && fetch && (i & 1) == 1 && (i >> 1) < NR) {
        fetch_one(sw, sx, nh, i >> 1, na, nb);
        fence = true;
      }
      if (i % PG == PG / 2 && i / PG < NPC) {
        if (more) {                             // wave-uniform: stage s + 2 exists
          if (PIECES == 1) issue_x_piece(i / PG);
          else issue_w_piece(i / PG);
        }
        fence = true;
      }
      if (fence) __builtin_amdgcn_sched_barrier(0);
    }
    if (PIECES == 1) advance(cx, NSX);
    else advance(cw, NSW);
    __builtin_amdgcn_sched_barrier(0);
  };
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  auto fetch_half = [&](int sw, int sx, int c, u32x4* fa, u32x4* fb) {
#pragma unroll
    for (int qi = 0; qi < NR; ++qi) fetch_one(sw, sx, c, qi, fa, fb);
  };
  auto next_slots = [&]() {
    rw = (rw + 1 == NSW) ? 0 : rw + 1;
    rx = (rx + 1 == NSX) ? 0 : rx + 1;
  };

  // ---- prologue: stages 0 and 1 on their way, stage 0 landed everywhere, its first fragments in registers
  issue_w_stage();
  issue_x_stage();
  if (n_st > 1) {
    issue_x_stage();
    issue_w_stage();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPC) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  fetch_half(0, 0, 0, a0, b0);

  // ---- stages.  ``more``: stage s + 2 exists (its pieces are issued during stage s); the last two stages issue nothing
  // and wait for everything.  One loop body (wave-uniform branches on ``more``): the accumulators stay in place.
  for (int s = 0; s < n_st; ++s) {
    const bool more = s + 2 < n_st;
    if (DOUBLE) {
      half_step(a0, b0, a1, b1, true, rw, rx, 1, P1{}, more);
    } else {
      half_step(a0, b0, a0, b0, false, 0, 0, 0, P1{}, more);
      fetch_half(rw, rx, 1, a0, b0);            // single register set: the second half's fragments replace the first's
    }
    // both halves of this stage are in registers (or consumed); own pieces of stage s + 1 landed
    if (more) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPC) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    next_slots();
    if (DOUBLE) {
      half_step(a1, b1, a0, b0, true, rw, rx, 0, P2{}, more);
    } else {
      half_step(a0, b0, a0, b0, false, 0, 0, 0, P2{}, more);
      fetch_half(rw, rx, 0, a0, b0);
    }
  }

  // ---- epilogue (as conv_gemm_kernel): lane holds channels ct*16 + 4g + {0..3} of pixel (tile, j); staged through LDS
  // in fp32 (row = one pixel x the wave's 128 couts, + 16 B pad) PP pixel tiles at a time, written as 16-byte chunks
  constexpr int ROWB = WC * 64 + 16;
  constexpr int PP = NW == 4 ? 4 : 2;
  constexpr int CH = WC * 2;
  static_assert(NW * PP * 16 * ROWB <= 160 * 1024, "epilogue staging does not fit");
  static_assert(WP % PP == 0, "WP must be a multiple of PP");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                               // every wave is done with the operand rings
  unsigned char* stg = smem + wave * (PP * 16 * ROWB);
  const int cout_base = (cblk * CT_BLK + wc * WC) * 16;
  const float* bias_ep = p.bias;       // the bias the store path still has to add
  // training-mode BatchNorm statistics of this wave's WP x 16 pixels (one chunk; whole or absent, as in conv_gemm_kernel)
  if (p.stats && (pblk * PT_BLK + wp * WP) * 16 < p.npix) {
    conv_gemm_stats_epilogue<T, WC, WP>(acc, p, cout_base, pblk * (PT_BLK / WP) + wp, j16, g);
    if (p.bias) bias_ep = nullptr;
  }
  if constexpr (!PAIR) {     // the store path shared with conv_gemm_kernel: every global read in front of the first store
    conv_gemm_staged_store<T, WC, WP, PP, RES2>(acc, p, stg, (pblk * PT_BLK + wp * WP) * 16, cout_base, bias_ep, lane, j16, g);
    return;
  }
  // (round 6: as conv_gemm_staged_store -- the bias once, a pass's residual components requested together after the pass is
  // staged (its accumulators are dead by then: 8 chunks x NC components fit), one wait per pass instead of one per load)
  constexpr int NIT = PP * 16 * CH / 64;
  constexpr int NSB = Split<T>::NS, NCC = Split<T>::NC;
  const int qc = lane % CH, pl0 = lane / CH;
  const int ch = cout_base + qc * 8;
  const bool ch_ok = ch < p.cout_s;
  f32x4 bb0 = (f32x4){0.f, 0.f, 0.f, 0.f}, bb1 = bb0;
  if (bias_ep && ch_ok) {
    bb0 = *reinterpret_cast<const f32x4*>(bias_ep + ch);
    bb1 = *reinterpret_cast<const f32x4*>(bias_ep + ch + 4);
  }
  auto epilogue_pass = [&](auto pass_tag) {
    constexpr int pass = decltype(pass_tag)::value;
#pragma unroll
    for (int tt = 0; tt < PP; ++tt)
#pragma unroll
      for (int c = 0; c < WC; ++c)
        *reinterpret_cast<f32x4*>(stg + (tt * 16 + j16) * ROWB + c * 64 + g * 16) = acc[c][pass * PP + tt];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int pix_base = (pblk * PT_BLK + wp * WP + pass * PP) * 16;
    u32x4 rs[NIT][NCC];
    if (p.has_res) {                                          // wave-uniform
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int pix = pix_base + it * (64 / CH) + pl0;
        size_t rpix = (size_t)(pix < p.npix ? pix : 0);
        if (p.res_ups && pix < p.npix) {
          const int ox = pix % p.w_out;
          const int r = pix / p.w_out;
          const int oy = r % p.h_out;
          const int nn = r / p.h_out;
          rpix = ((size_t)nn * (p.h_out >> 1) + (oy >> 1)) * (p.w_out >> 1) + (ox >> 1);
        }
        const uint16_t* rp = p.res + rpix * p.cout_s * NSB + ch;
#pragma unroll
        for (int k = 0; k < NCC; ++k) {
          rs[it][k] = (u32x4){0u, 0u, 0u, 0u};
          if (pix < p.npix && ch_ok) rs[it][k] = *reinterpret_cast<const u32x4*>(rp + k * p.cout_s);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): bias and this pass's residual, once (conv_gemm.h)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int pl = it * (64 / CH) + pl0;
      const int pix = pix_base + pl;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + pl * ROWB + qc * 32 + 16);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (bias_ep) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] += bb0[r];
          v[4 + r] += bb1[r];
        }
      }
      if (p.has_res) {
        float rsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = NCC - 1; k >= 0; --k) {          // smallest component first
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
      for (int k = 0; k < NCC; ++k) {                 // component k = round16 of what the previous ones left
        u32x4 comp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          comp[e] = pack2<T>(v[2 * e], v[2 * e + 1]);
          float q0, q1;
          unpack2<T>(comp[e], q0, q1);
          v[2 * e] -= q0;
          v[2 * e + 1] -= q1;
        }
        if (pix < p.npix && ch_ok) *reinterpret_cast<u32x4*>(yp + k * p.cout_s) = comp;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // staged rows are consumed before the next pass overwrites
  };
  static_assert(WP / PP == 2, "two staging passes");
  epilogue_pass(std::integral_constant<int, 0>{});
  epilogue_pass(std::integral_constant<int, 1>{});
}

template <typename T, int NW, bool PAIR = false, bool RES2 = false>
int launch_big(const ConvGemmArgs& a, hipStream_t s) {
  constexpr size_t smem = 160 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_big_kernel<T, NW, PAIR, RES2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("conv_gemm_big: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  const int npb = ceil_div(ceil_div(a.npix, 16), 16);
  const int ncb = ceil_div(a.ctiles, 16);
  const int grid = ceil_div(npb, 8) * 8 * ncb;
  hipLaunchKernelGGL((conv_gemm_big_kernel<T, NW, PAIR, RES2>), dim3(grid), dim3(NW * 64), smem, s, a, npb, ncb);
  return CGAN_OK;
}

}  // namespace

// whole 64-channel chunks, zero padding, <= 32 taps (validity mask), 32-bit byte offsets per lane (input below 1 GiB of
// 16-bit elements), and enough couts to fill the 256-row tile
bool conv_gemm_big_ok(const ConvGemmArgs& a) {
  return (a.cin_s & 63) == 0 && a.pad_mode != CGAN_PAD_REFLECT && a.kh * a.kw <= 32 && a.ctiles >= 12 &&
         (long)a.n * a.h_in * a.w_in * a.cin_s < (1L << 30) - (1L << 20);
}

// Eight waves (2 x 4 over the block tile, 128 couts x 64 pixels each, two per SIMD, 128 accumulator registers).  The
// four-wave form (one per SIMD, 128 x 128 per wave: 256 accumulator registers + a second fragment set) is kept in the
// template but not instantiated: hipcc (ROCm 7.2) cannot keep a 256-register accumulator in place -- it rotates tiles
// through a[0:3] with v_accvgpr_write / _read pairs around every MFMA (445 of them for 256 MFMAs in the loop body).
// development knob (cgan_debug_set_big_waves): 16 = the sixteen-wave form (4 x 4 waves of 64 couts x 64 pixels, four per SIMD at
// <= 128 registers: twice the waves to cover a wave's fragment reads and DMA issue, twice the fragment reads per MFMA)
CGAN_KNOB(int, g_big_waves, 8);
CGAN_DEV_ONLY(extern "C" void cgan_debug_set_big_waves(int v) { g_big_waves = v; })

int conv_gemm_big_launch(const ConvGemmArgs& a, int dtype, hipStream_t s) {
  if (g_big_waves == 16 && a.has_res != 3)
    return dtype == CGAN_F16 ? launch_big<F16, 16>(a, s) : launch_big<BF16, 16>(a, s);
  if (a.has_res == 3)      // the instantiation with the second epilogue map (two launches per train step: layer4's bottlenecks)
    return dtype == CGAN_F16 ? launch_big<F16, 8, false, true>(a, s) : launch_big<BF16, 8, false, true>(a, s);
  return dtype == CGAN_F16 ? launch_big<F16, 8>(a, s) : launch_big<BF16, 8>(a, s);
}

// split-precision forward on the 256 x 256 tile: ``a`` describes the conv over the K extent NB * csb; what is addressed with
// 32-bit byte offsets is the STORED map (NS * csb channels per pixel)
bool conv_gemm_big_pair_ok(const ConvGemmArgs& a, int dtype) {
  const int nb = cgan_split_blocks(dtype), ns = cgan_split_store_blocks(dtype);
  return (a.cin_s & 63) == 0 && (a.cin_s % nb) == 0 && a.pad_mode != CGAN_PAD_REFLECT && a.kh * a.kw <= 32 && a.ctiles >= 12 &&
         a.npix >= 16384 && a.stats == nullptr &&
         (long)a.n * a.h_in * a.w_in * (a.cin_s / nb) * ns < (1L << 30) - (1L << 20) &&
         (long)a.npix * a.cout_s * ns < (1L << 31);
}
int conv_gemm_big_pair_launch(const ConvGemmArgs& a, int dtype, hipStream_t s) {
  return dtype == CGAN_F16 ? launch_big<F16, 8, true>(a, s) : launch_big<BF16, 8, true>(a, s);
}
