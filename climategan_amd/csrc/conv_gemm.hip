// conv2d NHWC forward for WIDE layers (cin % 32 == 0, cout >= 64: the ResNet-101 / ASPP / depth / mask-decoder
// convolutions of the Masker) as an LDS-tiled implicit GEMM on MFMA 16x16x32, gfx950.
//
//   D[cout][pixel] = sum_k Wp[cout][k] * X[k][pixel],   k = tap * cin_s + c   (same operand roles as conv_mfma.hip)
//
// Workgroup = 4 waves, block tile = (WAVES_C*WC*16) couts x (WAVES_P*WP*16) pixels (128 x 256 by default), one MFMA
// k-step (32 channels of one tap) per pipeline stage, 3-stage ring in LDS filled by LDS-DMA
// (global_load_lds_dwordx4): both operands land in LDS already in MFMA FRAGMENT ORDER (1 KiB per 16x32 tile, lane l at
// byte 16*l), so every fragment read is a conflict-free linear ds_read_b128 and no register ever stages an operand.
//   * weights: pre-packed in fragment order -> a wave copies 1 KiB contiguous per (cout tile, k-step);
//   * pixels : lane (j = l&15, g = l>>4) fetches the 16 B "8 channels c0+8g.. of pixel j" of the tap-shifted input
//              pixel; zero padding reads a zero page, reflect padding / stride / dilation are index math.
// Per k-step a wave reads WC + WP fragments for WC*WP MFMAs (12 -> 32: 0.75 of the LDS read bandwidth at full MFMA
// rate); global->LDS traffic per k-step is 24 KiB per 2.1 MFLOP, so what matters is how much of it hits L2:
//   * the weights of a layer (<= a few MiB) always do; the block covers ALL couts when cout <= 256 (256 x 128 tile),
//     so the activations are then fetched from HBM exactly once;
//   * K order = channel chunk outer, tap inner: the 9 tap-shifted fetches of one 32-channel slice are issued back to
//     back and overlap almost entirely (L2 hits), instead of streaming the whole input once per tap;
//   * the grid is linearised so that one XCD (one L2) owns a CONTIGUOUS range of pixel blocks (halo rows of dilated
//     3x3 windows are shared between neighbouring blocks) and the cout blocks of a pixel block run back to back.
// Epilogue: accumulators are staged through LDS in fp32 and leave as coalesced 16-byte stores (residual reads
// likewise), bias / residual / activation applied in fp32 in the same order as the general kernel.
#include "conv_gemm.h"
#include <type_traits>

namespace {

__device__ __attribute__((aligned(16))) unsigned int g_gemm_zeros[4];

constexpr int NSTAGE = 3;
// development knob (cgan_debug_set_gemm_ws): 0 = automatic, 1 = never a K = 64 kernel, 5 / 6 = force conv_gemm_k64_kernel,
// 8 = force the 256 x 256 kernel of conv_gemm_big.hip, 10 = force the direct 1x1 kernel of conv1x1_direct.hip,
// 9 = automatic without those two and without the x-resident 1x1 kernel, 11 = force conv1x1_xres.hip, 12 = automatic
// without it, 13 = automatic without round 4's long-K 1x1 rule (those layers on the plain tiles), 14 = that rule for layer4's
// shapes only (same-box A/B of the rule: tools/gpu_ab_env.sh with CGAN_DEV_LIB=1 CGAN_DEBUG_GEMM_WS=13|14|0)
CGAN_KNOB(int, g_gemm_ws, 0);

__device__ __forceinline__ int reflect_i(int i, int n) {
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

// RES2: the instantiation that takes has_res == 3 (a second epilogue map, conv_gemm_staged_store) -- the 2-stage 128 x 128 tile only
template <typename T, int WAVES_C, int WC, int WP, bool REFLECT, int NS = NSTAGE, bool RES2 = false>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(ConvGemmArgs p, int npb, int ncb) {
  constexpr int WAVES_P = 4 / WAVES_C;
  constexpr int CT_BLK = WAVES_C * WC;
  constexpr int PT_BLK = WAVES_P * WP;
  constexpr int STAGE_BYTES = (CT_BLK + PT_BLK) * 1024;
  constexpr int W_PER_WAVE = CT_BLK / 4;
  constexpr int P_PER_WAVE = PT_BLK / 4;
  constexpr int DMA_PER_WAVE = W_PER_WAVE + P_PER_WAVE;
  static_assert(CT_BLK % 4 == 0 && PT_BLK % 4 == 0, "tiles must split evenly over the 4 waves");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: LDS-DMA bases stay in SGPRs
  const int j = lane & 15;
  const int g = lane >> 4;
  const int wc = wave % WAVES_C, wp = wave / WAVES_C;

  // block -> (pixel block, cout block): ids that differ by 8 share an XCD; each XCD owns a contiguous range of pixel
  // blocks, cout blocks vary fastest
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int cblk = slot % ncb;
  const int pblk = xcd * ((npb + 7) >> 3) + slot / ncb;
  if (pblk >= npb) return;

  // ---- per-lane coordinates of the pixel tiles this wave copies (tile i = wave + 4 m of the block)
  int pbase[P_PER_WAVE], py0[P_PER_WAVE], px0[P_PER_WAVE];
#pragma unroll
  for (int m = 0; m < P_PER_WAVE; ++m) {
    int pix = (pblk * PT_BLK + wave + 4 * m) * 16 + j;
    bool v = pix < p.npix;
    int pc = v ? pix : 0;
    int ox = pc % p.w_out;
    int r = pc / p.w_out;
    int oy = r % p.h_out;
    int nn = r / p.h_out;
    pbase[m] = nn * p.h_in * p.w_in * p.cin_s + g * 8;
    py0[m] = v ? oy * p.stride - p.pad : -(1 << 28);   // invalid pixels never pass the range test (zero pad) ...
    px0[m] = ox * p.stride - p.pad;
  }
  const int ccn = p.cin_s >> 5;   // k-steps per tap
  // byte offset of the zero page relative to x (select between two offsets, not two pointers: one v_cndmask pair)
  const long zero_off = reinterpret_cast<const unsigned char*>(g_gemm_zeros) - reinterpret_cast<const unsigned char*>(p.x);

  int i_ky = 0, i_kx = 0, i_cc = 0, i_buf = 0;   // (tap, channel chunk, ring slot) of the NEXT stage to issue
  // one LDS-DMA piece (1 KiB) of the next stage: pieces 0..W_PER_WAVE-1 are weight tiles, the rest pixel tiles
  auto issue_piece = [&](int piece) {
    unsigned char* buf = smem + i_buf * STAGE_BYTES;
    if (piece < W_PER_WAVE) {
      const int i = wave + 4 * piece;
      const int ct = min(cblk * CT_BLK + i, p.ctiles - 1);
      const u32x4* src = p.w + ((size_t)ct * p.ksteps + (i_ky * p.kw + i_kx) * ccn + i_cc) * 64 + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + i * 1024), 16, 0, 0);
    } else {
      const int m = piece - W_PER_WAVE;
      const int i = wave + 4 * m;
      int iy = py0[m] + i_ky * p.dil, ix = px0[m] + i_kx * p.dil;
      bool ok;
      if (REFLECT) {
        ok = py0[m] > -(1 << 27);
        iy = reflect_i(ok ? iy : 0, p.h_in);
        ix = reflect_i(ix, p.w_in);
      } else {
        ok = (unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in;
      }
      const long off = ok ? (long)(pbase[m] + (iy * p.w_in + ix) * p.cin_s + i_cc * 32) * 2 : zero_off;
      const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + off;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(buf + (CT_BLK + i) * 1024), 16, 0, 0);
    }
  };
  // advance to the following stage (taps innermost, channel chunks outermost) with selects only: the loop body must
  // stay one basic block.  The two stages issued past the end wrap to stage 0/1 and are never read.
  auto issue_advance = [&]() {
    const int kx1 = i_kx + 1;
    const bool wx = kx1 == p.kw;
    i_kx = wx ? 0 : kx1;
    const int ky1 = i_ky + (wx ? 1 : 0);
    const bool wy = ky1 == p.kh;
    i_ky = wy ? 0 : ky1;
    const int cc1 = i_cc + (wy ? 1 : 0);
    i_cc = cc1 == ccn ? 0 : cc1;
    i_buf = (i_buf + 1 == NS) ? 0 : i_buf + 1;
  };
  auto issue = [&]() {
#pragma unroll
    for (int q = 0; q < DMA_PER_WAVE; ++q) issue_piece(q);
    issue_advance();
  };

  f32x4 acc[WC][WP];
#pragma unroll
  for (int c = 0; c < WC; ++c)
#pragma unroll
    for (int t = 0; t < WP; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // The loop body is branch-free (stages past the end are issued anyway and ignored) so that the LDS-DMA issues can
  // be spread between the MFMAs: issued back to back right after the barrier they serialise every wave of the CU on
  // the vector-memory path before any MFMA starts.
  issue();
  if (NS > 2) issue();
  int r_buf = 0;
  for (int ks = 0; ks < p.ksteps; ++ks) {
    // stage ks has landed once at most the DMAs of the NS - 2 younger stages are still in flight (in-order completion)
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * DMA_PER_WAVE) : "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned char* buf = smem + r_buf * STAGE_BYTES;
    // the operand every MFMA of the first group needs goes first (LDS returns in order): the MFMAs start after
    // 1 + inner-count fragments instead of after all WC + WP
    u32x4 a[WC], b[WP];
#pragma unroll
    for (int t = 0; t < WP; ++t)
      b[t] = *reinterpret_cast<const u32x4*>(buf + (CT_BLK + wp * WP + t) * 1024 + lane * 16);
#pragma unroll
    for (int c = 0; c < WC; ++c) a[c] = *reinterpret_cast<const u32x4*>(buf + (wc * WC + c) * 1024 + lane * 16);
    // stage ks+2 refills the slot every wave finished reading before this barrier; its DMA pieces are issued one
    // at a time between groups of MFMAs (hard scheduling fences keep them there)
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

  // ---- epilogue: lane holds channels ct*16 + 4g + {0..3} of pixel (tile, j).  Staged through LDS (fp32, row =
  // one pixel x the wave's WC*16 couts, +16 B pad) PP pixel tiles at a time, then written as 16-byte chunks.
  constexpr int ROWB = WC * 64 + 16;
  constexpr int PP = (4 * 4 * 16 * ROWB <= NS * STAGE_BYTES && WP % 4 == 0) ? 4
                     : (4 * 2 * 16 * ROWB <= NS * STAGE_BYTES ? 2 : 1);                // pixel tiles per pass
  static_assert(4 * PP * 16 * ROWB <= NS * STAGE_BYTES, "epilogue staging does not fit");
  static_assert(WP % PP == 0, "WP must be a multiple of PP");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                               // every wave is done with the operand ring
  unsigned char* stg = smem + wave * (PP * 16 * ROWB);
  const int cout_base = (cblk * CT_BLK + wc * WC) * 16;
  const float* bias_ep = p.bias;       // the bias the store path still has to add
  // A wave's statistics chunk is whole or absent: npix is a multiple of the chunk (the dispatcher checked), but not
  // necessarily of the workgroup's pixel count -- the trailing waves of the last block own no pixel and no partial row.
  if (p.stats && (pblk * PT_BLK + wp * WP) * 16 < p.npix) {
    conv_gemm_stats_epilogue<T, WC, WP>(acc, p, cout_base, pblk * WAVES_P + wp, j, g);
    if (p.bias) bias_ep = nullptr;       // folded into the accumulators: the store path skips it (wave-uniform)
  }
  conv_gemm_staged_store<T, WC, WP, PP, RES2>(acc, p, stg, (pblk * PT_BLK + wp * WP) * 16, cout_base, bias_ep, lane, j, g);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised, persistent variant with K = 64 per stage and whole cache lines per pixel (round 2).
// The kernel above interleaves, in every wave, the LDS-DMA issues of the stage after next (each with its address
// arithmetic) with the MFMAs of the current stage; rocprofv3 counters put it at 31-33 % MFMA busy on the long-K layers,
// and no re-ordering at source level moves that.  Ablations of a producer / consumer split of the same loop showed why:
// the fill of the ring costs ~3 cycles per 128-byte line TOUCHED per CU, whatever the number of issuing waves, the ring
// depth or the path (LDS-DMA or registers), and whether the line is used in full -- a 1-KiB weight piece (8 contiguous
// lines) costs a wave ~67 cycles, a 1-KiB pixel piece of the K = 32 layout (16 pixels x 64 B: half of 16 different
// lines) ~217, so a 128 x 256 x 32 stage cannot be filled in less than ~1000 cycles against 544 cycles of MFMA work.
// Here:
//   * a workgroup is 8 waves on ONE CU (144 KiB of LDS): waves 4-7 = PRODUCERS (address arithmetic + LDS-DMA only, up
//     to two stages ahead, across tile boundaries: the workgroup walks a list of tiles and the next tile's first stages
//     are in flight during the epilogue of the current one), waves 0-3 = CONSUMERS (2 x 2 over the block tile, one per
//     SIMD): per k-half the WC * WP MFMAs of the current half from one register set, with the WC + WP fragment reads of
//     the next half slotted between them into the other set;
//   * a pixel piece is 8 pixels x 128 B (64 channels = two MFMA k-steps): 8 full lines.  A stage is 64 channels of one
//     tap: 2 KiB per cout tile + 2 KiB per pixel tile, 48 KiB for the 256 x 128 block, three stages = 144 KiB (no room
//     for an epilogue staging area: the accumulators are stored straight from registers, 8 bytes per lane; long-K layers
//     only, where the epilogue is < 2 % of a tile).
// Hand-over (one s_barrier per stage, numbered like the stages): a producer passes barrier j only after its share of stage
// j has landed (counted vmcnt: one younger stage may stay in flight); a consumer reads stage j between barrier j and
// barrier j + 1 and has its reads back (lgkmcnt(0)) before it arrives at barrier j + 1; the producers refill that slot
// (stage j + 3) only after barrier j + 1.  LDS-DMA data is visible to another wave's ds_read exactly under that sequence:
// issuing wave's counted vmcnt, then a barrier the reader has passed.
// Tiles: XCD x (blockIdx & 7) owns a contiguous range of (pixel block, cout block) tiles, cout blocks fastest, its
// workgroups take them round-robin: concurrently running workgroups of an XCD read neighbouring tiles.
// LDS image of a pixel piece (LDS-DMA writes lane l at byte 16 l): lane l = 8 q + s fetches pixel q (of the piece's 8)
// and the 16-byte channel chunk k = s ^ (4 h + ((q >> 1) & 3)), h = which half of the 16-pixel MFMA tile the piece is.
// The B fragment of k-half c is read back by lane (j, g) at 1024 h + 128 q + 16 ((4 c + g) ^ (4 h + ((q >> 1) & 3))),
// q = j & 7, h = j >> 3: every 16-lane group of the ds_read_b128 hits 16 distinct 16-byte bank slots (checked
// exhaustively); without the XOR the K-half selection alone would make it a 4-way conflict.
// K order: (64-channel chunk, tap, half) -- not the plain kernel's (32-channel chunk, tap): same products, another fp32
// summation order.  Measured (rocprofv3, bs 8, bf16): 80^2 512 -> 512 d4 337 -> 283 us, 2048 -> 256 d6 538 -> 459 us
// (1.16 PFLOP/s by events); short-K layers lose (one workgroup per CU: prologue / epilogue are not hidden by a second
// one), so the dispatcher takes it for 3x3 layers with >= 512 input channels only.
template <typename T, int WC, int WP>
__global__ __launch_bounds__(512, 2) void conv_gemm_k64_kernel(ConvGemmArgs p, int npb, int ncb, int tiles_total) {
  constexpr int CT_BLK = 2 * WC, PT_BLK = 2 * WP;
  constexpr int W_BYTES = CT_BLK * 2048, STAGE_BYTES = (CT_BLK + PT_BLK) * 2048;
  constexpr int NS = 3;
  constexpr int W_PER = CT_BLK / 2, P_PER = PT_BLK / 2, PIECES = W_PER + P_PER;   // 1-KiB pieces per producer wave
  static_assert(PIECES <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = gridDim.x >> 3;
  const int per = (tiles_total + 7) >> 3;
  const int t0 = xcd * per;
  const int t1 = min(t0 + per, tiles_total);
  const int my_n = (t1 - t0 > slot) ? (t1 - t0 - slot + nslot - 1) / nslot : 0;
  if (my_n <= 0) return;
  const int first = t0 + slot;
  const int cc2n = p.cin_s >> 6;                       // 64-channel chunks
  const int n_st = p.kh * p.kw * cc2n;                 // stages per tile
  const int n_it = my_n * n_st;

  if (wave >= 4) {
    // ------------------------------------------------------------------------------------------------ producers
    const int pw = wave - 4;
    const int half = pw & 1;                           // this wave's weight k-half and pixel-tile half
    const int ccn = p.cin_s >> 5;
    const long zero_off = reinterpret_cast<const unsigned char*>(g_gemm_zeros) - reinterpret_cast<const unsigned char*>(p.x);
    const int q = lane >> 3;
    const int kchunk = (lane & 7) ^ (4 * half + ((q >> 1) & 3));
    int poff[P_PER];
    unsigned vmask[P_PER];
    const u32x4* wtile[W_PER];
    auto set_tile = [&](int ti) {
      const int gt = first + ti * nslot;
      const int pblk = gt / ncb;
      const int cblk = gt - pblk * ncb;
#pragma unroll
      for (int m = 0; m < W_PER; ++m)
        wtile[m] = p.w + (size_t)min(cblk * CT_BLK + (pw >> 1) + 2 * m, p.ctiles - 1) * p.ksteps * 64 + half * 64 + lane;
#pragma unroll
      for (int m = 0; m < P_PER; ++m) {
        const int pix = (pblk * PT_BLK + (pw >> 1) + 2 * m) * 16 + 8 * half + q;
        const bool v = pix < p.npix;
        const int pc = v ? pix : 0;
        const int ox = pc % p.w_out;
        const int r = pc / p.w_out;
        const int oy = r % p.h_out;
        const int nn = r / p.h_out;
        const int py0 = oy * p.stride - p.pad, px0 = ox * p.stride - p.pad;
        poff[m] = ((nn * p.h_in * p.w_in + py0 * p.w_in + px0) * p.cin_s + kchunk * 8) * 2;
        unsigned mk = 0;
        for (int ky = 0; ky < p.kh; ++ky)
          for (int kx = 0; kx < p.kw; ++kx) {
            const bool ok = v && (unsigned)(py0 + ky * p.dil) < (unsigned)p.h_in && (unsigned)(px0 + kx * p.dil) < (unsigned)p.w_in;
            mk |= (ok ? 1u : 0u) << (ky * p.kw + kx);
          }
        vmask[m] = mk;
      }
    };
    int i_ky = 0, i_kx = 0, i_cc2 = 0, i_buf = 0, i_st = 0, i_tile = 0, issued = 0;
    auto issue_stage = [&]() {
      unsigned char* buf = smem + i_buf * STAGE_BYTES;
      const int tap = i_ky * p.kw + i_kx;                                   // wave-uniform
      const size_t w_ks = (size_t)(tap * ccn + 2 * i_cc2) * 64;
#pragma unroll
      for (int m = 0; m < W_PER; ++m)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wtile[m] + w_ks),
                                         (__attribute__((address_space(3))) void*)(buf + (pw + 4 * m) * 1024), 16, 0, 0);
      const int tap_off = ((i_ky * p.w_in + i_kx) * p.dil * p.cin_s + i_cc2 * 64) * 2;   // wave-uniform
#pragma unroll
      for (int m = 0; m < P_PER; ++m) {
        const long off = ((vmask[m] >> tap) & 1u) ? (long)(poff[m] + tap_off) : zero_off;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.x) + off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(buf + W_BYTES + (pw + 4 * m) * 1024), 16, 0, 0);
      }
      ++issued;
      i_buf = (i_buf + 1 == NS) ? 0 : i_buf + 1;
      if (++i_st == n_st) {
        i_st = 0; i_ky = 0; i_kx = 0; i_cc2 = 0;
        if (++i_tile < my_n) set_tile(i_tile);
      } else if (++i_kx == p.kw) {
        i_kx = 0;
        if (++i_ky == p.kh) {
          i_ky = 0;
          ++i_cc2;
        }
      }
    };
    set_tile(0);
    for (int s2 = 0; s2 < NS - 1; ++s2)
      if (issued < n_it) issue_stage();
    for (int j = 0; j < n_it; ++j) {
      if (issued - 1 - j >= NS - 2)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PIECES) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (issued < n_it) issue_stage();
    }
    return;
  }

  // -------------------------------------------------------------------------------------------------- consumers
  const int wc = wave & 1, wp = wave >> 1;
  const int j16 = lane & 15, g = lane >> 4;
  f32x4 acc[WC][WP];
#pragma unroll
  for (int c = 0; c < WC; ++c)
#pragma unroll
    for (int t = 0; t < WP; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4 a0[WC], b0[WP], a1[WC], b1[WP];
  int r_buf = 0;
  // byte offsets of this lane's fragments inside a stage: A (ctile a, half c) at ((wc*WC + a)*2 + c) KiB + 16 lane;
  // B (pixel tile b, half c) through the swizzle above
  const int a_off = wc * WC * 2048 + lane * 16;
  int b_off[2];
  {
    const int q = j16 & 7, h = j16 >> 3;
#pragma unroll
    for (int c = 0; c < 2; ++c)
      b_off[c] = W_BYTES + wp * WP * 2048 + 1024 * h + 128 * q + 16 * ((4 * c + g) ^ (4 * h + ((q >> 1) & 3)));
  }
  auto fetch_one = [&](const unsigned char* buf, int c, int qi, u32x4* na, u32x4* nb) {
    if (qi < WP)
      nb[qi] = *reinterpret_cast<const u32x4*>(buf + b_off[c] + qi * 2048);
    else
      na[qi - WP] = *reinterpret_cast<const u32x4*>(buf + a_off + ((qi - WP) * 2 + c) * 1024);
  };
  // One pipeline step = one k-half: the WC * WP MFMAs of the current half with the WC + WP fragment reads of the next
  // half between them.  ``next_half`` 1: the other half of the same stage; 0: the first half of the NEXT stage, after the
  // stage barrier; -1: nothing to fetch (a tile's last half).
  auto step = [&](const u32x4* ca, const u32x4* cb, u32x4* na, u32x4* nb, int next_half) {
    __builtin_amdgcn_sched_barrier(0);
    if (next_half == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every read of the stage being left is back
      __builtin_amdgcn_s_barrier();
      r_buf = (r_buf + 1 == NS) ? 0 : r_buf + 1;
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* buf = smem + r_buf * STAGE_BYTES;
    constexpr int NM = WC * WP, NR = WC + WP;
    constexpr int EVERY = NM / NR >= 2 ? 2 : 1;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      const int c = i / WP, t = i % WP;
      acc[c][t] = mfma16(as_vec8<T>(ca[c]), as_vec8<T>(cb[t]), acc[c][t]);
      if (next_half >= 0 && i % EVERY == EVERY - 1 && i / EVERY < NR) {
        fetch_one(buf, next_half, i / EVERY, na, nb);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto first_read = [&](bool advance) {                      // a tile's first half: nothing to overlap it with
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (advance) r_buf = (r_buf + 1 == NS) ? 0 : r_buf + 1;
    const unsigned char* buf = smem + r_buf * STAGE_BYTES;
#pragma unroll
    for (int qi = 0; qi < WC + WP; ++qi) fetch_one(buf, 0, qi, a0, b0);
  };
  auto epilogue = [&](int ti) {
    const int gt = first + ti * nslot;
    const int pblk = gt / ncb;
    const int cblk = gt - pblk * ncb;
    // round 6: the residual chunks of the whole tile are requested before the first store (a load in front of every
    // store compiled to s_waitcnt vmcnt(0) per chunk -- WC * WP serialized round trips, the stores' acknowledgements included)
    uint2 rv[WP][WC];
    if (p.has_res) {                                          // wave-uniform
#pragma unroll
      for (int t = 0; t < WP; ++t) {
        const int pix = (pblk * PT_BLK + wp * WP + t) * 16 + j16;
        size_t rbase = 0;
        if (pix < p.npix) {
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
#pragma unroll
        for (int c = 0; c < WC; ++c) {
          const int ch = (cblk * CT_BLK + wc * WC + c) * 16 + 4 * g;
          rv[t][c] = make_uint2(0u, 0u);
          if (pix < p.npix && ch < p.cout_s) rv[t][c] = *reinterpret_cast<const uint2*>(p.res + rbase + ch);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0) once, visible to the wait-count pass (conv_gemm.h)
    }
#pragma unroll
    for (int t = 0; t < WP; ++t) {
      const int pix = (pblk * PT_BLK + wp * WP + t) * 16 + j16;
#pragma unroll
      for (int c = 0; c < WC; ++c) {
        const int ch = (cblk * CT_BLK + wc * WC + c) * 16 + 4 * g;
        float v[4] = {acc[c][t][0], acc[c][t][1], acc[c][t][2], acc[c][t][3]};
        acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (pix >= p.npix || ch >= p.cout_s) continue;
        if (p.bias) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += p.bias[ch + r];
        }
        if (p.has_res) {
          float r0, r1;
          unpack2<T>(rv[t][c].x, r0, r1);
          v[0] = cgan_res_apply(v[0], r0, p.has_res); v[1] = cgan_res_apply(v[1], r1, p.has_res);
          unpack2<T>(rv[t][c].y, r0, r1);
          v[2] = cgan_res_apply(v[2], r0, p.has_res); v[3] = cgan_res_apply(v[3], r1, p.has_res);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = act_apply(v[r], p.act, p.slope);
          if (ch + r >= p.cout) v[r] = 0.f;
        }
        u32x2 o;
        o.x = pack2<T>(v[0], v[1]);
        o.y = pack2<T>(v[2], v[3]);
        CGAN_ST_STREAM(o, reinterpret_cast<u32x2*>(p.y + (size_t)pix * p.cout_s + ch));
      }
    }
  };

  for (int ti = 0; ti < my_n; ++ti) {
    first_read(ti > 0);
    for (int st = 0; st + 1 < n_st; ++st) {
      step(a0, b0, a1, b1, 1);
      step(a1, b1, a0, b0, 0);
    }
    step(a0, b0, a1, b1, 1);
    step(a1, b1, a0, b0, -1);
    epilogue(ti);
  }
}

template <typename T, int WC, int WP>
int launch_k64(const ConvGemmArgs& a, hipStream_t s) {
  constexpr int CT_BLK = 2 * WC, PT_BLK = 2 * WP;
  constexpr size_t smem = (size_t)3 * (CT_BLK + PT_BLK) * 2048;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_k64_kernel<T, WC, WP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("conv_gemm_k64: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  const int npb = ceil_div(ceil_div(a.npix, 16), PT_BLK);
  const int ncb = ceil_div(a.ctiles, CT_BLK);
  const int tiles = npb * ncb;
  const int per = ceil_div(tiles, 8);
  const int nslot = per < 32 ? per : 32;
  hipLaunchKernelGGL((conv_gemm_k64_kernel<T, WC, WP>), dim3(8 * nslot), dim3(512), smem, s, a, npb, ncb, tiles);
  return CGAN_OK;
}
// the K = 64 kernel's preconditions (beyond conv_gemm_applicable): whole 64-channel chunks, zero padding
// (32-bit BYTE offsets per lane: the input must stay below 2 GiB)
bool k64_ok(const ConvGemmArgs& a) {
  return (a.cin_s & 63) == 0 && a.pad_mode != CGAN_PAD_REFLECT && a.kh * a.kw <= 32 &&
         (long)a.n * a.h_in * a.w_in * a.cin_s < (1L << 30) - (1L << 20);
}


template <typename T, int WAVES_C, int WC, int WP, bool REFLECT, int NS = NSTAGE, bool RES2 = false>
int launch_cfg2(const ConvGemmArgs& a, hipStream_t s) {
  constexpr int WAVES_P = 4 / WAVES_C;
  constexpr int CT_BLK = WAVES_C * WC, PT_BLK = WAVES_P * WP;
  constexpr size_t smem = (size_t)NS * (CT_BLK + PT_BLK) * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<T, WAVES_C, WC, WP, REFLECT, NS, RES2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  const int npb = ceil_div(ceil_div(a.npix, 16), PT_BLK);
  const int ncb = ceil_div(a.ctiles, CT_BLK);
  const int grid = ceil_div(npb, 8) * 8 * ncb;
  hipLaunchKernelGGL((conv_gemm_kernel<T, WAVES_C, WC, WP, REFLECT, NS, RES2>), dim3(grid), dim3(256), smem, s, a, npb, ncb);
  return CGAN_OK;
}

template <typename T, int WAVES_C, int WC, int WP, int NS = NSTAGE>
int launch_cfg(const ConvGemmArgs& a, hipStream_t s) {
  return a.pad_mode == CGAN_PAD_REFLECT ? launch_cfg2<T, WAVES_C, WC, WP, true, NS>(a, s)
                                        : launch_cfg2<T, WAVES_C, WC, WP, false, NS>(a, s);
}

CGAN_KNOB(int, g_gemm_cfg, 0);   // development knob (tools/bench_conv.py): 0 = automatic, 1..4 = force a block tile

enum { KIND_PLAIN = 0, KIND_K64_256x128, KIND_K64_128x256, KIND_BIG, KIND_DIRECT, KIND_XRES };
struct Choice {
  int kind;
  int cfg;    // KIND_PLAIN: 1 = <1,4,4> (64 couts x 256 pixels), 2 = <2,8,4> (256 x 128), 3 = <2,4,8> (128 x 256),
              //             4 = <2,4,4> (128 x 128), 5 = <2,4,4> with a 2-stage ring
};

// the kernel (and block tile) a descriptor runs: a pure function of the shape (and, in the dev build, of the knobs)
Choice choose(const ConvGemmArgs& a, bool bf16) {
  const int ptiles = ceil_div(a.npix, 16);
  const bool no_big = g_gemm_ws == 9;
  const bool no_xres = g_gemm_ws == 9 || g_gemm_ws == 12;
  const int ws = (no_big || no_xres || g_gemm_ws == 13 || g_gemm_ws == 14) ? 0 : g_gemm_ws;     // 13: automatic as before round 4's long-K 1x1 rule
  switch (ws) {
    case 5: if (k64_ok(a)) return {KIND_K64_256x128, 0}; break;         // K = 64 stages, 256 couts x 128 pixels
    case 6: if (k64_ok(a)) return {KIND_K64_128x256, 0}; break;         // K = 64 stages, 128 x 256
    case 8: if (conv_gemm_big_ok(a)) return {KIND_BIG, 0}; break;
    case 10: if (conv1x1_allc_ok(a)) return {KIND_DIRECT, 0}; break;
    case 11: if (conv1x1_xres_ok(a)) return {KIND_XRES, 0}; break;
    default: break;
  }
  // (Shape-only: fp16 and bf16 take the same kernel.  Until round 3 the specialised kernels were bf16-only because an fp16
  // apply_events fixture had been captured with the plain kernel's fp32 summation order.)
  const bool automatic = ws == 0 && g_gemm_cfg == 0;
  (void)bf16;
  // 1x1 layers with <= 64 output channels: activations straight from global memory into the B-fragment registers of the
  // wave that owns the pixels, all couts per workgroup (conv1x1_direct.hip).  Same-box A/B (tools/gpu_ab_conv.sh): it wins
  // there (256 -> 64 at 8 x 160^2 50 -> 37 us, 128 -> 64 14.7 -> 10.2 us, their data gradients 48 -> 38 us) and loses from 128
  // couts up: at 1024 input channels the B-fragment loads are 16 rows x 64 B at a 2-KiB stride per instruction (1024 -> 256:
  // 55 -> 68 us with all 16 cout tiles per wave, 126 us with a 2 x 4 wave grid) -- the row-strided half-line pattern costs
  // more on the vector-memory path than the LDS-DMA pieces do
  if (automatic && !no_big && conv1x1_allc_ok(a) && a.ctiles <= 4 && a.npix >= 16384) return {KIND_DIRECT, 0};
  // short-K 1x1 layers with >= 256 couts (bottleneck expands and the reduce layers' data gradients): the activation tile
  // resident in LDS, weights straight into registers, all couts per workgroup (conv1x1_xres.hip)
  if (automatic && !no_xres && conv1x1_xres_ok(a) && a.ctiles >= 16 && a.npix >= 16384) return {KIND_XRES, 0};
  // 256 x 256 block tiles (conv_gemm_big.hip) halve the L2 -> LDS fill per FLOP.  Same-box A/B over the step's shapes: they
  // win where K is long enough to amortise a prologue and an epilogue that nothing overlaps with one workgroup per CU
  // -- 512 -> 512 3x3 at 8 x 80^2 289 -> 256 us, at 4 x 80^2 163 -> 135 us, 2048 -> 512 1x1 181 -> 157 us -- and lose on the
  // short-K bottleneck layers (256 -> 1024 1x1: 68 -> 118 us, 256 -> 256 3x3: 72 -> 84 us at 200 tiles for 256 CUs)
  // Round 4, cold operands (tools/micro/gemm_yardstick2.py, hipBLASLt beside it): the 1x1 layers with >= 1024 input
  // channels too -- 2048 -> 512 182 -> 114 us (hipBLASLt 116), 1024 -> 256 42.8 -> 35.2 (37.3), 1024 -> 2048 304 -> 226
  // (244), 1024 -> 512 91 -> 65 (58), and 512 -> 2048 178 -> 147 (178) -- now that the kernel writes the BatchNorm
  // statistics from its epilogue as well
  const bool long_1x1 = a.kh * a.kw == 1 && g_gemm_ws != 13 &&
                        (g_gemm_ws == 14 ? (a.cin_s >= 2048 || (a.cin_s >= 512 && a.ctiles >= 128))
                                         : (a.cin_s >= 1024 || (a.cin_s >= 512 && a.ctiles >= 128)));
  if (automatic && !no_big && conv_gemm_big_ok(a) && a.npix >= 16384 &&
      ((a.cin_s * a.kh * a.kw >= 4096 && a.cin_s >= 512 && !(a.kh * a.kw >= 9 && a.cin_s >= 2048)) || long_1x1))
    return {KIND_BIG, 0};
  // long-K 3x3 layers (>= 512 input channels: ASPP's 2048 -> 256): the K = 64 / whole-line producer / consumer kernel,
  // 14-16 % faster than the plain one there (rocprofv3, bs 8: 2048 -> 256 d6 538 -> 459 us); everywhere else it is slower
  // (one workgroup per CU: short K loops are all prologue / epilogue; 256 -> 256 d2 76 -> 79 us)
  if (automatic && k64_ok(a) && a.kh * a.kw >= 9 && a.cin_s >= 512 && a.npix >= 16384) return {KIND_K64_256x128, 0};
  switch (g_gemm_cfg) {
    case 1: return {KIND_PLAIN, 1};
    case 2: if (a.ctiles <= 16) return {KIND_PLAIN, 2}; break;
    case 3: return {KIND_PLAIN, 3};
    case 4: return {KIND_PLAIN, 4};
    case 5: return {KIND_PLAIN, 5};      // 128 x 128 with a 2-stage ring (32 KiB: 4 workgroups per CU)
    default: break;
  }
  if (a.ctiles <= 4) return {KIND_PLAIN, 1};                                    // 64 couts x 256 pixels
  // 1x1 layers with a short K (bottleneck expands, K <= 512): the workgroup is all prologue / epilogue, and 128 x 128
  // blocks (twice as many, half the epilogue each) overlap them better: 256 -> 1024 at 16 x 80^2: 175 -> 120 us,
  // 64 -> 256 at 16 x 160^2: 128 -> 80 us (rocprofv3 kernel durations, tools/prof_conv.sh)
  // (with a 2-stage ring: 32 KiB per workgroup, so four of them share a CU and cover each other's prologue / epilogue:
  // 64 -> 256 at 8 x 160^2 41 -> 37 us, 256 -> 1024 at 8 x 80^2 61.3 -> 60.4 us)
  if (a.kh * a.kw == 1 && a.ksteps <= 16) return {KIND_PLAIN, 5};
  // all couts in one block (activations read once) when cout <= 256 and the grid still fills the chip
  if (a.ctiles > 8 && a.ctiles <= 16 && ceil_div(ptiles, 8) >= 384) return {KIND_PLAIN, 2};
  // 128 couts x 256 pixels while that still gives every CU a couple of workgroups, else 128 x 128
  if ((long)ceil_div(ptiles, 16) * ceil_div(a.ctiles, 8) >= 384) return {KIND_PLAIN, 3};
  return {KIND_PLAIN, 4};
}

template <typename T>
int launch(const ConvGemmArgs& a, hipStream_t s) {
  const Choice ch = choose(a, std::is_same<T, BF16>::value);
  switch (ch.kind) {
    case KIND_K64_256x128: return launch_k64<T, 8, 4>(a, s);
    case KIND_K64_128x256: return launch_k64<T, 4, 8>(a, s);
    case KIND_BIG: return conv_gemm_big_launch(a, T::id, s);
    case KIND_DIRECT: return conv1x1_allc_launch(a, T::id, s);
    case KIND_XRES: return conv1x1_xres_launch(a, T::id, s);
    default: break;
  }
  switch (ch.cfg) {
    case 1: return launch_cfg<T, 1, 4, 4>(a, s);
    case 2: return launch_cfg<T, 2, 8, 4>(a, s);
    case 3: return launch_cfg<T, 2, 4, 8>(a, s);
    case 5:
      if (a.has_res == 3) return launch_cfg2<T, 2, 4, 4, false, 2, true>(a, s);      // (conv_gemm_res2_ok: zero padding)
      return launch_cfg<T, 2, 4, 4, 2>(a, s);
    default: return launch_cfg<T, 2, 4, 4>(a, s);
  }
}

}  // namespace

bool conv_gemm_applicable(const CganConvDesc* d) {
  const int cin_s = cgan_cs(d->c_in), cout_s = cgan_cs(d->c_out);
  const long npix = (long)d->n * d->h_out * d->w_out;
  const long in_elems = (long)d->n * d->h_in * d->w_in * cin_s;
  return (cin_s % 32) == 0 && cout_s >= 64 && !d->in_upsample && npix >= 2048 && in_elems < (1L << 31) - (1L << 20) &&
         (long)ceil_div((int)ceil_div((int)npix, 16), 8) * ceil_div(ceil_div(cout_s, 16), 8) >= 128;
}

CGAN_DEV_ONLY(extern "C" void cgan_debug_set_gemm_cfg(int v) { g_gemm_cfg = v; })
CGAN_DEV_ONLY(extern "C" void cgan_debug_set_gemm_ws(int v) { g_gemm_ws = v; })

int conv_gemm_launch(const ConvGemmArgs& a, int dtype, hipStream_t s) {
  return dtype == CGAN_F16 ? launch<F16>(a, s) : launch<BF16>(a, s);
}

bool conv_gemm_res2_ok(const ConvGemmArgs& a, int dtype) {
  const Choice ch = choose(a, dtype == CGAN_BF16);
  return (ch.kind == KIND_PLAIN && ch.cfg == 5 && a.pad_mode != CGAN_PAD_REFLECT) || ch.kind == KIND_BIG;
}

int conv_gemm_stats_chunk_pixels(const ConvGemmArgs& a, int dtype) {
  const Choice ch = choose(a, dtype == CGAN_BF16);
  if (ch.kind == KIND_XRES) return a.npix % 128 == 0 ? 128 : 0;   // a wave's 8 pixel tiles
  if (ch.kind == KIND_BIG) return a.npix % 64 == 0 ? 64 : 0;      // eight waves of 128 couts x 64 pixels
  if (ch.kind != KIND_PLAIN) return 0;
  const int ppb = ch.cfg == 3 ? 128 : 64;        // WP x 16 pixels of a wave: <.,.,8> -> 128, <.,.,4> -> 64
  return a.npix % ppb == 0 ? ppb : 0;
}
