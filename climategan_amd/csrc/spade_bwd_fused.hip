// Fused backward of SPADE's hidden map (round 5; reference climategan/norms.py:163-186 under autograd: the gradient path
//   d(gamma | beta) -> mlp_gamma / mlp_beta (3x3, 128 -> C each) -> ReLU -> mlp_shared (3x3, cond -> 128)).
//
// Until round 5 the backward re-materialised the 128-channel hidden map h at full resolution (a conv launch), ran the data
// gradient of the gamma||beta conv into a second 128-channel map dh (masked by h > 0 in its epilogue), and ran mlp_shared's
// weight gradient over dh: three 128-channel maps through HBM per SPADE layer (105 MB per image each at 640 x 640), 69
// launches per train step.  Here ONE kernel per layer
//   * computes dh = conv3x3^T(dgb) for a 16 x 16 pixel tile x 64 hidden channels with the LDS-tiled 3x3 loop of
//     conv3x3_lds.hip (dgb halo chunks and operator fragments by LDS-DMA, MFMA 16x16x32),
//   * RE-COMPUTES the hidden tile from the 3-channel conditioning image in registers (the folded-tap form of
//     conv3x3_c4_kernel: two MFMAs per 16 x 16 block) and masks dh with h > 0 -- h is never read,
//   * contracts the masked dh tile with the conditioning image's 3 x 3 x 4 neighbourhood over the tile's 256 pixels
//     (dh staged 16-bit in LDS row-major and read back with the transposing LDS read, like conv_wgrad.hip): mlp_shared's weight
//     and bias gradient, accumulated in registers across the tiles a workgroup walks -- dh is never written.
// A second small kernel sums the workgroups' partial [64 hidden][36 + 1] blocks in workgroup order (deterministic).
// Conditioning images of <= 4 channels only (the Painter: x (1 - m), painter.py:149-168); the SPADE mask decoder's 15-channel
// conditioning map, a conditioning map that wants a gradient itself, and maps below 80 x 80 keep the unfused path.
#include "cgan_common.h"
#include <type_traits>

namespace {

constexpr int TW = 16, TH = 16, WAVES = 4, PT = TH / WAVES, NCT = 4;
constexpr int HPW = TW + 2, HPH = TH + 2, HP = HPH * HPW;
constexpr int XDMA = (HP * 4 + 63) / 64;
constexpr int XBUF_BYTES = XDMA * 1024;
constexpr int STAGE_BYTES = 3 * NCT * 1024;
constexpr int KLOOP_BYTES = 2 * XBUF_BYTES + 2 * STAGE_BYTES;
constexpr int SEG_BYTES = ((HP * 8 + 1023) / 1024) * 1024;        // 18 x 18 x 8 B, rounded to 1 KiB
constexpr int SLAB_BYTES = 256 * 128;                              // [256 pixels][64 x 2 B]
constexpr int EPI_BYTES = SEG_BYTES + 2 * SLAB_BYTES;
constexpr int WORK_BYTES = KLOOP_BYTES > EPI_BYTES ? KLOOP_BYTES : EPI_BYTES;
constexpr int WSH_BYTES = NCT * 64 * 32 + NCT * 64 * 16;           // mlp_shared's folded A fragments (a0 | a1) + bias quads, per lane
constexpr int SMEM_BYTES = WORK_BYTES + WSH_BYTES;
constexpr int PART_COLS = 64;                                      // columns of a partial row: 36 (tap, c4) + bias at 48

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct SpadeHidBwdArgs {
  const uint16_t* dgb;     // [n][h][w][gs]
  const u32x4* w_dg;       // data-gradient operator of the gamma||beta conv: rows = hidden, [ct][tap * nq + q][lane]
  const uint16_t* seg;     // [n][h][w][seg_cs]: the conditioning image at (h, w), channels 0..3 read
  const u32x4* w_sh;       // mlp_shared's packed forward weights (one k-step per tap: cin_p = 32)
  const float* b_sh;       // its bias, padded to whole cout tiles
  float* part;             // [workgroups.x][2][64 hidden][PART_COLS] fp32
  int n, h, w, gs, cin_p, ksteps_dg, ksteps_sh, seg_cs, ctiles_dg, ctiles_sh, ntiles;
  int tail_r;              // 1 / 2: the last 32-channel chunk holds 8 / 16 live channels and runs as dense tail passes (0: off)
};

__device__ __attribute__((aligned(16))) unsigned int g_sb_zeros[64];

__device__ __forceinline__ int xq_addr(int q, int slot) { return q * 64 + ((slot ^ ((q >> 1) & 3)) << 4); }   // (bits 1..2: conv3x3_lds.hip)
__device__ __forceinline__ int swz(int r) { return ((r & 3) << 1) | ((r >> 2) & 1); }

// the 32 pixel rows q0 .. q0+31 of 16-channel tile `tile` of a [pixel][64 ch] slab (row q keeps 16-byte chunk c in slot
// c ^ swz(q & 7)): an MFMA operand fragment with the channel as row / column index and the pixel as K (conv_wgrad.hip)
__device__ __forceinline__ u32x4 tr_frag_at(const unsigned char* slab, int q0, int tile, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int k = i >> 2;
  const int chunk = tile * 2 + ((i & 3) >> 1);
  const int qa = q0 + 8 * g + k, qb = qa + 4;
  const unsigned char* a0 = slab + qa * 128 + ((chunk ^ swz(qa & 7)) << 4) + (i & 1) * 8;
  const unsigned char* a1 = slab + qb * 128 + ((chunk ^ swz(qb & 7)) << 4) + (i & 1) * 8;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  u32x4 r;
  r[0] = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  r[1] = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
  r[2] = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
  r[3] = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
  return r;
}

template <typename T>
__device__ __forceinline__ u32x4 ones_frag() {
  constexpr uint32_t one = std::is_same<T, BF16>::value ? 0x3f803f80u : 0x3c003c00u;
  return (u32x4){one, one, one, one};
}

#define SB_BARRIER()                                               \
  do {                                                             \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                                  \
    asm volatile("" ::: "memory");                                 \
  } while (0)

template <typename T>
__global__ __launch_bounds__(WAVES * 64, 2) void spade_hidden_bwd_kernel(SpadeHidBwdArgs p) {
  const u32x4* zero_page = reinterpret_cast<const u32x4*>(g_sb_zeros);
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* xbuf = smem;
  unsigned char* wbuf = smem + 2 * XBUF_BYTES;
  // the epilogue's view of the same memory
  u32x2* segh = reinterpret_cast<u32x2*>(smem);                    // [18 * 18] conditioning halo, channels 0..3
  unsigned char* dhs = smem + SEG_BYTES;                           // masked dh, [256 px][64 hidden]
  unsigned char* cis = dhs + SLAB_BYTES;                           // conditioning neighbourhood, [256 px][(tap, c4) | zeros]

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int g = lane >> 4;
  const int ct0 = blockIdx.y * NCT;                                // first hidden-channel tile of this workgroup
  const int nq = p.cin_p / 32;
  const int tiles_x = (p.w + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;

  // mlp_shared's folded A fragments (conv3x3_c4_kernel: k-step 0 = taps 0..7 x 4 channels, k-step 1 = tap 8) and bias quads of this
  // workgroup's four channel tiles, once per workgroup into LDS (per lane 32 + 16 bytes per tile): the epilogue of every tile reads
  // them back instead of waiting on L2 (their registers would not survive the K loop: 48 of the 256)
  unsigned char* wsh = smem + WORK_BYTES;
  if (wave < NCT) {
    const int c = wave;
    const int ct = min(ct0 + c, p.ctiles_sh - 1);
    const u32x2* wt = reinterpret_cast<const u32x2*>(p.w_sh + (size_t)ct * p.ksteps_sh * 64);
    const u32x2 wlo = wt[((2 * g) * 64 + j) * 2], whi = wt[((2 * g + 1) * 64 + j) * 2];
    const u32x2 wlast = wt[(8 * 64 + j) * 2];
    *reinterpret_cast<u32x4*>(wsh + (c * 64 + lane) * 32) = (u32x4){wlo[0], wlo[1], whi[0], whi[1]};
    *reinterpret_cast<u32x4*>(wsh + (c * 64 + lane) * 32 + 16) = g == 0 ? (u32x4){wlast[0], wlast[1], 0u, 0u} : (u32x4){0u, 0u, 0u, 0u};
    *reinterpret_cast<f32x4*>(wsh + NCT * 64 * 32 + (c * 64 + lane) * 16) = *reinterpret_cast<const f32x4*>(p.b_sh + ct * 16 + g * 4);
  }
  // this workgroup's share of mlp_shared's gradient: wave w owns hidden rows 16 w .. 16 w + 15 of the 64, columns = three
  // 16-wide tiles of (tap, c4) + the all-ones column of the bias
  f32x4 accs[3], accb;
#pragma unroll
  for (int b = 0; b < 3; ++b) accs[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accb = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tile_i = blockIdx.x; tile_i < p.ntiles; tile_i += gridDim.x) {
    int tile = tile_i;
    const int txi = tile % tiles_x;
    tile /= tiles_x;
    const int tyi = tile % tiles_y;
    const int n = tile / tiles_y;
    const int ty0 = tyi * TH, tx0 = txi * TW;

    auto issue_x = [&](int q) {
      unsigned char* dst = xbuf + (q & 1) * XBUF_BYTES;
      for (int i = wave; i < XDMA; i += WAVES) {
        const int idx = i * 64 + lane;
        const int pix = idx >> 2, spos = idx & 3;
        const int slot = spos ^ ((pix >> 1) & 3);
        const int py = pix / HPW, px = pix - py * HPW;
        const int yy = ty0 - 1 + py, xx = tx0 - 1 + px;
        const int ch = q * 32 + slot * 8;
        const u32x4* src = zero_page;
        if (pix < HP && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w && ch < p.gs)
          src = reinterpret_cast<const u32x4*>(p.dgb + (((size_t)n * p.h + yy) * p.w + xx) * p.gs + ch);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      }
    };
    // Stages: every FULL 32-channel chunk q runs three dx stages (3 dy taps x NCT operator fragments each).  A last chunk that
    // holds only 8 / 16 live channels (2C = 40 / 80: the Painter's 640 x 640 and 320 x 320 layers) would multiply 24 / 16 zero
    // channels per tap -- it runs as DENSE TAIL passes instead: one k-step = 4 / 2 taps x the live 8-channel groups, lane group
    // g takes tap tp * G + g / r (and channel group g % r): 3 / 5 passes instead of 9 (K = 12 / 23 k-steps instead of 18 / 27).
    // Its operator fragments are gathered per lane from the standard pack by the same LDS-DMA (16 bytes per lane from the
    // k-step of the lane's own tap), its B fragments are per-lane reads of the halo chunk at the lane's own tap shift.
    const int tr = p.tail_r;
    const int nqf = tr ? nq - 1 : nq;                   // full chunks
    const int tgp = tr ? 4 / tr : 1;                    // taps per tail pass
    const int ntp = tr ? (9 + tgp - 1) / tgp : 0;       // tail passes, three to a stage (12 operator pieces = a stage's 12 KiB)
    const int nts = (ntp + 2) / 3;                      // tail stages
    const int nst = 3 * nqf + nts;
    const int t_cg = tr == 2 ? (g & 1) : 0;             // this lane's live channel group in a tail pass
    const int t_tl = tr == 2 ? (g >> 1) : g;            // ... and its tap within the pass
    auto issue_w = [&](int s) {
      unsigned char* dst = wbuf + (s & 1) * STAGE_BYTES;
      if (s >= 3 * nqf) {
        const int tp0 = (s - 3 * nqf) * 3;
#pragma unroll
        for (int i0 = 0; i0 < 3 * NCT; i0 += WAVES) {
          const int i = i0 + wave;
          const int pl = i / NCT, c = i - pl * NCT;     // pass of the stage, operator tile
          if (i < 3 * NCT && tp0 + pl < ntp) {
            const int tap = (tp0 + pl) * tgp + t_tl;
            const int ct = min(ct0 + c, p.ctiles_dg - 1);
            const u32x4* src = tap < 9 ? p.w_dg + ((size_t)ct * p.ksteps_dg + tap * nq + (nq - 1)) * 64 + t_cg * 16 + j : zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
          }
        }
        return;
      }
      const int q = s / 3, dx = s - q * 3;
#pragma unroll
      for (int i0 = 0; i0 < 3 * NCT; i0 += WAVES) {
        const int i = i0 + wave;
        if (i < 3 * NCT) {
          const int dy = i / NCT, c = i - dy * NCT;
          const int ct = min(ct0 + c, p.ctiles_dg - 1);
          const u32x4* src = p.w_dg + ((size_t)ct * p.ksteps_dg + (dy * 3 + dx) * nq + q) * 64 + lane;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        }
      }
    };

    SB_BARRIER();                    // the previous tile's epilogue is done with the shared memory
    issue_x(0);
    issue_w(0);
    // the conditioning halo's two pixels of this thread, requested now and parked in registers until the epilogue stores them
    u32x2 sv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pix = threadIdx.x + u * WAVES * 64;
      const int py = pix / HPW, px = pix - py * HPW;
      const int yy = ty0 - 1 + py, xx = tx0 - 1 + px;
      sv[u] = (u32x2){0u, 0u};
      if (pix < HP && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w)
        sv[u] = *reinterpret_cast<const u32x2*>(p.seg + (((size_t)n * p.h + yy) * p.w + xx) * p.seg_cs);
    }
    f32x4 acc[NCT][PT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int t = 0; t < PT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- dh (before the mask) = the 3x3 data gradient over the dgb halo: conv3x3_lds_kernel's loop
    for (int q = 0; q < nqf; ++q) {
      const unsigned char* xb = xbuf + (q & 1) * XBUF_BYTES;
      for (int dx = 0; dx < 3; ++dx) {
        const int s = q * 3 + dx;
        SB_BARRIER();
        if (dx == 0 && q + 1 < nq) issue_x(q + 1);
        if (s + 1 < nst) issue_w(s + 1);
        u32x4 bfr[PT + 2];
#pragma unroll
        for (int r = 0; r < PT + 2; ++r) {
          const int qq = (wave * PT + r) * HPW + (j + dx);
          bfr[r] = *reinterpret_cast<const u32x4*>(xb + xq_addr(qq, g));
        }
        const unsigned char* wb = wbuf + (s & 1) * STAGE_BYTES + lane * 16;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          u32x4 a[NCT];
#pragma unroll
          for (int c = 0; c < NCT; ++c) a[c] = *reinterpret_cast<const u32x4*>(wb + (dy * NCT + c) * 1024);
#pragma unroll
          for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int t = 0; t < PT; ++t)
              acc[c][t] = mfma16(as_vec8<T>(a[c]), as_vec8<T>(bfr[t + dy]), acc[c][t]);
        }
      }
    }
    for (int ts = 0; ts < nts; ++ts) {                  // dense tail stages (three passes each) over chunk nq - 1
      const unsigned char* xb = xbuf + ((nq - 1) & 1) * XBUF_BYTES;
      const int s = 3 * nqf + ts;
      SB_BARRIER();
      if (s + 1 < nst) issue_w(s + 1);
      const unsigned char* wb = wbuf + (s & 1) * STAGE_BYTES + lane * 16;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const int tp = ts * 3 + pl;
        if (tp < ntp) {                                 // (wave-uniform)
          const int tap = tp * tgp + t_tl;
          const bool tv = tap < 9;
          const int tc = tv ? tap : 0;
          const int dy = tc / 3, dx = tc - dy * 3;
          u32x4 bt[PT];
#pragma unroll
          for (int t = 0; t < PT; ++t) {
            const int qq = (wave * PT + t + dy) * HPW + (j + dx);
            bt[t] = *reinterpret_cast<const u32x4*>(xb + xq_addr(qq, t_cg));
            if (!tv) bt[t] = (u32x4){0u, 0u, 0u, 0u};
          }
          u32x4 a[NCT];
#pragma unroll
          for (int c = 0; c < NCT; ++c) a[c] = *reinterpret_cast<const u32x4*>(wb + (pl * NCT + c) * 1024);
#pragma unroll
          for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int t = 0; t < PT; ++t) acc[c][t] = mfma16(as_vec8<T>(a[c]), as_vec8<T>(bt[t]), acc[c][t]);
        }
      }
    }

    // ---- epilogue 1: conditioning halo -> LDS (8 bytes = channels 0..3 per pixel, zeros outside the image)
    SB_BARRIER();
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pix = threadIdx.x + u * WAVES * 64;
      if (pix < HP) segh[pix] = sv[u];
    }
    __syncthreads();
    // ---- epilogue 2: hidden tile re-computed (two folded MFMAs per 16 x 16 block), dh masked with h > 0 and staged 16-bit
    // as [pixel][hidden]; the conditioning neighbourhood of every pixel as [pixel][(tap, c4)] beside it
    const int tA = 2 * g, tB = 2 * g + 1;
    const int offA = (tA / 3) * HPW + tA % 3, offB = (tB / 3) * HPW + tB % 3;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const u32x4 a0 = *reinterpret_cast<const u32x4*>(wsh + (c * 64 + lane) * 32);
      const u32x4 a1 = *reinterpret_cast<const u32x4*>(wsh + (c * 64 + lane) * 32 + 16);
      const f32x4 bq = *reinterpret_cast<const f32x4*>(wsh + NCT * 64 * 32 + (c * 64 + lane) * 16);
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        const int row = wave * PT + t;
        const int base = row * HPW + j;
        const u32x2 lo = segh[base + offA], hi = segh[base + offB];
        const u32x2 last = segh[base + 2 * HPW + 2];
        const u32x4 b0 = {lo[0], lo[1], hi[0], hi[1]};
        const u32x4 b1 = g == 0 ? (u32x4){last[0], last[1], 0u, 0u} : (u32x4){0u, 0u, 0u, 0u};
        const int px = row * 16 + j;
        const bool pin = ty0 + row < p.h && tx0 + j < p.w;         // pixels past the image contribute nothing
        f32x4 hacc = (f32x4){0.f, 0.f, 0.f, 0.f};
        hacc = mfma16(as_vec8<T>(a0), as_vec8<T>(b0), hacc);
        hacc = mfma16(as_vec8<T>(a1), as_vec8<T>(b1), hacc);
        // the mask is taken from h AS THE 16-BIT MAP WOULD HOLD IT (a positive h below the type's smallest number rounds to
        // zero there): bit for bit the mask of the unfused path, which reads the stored map
        float hr[4];
        unpack2<T>(pack2<T>(hacc[0] + bq[0], hacc[1] + bq[1]), hr[0], hr[1]);
        unpack2<T>(pack2<T>(hacc[2] + bq[2], hacc[3] + bq[3]), hr[2], hr[3]);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (pin && hr[r] > 0.f) ? acc[c][t][r] : 0.f;
        const u32x2 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
        const int slot = c * 2 + (g >> 1);
        *reinterpret_cast<u32x2*>(dhs + px * 128 + ((slot ^ swz(px & 7)) << 4) + (g & 1) * 8) = o;
      }
    }
    {
      const int px = threadIdx.x;                                  // 256 threads = 256 pixels
      const int row = px >> 4, col = px & 15;
      const int base = row * HPW + col;
      const int sw = swz(px & 7);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (s < 5) {
          const int ta = 2 * s, tb = 2 * s + 1;
          const u32x2 lo = segh[base + (ta / 3) * HPW + ta % 3];
          v[0] = lo[0]; v[1] = lo[1];
          if (tb < 9) {
            const u32x2 hi = segh[base + (tb / 3) * HPW + tb % 3];
            v[2] = hi[0]; v[3] = hi[1];
          }
        }
        *reinterpret_cast<u32x4*>(cis + px * 128 + ((s ^ sw) << 4)) = v;
      }
    }
    __syncthreads();
    // ---- epilogue 3: mlp_shared's gradient over this tile's 256 pixels: D[hidden 16 w ..][(tap, c4)] += dh^T x neighbourhood
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const u32x4 fa = tr_frag_at(dhs, ks * 32, wave, lane);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const u32x4 fb = tr_frag_at(cis, ks * 32, b, lane);
        accs[b] = mfma16(as_vec8<T>(fa), as_vec8<T>(fb), accs[b]);
      }
      accb = mfma16(as_vec8<T>(fa), as_vec8<T>(ones_frag<T>()), accb);
    }
  }

  // ---- this workgroup's partial block: rows = hidden 16 wave + 4 g + r, columns = 16 b + j (tap * 4 + c), bias at column 48
  float* out = p.part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 64 * PART_COLS;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float* rowp = out + (size_t)(wave * 16 + 4 * g + r) * PART_COLS;
#pragma unroll
    for (int b = 0; b < 3; ++b) rowp[b * 16 + j] = accs[b][r];
    if (j == 0) rowp[48] = accb[r];
  }
}

// dW[hidden][c][ky][kx] += sum over workgroups of part[wg][half][hidden % 64][tap * 4 + c]; db[hidden] += ... [48].
// One block per hidden row: thread (column = t & 63, lane = t >> 6) adds the workgroups lane, lane + 4, ... (in that order), the
// four lanes meet in LDS and are added in lane order: a fixed order whatever the launch looks like.  (The first version walked
// all workgroups from one thread per element: 66 us of dependent strided loads per layer.)
__global__ __launch_bounds__(256) void spade_hidden_bwd_reduce_kernel(const float* __restrict__ part, int wgs, int halves,
                                                                      float* __restrict__ dw, float* __restrict__ db, int hidden,
                                                                      int cond_c) {
  __shared__ float red[4][PART_COLS];
  const int hid = blockIdx.x;
  const int col = threadIdx.x & 63, ln = threadIdx.x >> 6;
  const int half = hid >> 6, row = hid & 63;
  float s0 = 0.f, s1 = 0.f;
  int wgi = ln;
  for (; wgi + 4 < wgs; wgi += 8) {        // two loads in flight, summed in workgroup order
    s0 += part[(((size_t)wgi * halves + half) * 64 + row) * PART_COLS + col];
    s1 += part[(((size_t)(wgi + 4) * halves + half) * 64 + row) * PART_COLS + col];
  }
  if (wgi < wgs) s0 += part[(((size_t)wgi * halves + half) * 64 + row) * PART_COLS + col];
  red[ln][col] = s0 + s1;
  __syncthreads();
  if (ln != 0) return;
  const float s = ((red[0][col] + red[1][col]) + red[2][col]) + red[3][col];
  if (col == 48) {
    if (db) db[hid] += s;
    return;
  }
  const int tap = col >> 2, c = col & 3;
  if (tap < 9 && c < cond_c) dw[((size_t)hid * cond_c + c) * 9 + tap] += s;
}

int hid_bwd_workgroups(const CganSpadeDesc* d) {
  const long tiles = (long)d->n * ((d->h + TH - 1) / TH) * ((d->w + TW - 1) / TW);
  return (int)(tiles < 256 ? tiles : 256);      // x 2 hidden halves = two workgroups per CU, each walking tiles / 256 tiles
}

}  // namespace

extern "C" size_t cgan_spade_hidden_bwd_workspace_bytes(const CganSpadeDesc* d) {
  if (!d || d->n <= 0 || d->h <= 0 || d->w <= 0) return 0;
  return (size_t)hid_bwd_workgroups(d) * 2 * 64 * PART_COLS * sizeof(float);
}

extern "C" int cgan_spade_hidden_bwd(const void* dgb, const void* packed_dgrad_gb, const void* cond_hw, const void* packed_w_shared,
                                     const float* bias_shared_padded, float* dw_shared, float* db_shared, void* workspace,
                                     size_t workspace_bytes, const CganSpadeDesc* d, void* stream) {
  CGAN_REQUIRE(d != nullptr && dgb && packed_dgrad_gb && cond_hw && packed_w_shared && bias_shared_padded && dw_shared && workspace,
               "spade_hidden_bwd: null pointer");
  CGAN_REQUIRE(d->dtype == CGAN_F16 || d->dtype == CGAN_BF16, "spade_hidden_bwd: bad dtype %d", d->dtype);
  CGAN_REQUIRE(d->hidden == 128 && d->ksize == 3, "spade_hidden_bwd: hidden width 128 and 3x3 kernels only (got %d, %d)", d->hidden,
               d->ksize);
  CGAN_REQUIRE(d->cond_c >= 1 && d->cond_c <= 4, "spade_hidden_bwd: conditioning images of <= 4 channels only (got %d)", d->cond_c);
  CGAN_REQUIRE(d->cond_h == d->h && d->cond_w == d->w, "spade_hidden_bwd: the conditioning image must be given at the map's extent");
  CGAN_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->c > 0, "spade_hidden_bwd: bad shape");
  CGAN_REQUIRE(workspace_bytes >= cgan_spade_hidden_bwd_workspace_bytes(d), "spade_hidden_bwd: workspace too small (%zu bytes)",
               workspace_bytes);
  SpadeHidBwdArgs a;
  a.dgb = (const uint16_t*)dgb; a.w_dg = (const u32x4*)packed_dgrad_gb; a.seg = (const uint16_t*)cond_hw;
  a.w_sh = (const u32x4*)packed_w_shared; a.b_sh = bias_shared_padded; a.part = (float*)workspace;
  a.n = d->n; a.h = d->h; a.w = d->w;
  a.gs = cgan_cs(2 * d->c);
  a.cin_p = (a.gs + 31) & ~31;
  a.ksteps_dg = 9 * (a.cin_p / 32);                   // 3x3 layout: one k-step per (tap, 32-channel chunk)
  a.ksteps_sh = 9;                                    // cond_c <= 4 -> cin_s 8 -> cin_p 32: one k-step per tap
  a.seg_cs = cgan_cs(d->cond_c);
  a.ctiles_dg = 8; a.ctiles_sh = 8;                   // 128 hidden channels
  {
    const int live = (a.gs - 32 * (a.cin_p / 32 - 1)) / 8;       // live 8-channel groups of the last 32-channel chunk
    a.tail_r = (live == 1 || live == 2) ? live : 0;
  }
  CGAN_REQUIRE((double)d->n * d->h * d->w * a.gs * 2.0 < 4294967295.0, "spade_hidden_bwd: map of 4 GiB or more");
  a.ntiles = d->n * ((d->h + TH - 1) / TH) * ((d->w + TW - 1) / TW);
  const int wgs = hid_bwd_workgroups(d);
  hipStream_t s = (hipStream_t)stream;
  static bool attr_set[2] = {false, false};
  const int ti = d->dtype == CGAN_F16 ? 0 : 1;
  if (!attr_set[ti]) {
    hipError_t e = ti == 0 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&spade_hidden_bwd_kernel<F16>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                           : hipFuncSetAttribute(reinterpret_cast<const void*>(&spade_hidden_bwd_kernel<BF16>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("spade_hidden_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set[ti] = true;
  }
  if (d->dtype == CGAN_F16)
    hipLaunchKernelGGL(spade_hidden_bwd_kernel<F16>, dim3(wgs, 2), dim3(WAVES * 64), SMEM_BYTES, s, a);
  else
    hipLaunchKernelGGL(spade_hidden_bwd_kernel<BF16>, dim3(wgs, 2), dim3(WAVES * 64), SMEM_BYTES, s, a);
  CGAN_CHECK_LAUNCH("spade_hidden_bwd");
  hipLaunchKernelGGL(spade_hidden_bwd_reduce_kernel, dim3(d->hidden), dim3(256), 0, s, (const float*)workspace, wgs, 2,
                     dw_shared, db_shared, d->hidden, d->cond_c);
  CGAN_CHECK_LAUNCH("spade_hidden_bwd(reduce)");
  return CGAN_OK;
}
