// LDS-tiled 3x3 / stride 1 / pad 1 convolution (NHWC 16-bit, MFMA 16x16x32, fp32 accumulate) for gfx950.
// Used for the SPADE-ResBlk main convs (reference climategan/blocks.py:350-351,372-375) at 32x32 and above, where
// the general gather kernel (conv_mfma.hip) re-reads every input pixel 9 times through L1/L2.
//
// One workgroup (4 waves, two workgroups per CU) = a 16 x 16 output tile x NCT <= 5 output-channel tiles:
//   - input channels are consumed in chunks of 32 (one MFMA k-step per tap): the 18 x 18 x 32 halo of the chunk is
//     brought into LDS by LDS-DMA (global_load_lds_dwordx4; zero padding comes from a zero page, the folded x2
//     nearest upsample is address math), double-buffered, XOR-swizzled on the SOURCE side so the MFMA B-fragment
//     reads are spread over the banks while the DMA writes LDS linearly;
//   - weights stream L2 -> LDS by LDS-DMA, one (chunk, dx) stage = 3 taps x NCT fragments ahead;
//   - the B fragments of the 6 halo rows a wave needs are read once per (chunk, dx) and reused for the 3 dy taps
//     ((3*NCT A + 6 B) fragment reads per 12*NCT MFMAs);
//   - epilogue: bias + residual (optionally through the folded upsample) + activation, staged through LDS so
//     the tile is written as whole 16-byte channel chunks (contiguous rows).
// HBM traffic: input once (+ halo) + output once (+ residual): the kernel is HBM-bound at 640x640 / Cout <= 40.
#include "conv3x3_lds.h"

namespace {

constexpr int TW = 16, TH = 16, WAVES = 4, PT = TH / WAVES;
constexpr int HPW = TW + 2, HPH = TH + 2, HP = HPH * HPW;   // 18 x 18 halo
constexpr int XDMA = (HP * 4 + 63) / 64;                    // 21 wave-wide DMAs per input chunk
constexpr int XBUF_BYTES = XDMA * 1024;                     // 21504 (324 px x 64 B + tail of the last DMA)
constexpr int MAX_NCT = 5;
CGAN_KNOB(int, g_lds_nct, 0);       // development knob (cgan_debug_set_conv3x3_nct): force the channel tiles per workgroup
CGAN_KNOB(int, g_c4_enabled, 1);    // development knob (cgan_debug_set_conv3x3_c4): 0 = never the folded-tap kernel

// [halo pixel q][4 slots of 16 B]: logical slot s of pixel q lives at slot position s ^ ((q >> 1) & 3).  A B-fragment
// ds_read_b128 -- lane (j, g) reads slot g of pixel base + j -- is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11,
// 16-19, 28-31} (+ 32): with bits 1..2 of q every group covers the 64 banks once, for every base (round 6; rounds 1-5 used
// bits 2..3, conflict-free only for 16 CONSECUTIVE lanes: 8 LDS cycles per read instead of 4 for 14 of 16 bases, the
// family's SQ_LDS_BANK_CONFLICT at 0.66 of its LDS-active cycles)
__device__ __forceinline__ int xq_addr(int q, int slot) { return q * 64 + ((slot ^ ((q >> 1) & 3)) << 4); }

// 16-byte-aligned zeros in device memory: the source of every out-of-image / pad-channel DMA lane
__device__ __attribute__((aligned(16))) unsigned int g_zeros[64];

// halo coordinate -> input coordinate: the conv's padding shifts the halo's origin (pad 1: the 'same' conv; 2: the 'full' conv
// = the data gradient of a pad-0 layer; 0: 'valid'), reflect padding (nn.ReflectionPad2d(1) folded into the mask / depth
// decoders' convs: reference blocks.py:21-78) mirrors the one ring it can reach.  Out-of-image positions stay out of range
// (zero page); positions of a tile's overhang past the image only feed outputs that are never stored.
__device__ __forceinline__ void halo_coord(const Conv3x3LdsArgs& p, int& yy, int& xx) {
  if (p.reflect) {
    yy = yy < 0 ? -yy : yy;
    xx = xx < 0 ? -xx : xx;
    yy = yy >= p.hi ? 2 * p.hi - 2 - yy : yy;
    xx = xx >= p.wi ? 2 * p.wi - 2 - xx : xx;
  }
}

#define STAGE_BARRIER()                                            \
  do {                                                             \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                                  \
    asm volatile("" ::: "memory");                                 \
  } while (0)

// Epilogue shared by both kernels: bias + residual (optionally through the folded upsample) + activation, staged
// through LDS so that the tile leaves as whole 16-byte channel chunks.
// RES: whether the kernel takes a residual at all (the first-layer kernels never do: no registers for its prefetch there)
template <typename T, int NCT, bool RES = true>
__device__ __forceinline__ void conv3x3_epilogue(const Conv3x3LdsArgs& p, f32x4 (&acc)[NCT][PT], unsigned char* smem,
                                                 int n, int ty0, int tx0, int ct0, int wave, int j, int g) {
  // ---------------- epilogue: lane holds channels (ct0+c)*16 + 4g + {0..3} of pixel (row wave*PT+t, column j)
  if (p.shuffle) {
    // sub-pixel data gradient of a 4x4 / stride-2 / pad-1 convolution with <= 4 input channels (the discriminators' first
    // layer): channel 4 g + r of this 3x3 convolution over dy = gradient channel r of pixel (2 y + (g >> 1), 2 x + (g & 1));
    // one 16-byte store per lane (4 values + the 4 zero pad channels of the 8-channel storage), the lanes of g and g ^ 1
    // fill 32 contiguous bytes, a 16-lane row 512
    const int a = g >> 1, b = g & 1;
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int yy = ty0 + wave * PT + t, xx = tx0 + j;
      const int oy = 2 * yy + a, ox = 2 * xx + b;
      if (yy < p.h && xx < p.w_ && oy < p.shuffle && ox < p.shuffle_w && ct0 == 0) {
        const f32x4 v = acc[0][t];
        const u32x4 o = {pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), 0u, 0u};
        *reinterpret_cast<u32x4*>(p.y + (((size_t)n * p.shuffle + oy) * p.shuffle_w + ox) * 8) = o;
      }
    }
    return;
  }
  __syncthreads();                                   // everyone is done with xbuf / wbuf
  unsigned char* yt = smem;                          // [256 px][NCT * 32 B]
  // the bias quad of a cout tile is loaded once (not per pixel row); the pad-channel test only runs when there are any
  const bool pad_c = p.cout < p.cout_s;
  f32x4 bias_q[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int ch = (ct0 + c) * 16 + g * 4;           // (the bias vector is padded to whole cout tiles)
    bias_q[c] = (p.bias && ch < p.cout_s) ? *reinterpret_cast<const f32x4*>(p.bias + ch) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // the residual quads of the whole tile are requested together (round 6): a load next to its use compiled to
  // global_load; s_waitcnt vmcnt(0) per (row, cout tile) -- PT * NCT serialized round trips per wave
  u32x2 rq[RES ? PT : 1][RES ? NCT : 1];
  if (RES && p.has_res) {                            // wave-uniform
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int yy = ty0 + wave * PT + t, xx = tx0 + j;
      const bool pin = yy < p.h && xx < p.w_;
      size_t rbase = 0;
      if (pin) {
        if (p.res_ups) rbase = (((size_t)n * (p.h >> 1) + (yy >> 1)) * (p.w_ >> 1) + (xx >> 1)) * p.cout_s;
        else rbase = (((size_t)n * p.h + yy) * p.w_ + xx) * p.cout_s;
      }
#pragma unroll
      for (int c = 0; c < NCT; ++c) {
        const int ch = (ct0 + c) * 16 + g * 4;
        rq[t][c] = (u32x2){0u, 0u};
        if (pin && ch < p.cout_s) rq[t][c] = *reinterpret_cast<const u32x2*>(p.res + rbase + ch);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);              // vmcnt(0), once, as a builtin the wait-count pass sees (conv_gemm.h)
  }
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int row = wave * PT + t;
    const int yy = ty0 + row, xx = tx0 + j;
    const bool pin = yy < p.h && xx < p.w_;
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int ch = (ct0 + c) * 16 + g * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[c][t][r];
      if (ch < p.cout_s) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += bias_q[c][r];
        if (RES && p.has_res && pin) {
          const u32x2 rv = rq[RES ? t : 0][RES ? c : 0];
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
            if (ch + r >= p.cout) v[r] = 0.f;        // keep pad channels zero
        }
      }
      u32x2 o;
      o[0] = pack2<T>(v[0], v[1]);
      o[1] = pack2<T>(v[2], v[3]);
      *reinterpret_cast<u32x2*>(yt + ((row * 16 + j) * NCT + c) * 32 + g * 8) = o;
    }
  }
  __syncthreads();
  // whole 16-byte chunks, lane-linear over [256 px][2*NCT chunks]
#pragma unroll
  for (int k = 0; k < 2 * NCT; ++k) {
    const int id = k * (WAVES * 64) + threadIdx.x;
    const int pix = id / (2 * NCT), cc = id - pix * (2 * NCT);
    const int yy = ty0 + (pix >> 4), xx = tx0 + (pix & 15);
    const int ch = ct0 * 16 + cc * 8;
    if (yy < p.h && xx < p.w_ && ch < p.cout_s)
      CGAN_ST_STREAM(*reinterpret_cast<const u32x4*>(yt + id * 16),
                     reinterpret_cast<u32x4*>(p.y + (((size_t)n * p.h + yy) * p.w_ + xx) * p.cout_s + ch));
  }
}

template <typename T, int NCT>
__global__ __launch_bounds__(WAVES * 64, 2) void conv3x3_lds_kernel(Conv3x3LdsArgs p) {
  const u32x4* zero_page = reinterpret_cast<const u32x4*>(g_zeros);
  constexpr int STAGE_BYTES = 3 * NCT * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xbuf = smem;                       // 2 * XBUF_BYTES
  unsigned char* wbuf = xbuf + 2 * XBUF_BYTES;      // 2 * STAGE_BYTES

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int g = lane >> 4;

  const int tiles_x = (p.w_ + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
  int tile = blockIdx.x;
  const int txi = tile % tiles_x;
  tile /= tiles_x;
  const int tyi = tile % tiles_y;
  const int n = tile / tiles_y;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int ct0 = blockIdx.y * NCT;                 // first output-channel tile of this workgroup
  const int nq = p.cin_p / 32;                      // input-channel chunks
  const int nstages = nq * 3;

  // ---- LDS-DMA of input chunk q (18 x 18 halo x 32 channels) into xbuf[q & 1]
  auto issue_x = [&](int q) {
    unsigned char* dst = xbuf + (q & 1) * XBUF_BYTES;
    for (int i = wave; i < XDMA; i += WAVES) {
      const int idx = i * 64 + lane;                // LDS position: pixel idx/4, slot position idx%4
      const int pix = idx >> 2, spos = idx & 3;
      const int slot = spos ^ ((pix >> 1) & 3);     // logical 8-channel group that belongs at this position
      const int py = pix / HPW, px = pix - py * HPW;
      int yy = ty0 - p.pad + py, xx = tx0 - p.pad + px;
      halo_coord(p, yy, xx);
      const int ch = q * 32 + slot * 8;
      const u32x4* src = zero_page;
      if (pix < HP && yy >= 0 && yy < p.hi && xx >= 0 && xx < p.wi && ch < p.cin_s) {
        const int sy = p.in_ups ? (yy >> 1) : yy, sx = p.in_ups ? (xx >> 1) : xx;
        src = reinterpret_cast<const u32x4*>(p.x + (((size_t)n * p.hx + sy) * p.wx + sx) * p.cin_s + ch);
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    }
  };
  // ---- LDS-DMA of weight stage s = (chunk q, dx): 3 dy taps x NCT channel tiles -> wbuf[s & 1], slot dy*NCT + c
  auto issue_w = [&](int s) {
    const int q = s / 3, dx = s - q * 3;
    unsigned char* dst = wbuf + (s & 1) * STAGE_BYTES;
#pragma unroll
    for (int i0 = 0; i0 < 3 * NCT; i0 += WAVES) {
      const int i = i0 + wave;
      if (i < 3 * NCT) {
        const int dy = i / NCT, c = i - dy * NCT;
        const int ct = min(ct0 + c, p.ctiles - 1);
        const u32x4* src = p.w + ((size_t)ct * p.ksteps + (dy * 3 + dx) * nq + q) * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      }
    }
  };

  issue_x(0);
  issue_w(0);

  f32x4 acc[NCT][PT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---------------- K loop
  for (int q = 0; q < nq; ++q) {
    const unsigned char* xb = xbuf + (q & 1) * XBUF_BYTES;
    for (int dx = 0; dx < 3; ++dx) {
      const int s = q * 3 + dx;
      STAGE_BARRIER();                               // chunk q and weight stage s have landed; other buffers free
      if (dx == 0 && q + 1 < nq) issue_x(q + 1);
      if (s + 1 < nstages) issue_w(s + 1);
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

  conv3x3_epilogue<T, NCT>(p, acc, smem, n, ty0, tx0, ct0, wave, j, g);
}

// Cin <= 32 (one input chunk): a single barrier.  The whole 9-tap weight set (9*NCT fragments) and the input halo
// are DMA'd up front, no double buffering (39.5 KB of LDS at NCT = 2, <= 128 VGPRs: four workgroups per CU, which
// is what hides the DMA latency of these short, HBM-bound workgroups: the 640x640 20/40-channel Painter convs).
template <typename T, int NCT>
__global__ __launch_bounds__(WAVES * 64, 4) void conv3x3_lds_onechunk_kernel(Conv3x3LdsArgs p) {
  const u32x4* zero_page = reinterpret_cast<const u32x4*>(g_zeros);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* xbuf = smem;                       // XBUF_BYTES
  unsigned char* wbuf = xbuf + XBUF_BYTES;          // 9 * NCT KiB, slot tap*NCT + c

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int g = lane >> 4;
  const int tiles_x = (p.w_ + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
  int tile = blockIdx.x;
  const int txi = tile % tiles_x;
  tile /= tiles_x;
  const int tyi = tile % tiles_y;
  const int n = tile / tiles_y;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int ct0 = blockIdx.y * NCT;

  for (int i = wave; i < XDMA; i += WAVES) {
    const int idx = i * 64 + lane;
    const int pix = idx >> 2, spos = idx & 3;
    const int slot = spos ^ ((pix >> 1) & 3);
    const int py = pix / HPW, px = pix - py * HPW;
    int yy = ty0 - p.pad + py, xx = tx0 - p.pad + px;
    halo_coord(p, yy, xx);
    const int ch = slot * 8;
    const u32x4* src = zero_page;
    if (pix < HP && yy >= 0 && yy < p.hi && xx >= 0 && xx < p.wi && ch < p.cin_s) {
      const int sy = p.in_ups ? (yy >> 1) : yy, sx = p.in_ups ? (xx >> 1) : xx;
      src = reinterpret_cast<const u32x4*>(p.x + (((size_t)n * p.hx + sy) * p.wx + sx) * p.cin_s + ch);
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(xbuf + i * 1024), 16, 0, 0);
  }
#pragma unroll
  for (int i0 = 0; i0 < 9 * NCT; i0 += WAVES) {
    const int i = i0 + wave;
    if (i < 9 * NCT) {
      const int tap = i / NCT, c = i - tap * NCT;
      const int ct = min(ct0 + c, p.ctiles - 1);
      const u32x4* src = p.w + ((size_t)ct * p.ksteps + tap) * 64 + lane;   // nq == 1: ks = tap
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(wbuf + i * 1024), 16, 0, 0);
    }
  }

  f32x4 acc[NCT][PT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  STAGE_BARRIER();
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    u32x4 bfr[PT + 2];
#pragma unroll
    for (int r = 0; r < PT + 2; ++r) {
      const int qq = (wave * PT + r) * HPW + (j + dx);
      bfr[r] = *reinterpret_cast<const u32x4*>(xbuf + xq_addr(qq, g));
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      u32x4 a[NCT];
#pragma unroll
      for (int c = 0; c < NCT; ++c)
        a[c] = *reinterpret_cast<const u32x4*>(wbuf + ((dy * 3 + dx) * NCT + c) * 1024 + lane * 16);
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int t = 0; t < PT; ++t)
          acc[c][t] = mfma16(as_vec8<T>(a[c]), as_vec8<T>(bfr[t + dy]), acc[c][t]);
    }
  }
  conv3x3_epilogue<T, NCT>(p, acc, smem, n, ty0, tx0, ct0, wave, j, g);
}

// <= 4 input channels (mlp_shared of every SPADE layer re-materialised in the backward, conv 3 -> 128 on the conditioning
// image: reference norms.py:158-162; round 3).  The kernels above spend one MFMA k-step (32 K values) per TAP, i.e. 9 MFMAs
// per tile pair for 27 real K values: at 4 x 640^2 -> 128 channels 343 us, four times the 84 us the 420 MB of output take
// at the HBM write rate.  Here the taps are FOLDED into K: k-step 0 = taps 0..7 x 4 channels, k-step 1 = tap 8 (+ 28
// zeros): two MFMAs instead of nine.
//   * the 18 x 18 x 4-channel halo (2.6 KiB) is staged in LDS with plain 8-byte loads (channels 0..3 of each stored pixel);
//   * B fragment of a pixel tile (one output row): lane (j, g) needs taps 2g and 2g + 1 of pixel j = two 8-byte LDS reads;
//   * A fragments come straight from the standard packed weights (one k-step per tap, channel = K index): lane (j, g) of
//     the folded fragment is the first 8 bytes of lane (j, 0) of the k-steps of taps 2g and 2g + 1 -- 3 small loads per cout
//     tile, once per workgroup;
//   * all couts (<= 8 tiles) in one workgroup, epilogue shared with the kernels above.
template <typename T, int NCT>
__global__ __launch_bounds__(WAVES * 64, NCT <= 2 ? 8 : (NCT <= 4 ? 4 : 2)) void conv3x3_c4_kernel(Conv3x3LdsArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int g = lane >> 4;
  const int tiles_x = (p.w_ + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
  int tile = blockIdx.x;
  const int txi = tile % tiles_x;
  tile /= tiles_x;
  const int tyi = tile % tiles_y;
  const int n = tile / tiles_y;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int ct0 = blockIdx.y * NCT;

  // ---- halo -> LDS: 8 bytes (channels 0..3) per pixel, zeros outside the image
  u32x2* xh = reinterpret_cast<u32x2*>(smem);        // [18 * 18]
  for (int pix = threadIdx.x; pix < HP; pix += WAVES * 64) {
    const int py = pix / HPW, px = pix - py * HPW;
    int yy = ty0 - p.pad + py, xx = tx0 - p.pad + px;
    halo_coord(p, yy, xx);
    u32x2 v = {0u, 0u};
    if (yy >= 0 && yy < p.hi && xx >= 0 && xx < p.wi) {
      const int sy = p.in_ups ? (yy >> 1) : yy, sx = p.in_ups ? (xx >> 1) : xx;
      v = *reinterpret_cast<const u32x2*>(p.x + (((size_t)n * p.hx + sy) * p.wx + sx) * p.cin_s);
    }
    xh[pix] = v;
  }
  // ---- folded A fragments
  u32x4 a0[NCT], a1[NCT];
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int ct = min(ct0 + c, p.ctiles - 1);
    const u32x2* wt = reinterpret_cast<const u32x2*>(p.w + (size_t)ct * p.ksteps * 64);   // lane l of k-step ks: wt[(ks * 64 + l) * 2]
    const u32x2 lo = wt[((2 * g) * 64 + j) * 2], hi = wt[((2 * g + 1) * 64 + j) * 2];
    const u32x2 last = wt[(8 * 64 + j) * 2];
    a0[c] = (u32x4){lo[0], lo[1], hi[0], hi[1]};
    a1[c] = g == 0 ? (u32x4){last[0], last[1], 0u, 0u} : (u32x4){0u, 0u, 0u, 0u};
  }
  f32x4 acc[NCT][PT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int tA = 2 * g, tB = 2 * g + 1;               // this lane's taps of k-step 0
  const int offA = (tA / 3) * HPW + tA % 3, offB = (tB / 3) * HPW + tB % 3;
#pragma unroll
  for (int t = 0; t < PT; ++t) {
    const int base = (wave * PT + t) * HPW + j;       // halo pixel of tap (0, 0) for output (row, j)
    const u32x2 lo = xh[base + offA], hi = xh[base + offB];
    const u32x2 last = xh[base + 2 * HPW + 2];
    const u32x4 b0 = {lo[0], lo[1], hi[0], hi[1]};
    const u32x4 b1 = g == 0 ? (u32x4){last[0], last[1], 0u, 0u} : (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      acc[c][t] = mfma16(as_vec8<T>(a0[c]), as_vec8<T>(b0), acc[c][t]);
      acc[c][t] = mfma16(as_vec8<T>(a1[c]), as_vec8<T>(b1), acc[c][t]);
    }
  }
  conv3x3_epilogue<T, NCT, false>(p, acc, smem, n, ty0, tx0, ct0, wave, j, g);       // (launch_nct: never with a residual)
}

// First-layer convolutions (round 5): 8 storage channels per input pixel (<= 8 real: the image, the mask || image pair of the
// PatchGAN), k x k taps with stride 1 / 2, <= 64 output channels -- discriminator.py:100-120 (4x4 s2), resnet101_v3.py:176
// (7x7 s2 stem).  The general gather kernel read every tap of every pixel as its own 16-byte global load and stored 8 bytes per
// lane: 95 / 142 us at 8 x 640 x 640 for the 20 us their 105 MB of output take.  Here the input halo of a 16 x 16 output tile
// ((15 s + k)^2 pixels x 16 B <= 22 KiB) is staged once in LDS; a k-step = four taps (the standard dense pack: k = tap * 8 + c),
// lane (j, g) reads tap 4 ks + g of its pixel as ONE ds_read_b128; A fragments straight from L2 (16 B per lane, prefetched
// one k-step ahead); all output channels in one workgroup; epilogue shared with the 3x3 kernels (whole 16-byte chunks).
template <typename T, int NCT>
__global__ __launch_bounds__(WAVES * 64, 4) void conv_smallcin_kernel(Conv3x3LdsArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int g = lane >> 4;
  const int tiles_x = (p.w_ + TW - 1) / TW, tiles_y = (p.h + TH - 1) / TH;
  int tile = blockIdx.x;
  const int txi = tile % tiles_x;
  tile /= tiles_x;
  const int tyi = tile % tiles_y;
  const int n = tile / tiles_y;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int hp = (TW - 1) * p.stride + p.k;          // halo extent (pixels) per axis
  const int taps = p.k * p.k;

  // ---- input halo -> LDS, 16 bytes (8 storage channels) per pixel, zeros outside the image
  u32x4* xh = reinterpret_cast<u32x4*>(smem);
  const int iy0 = ty0 * p.stride - p.pad, ix0 = tx0 * p.stride - p.pad;
  for (int pix = threadIdx.x; pix < hp * hp; pix += WAVES * 64) {
    const int py = pix / hp, px = pix - py * hp;
    const int yy = iy0 + py, xx = ix0 + px;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (yy >= 0 && yy < p.hi && xx >= 0 && xx < p.wi)
      v = *reinterpret_cast<const u32x4*>(p.x + (((size_t)n * p.hi + yy) * p.wi + xx) * 8);
    xh[pix] = v;
  }
  f32x4 acc[NCT][PT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto load_a = [&](int ks, u32x4 (&a)[NCT]) {
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
      const int ct = min(c, p.ctiles - 1);
      a[c] = p.w[((size_t)ct * p.ksteps + ks) * 64 + lane];
    }
  };
  u32x4 a_cur[NCT], a_nxt[NCT];
  load_a(0, a_cur);
  __syncthreads();
  for (int ks = 0; ks < p.ksteps; ++ks) {
    if (ks + 1 < p.ksteps) load_a(ks + 1, a_nxt);
    const int tap = ks * 4 + g;
    const bool tv = tap < taps;
    const int tc = tv ? tap : 0;
    const int ky = tc / p.k, kx = tc - ky * p.k;
    u32x4 b[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const int row = wave * PT + t;
      b[t] = xh[(row * p.stride + ky) * hp + j * p.stride + kx];
      if (!tv) b[t] = (u32x4){0u, 0u, 0u, 0u};     // a lane without a tap multiplies zeros (its packed weights are zero too)
    }
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int t = 0; t < PT; ++t) acc[c][t] = mfma16(as_vec8<T>(a_cur[c]), as_vec8<T>(b[t]), acc[c][t]);
#pragma unroll
    for (int c = 0; c < NCT; ++c) a_cur[c] = a_nxt[c];
  }
  conv3x3_epilogue<T, NCT, false>(p, acc, smem, n, ty0, tx0, 0, wave, j, g);          // (has_res = 0 by construction)
}

template <typename T, int NCT>
int launch_smallcin(const Conv3x3LdsArgs& a, hipStream_t s) {
  const int tiles = a.n * ((a.h + TH - 1) / TH) * ((a.w_ + TW - 1) / TW);
  const int hp = (TW - 1) * a.stride + a.k;
  size_t smem = (size_t)hp * hp * 16;
  const size_t epi = (size_t)256 * NCT * 32;
  if (epi > smem) smem = epi;
  hipLaunchKernelGGL((conv_smallcin_kernel<T, NCT>), dim3(tiles), dim3(WAVES * 64), smem, s, a);
  return CGAN_OK;
}

template <typename T>
int launch_smallcin_any(const Conv3x3LdsArgs& a, hipStream_t s) {
  switch (a.ctiles) {
    case 1: return launch_smallcin<T, 1>(a, s);
    case 2: return launch_smallcin<T, 2>(a, s);
    case 3: return launch_smallcin<T, 3>(a, s);
    default: return launch_smallcin<T, 4>(a, s);
  }
}

template <typename T, int NCT>
int launch_c4(const Conv3x3LdsArgs& a, hipStream_t s) {
  const int tiles = a.n * ((a.h + TH - 1) / TH) * ((a.w_ + TW - 1) / TW);
  const size_t smem = (size_t)256 * NCT * 32;          // the epilogue's staging area (>= the 2.6 KiB halo)
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c4_kernel<T, NCT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("conv3x3_c4: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_c4_kernel<T, NCT>), dim3(tiles, ceil_div(a.ctiles, NCT)), dim3(WAVES * 64), smem, s, a);
  return CGAN_OK;
}

template <typename T, int NCT>
int launch(const Conv3x3LdsArgs& a, hipStream_t s) {
  const int tiles = a.n * ((a.h + TH - 1) / TH) * ((a.w_ + TW - 1) / TW);
  const int chunks = ceil_div(a.ctiles, NCT);
  size_t smem = (size_t)2 * XBUF_BYTES + 2 * 3 * NCT * 1024;
  const size_t epi = (size_t)256 * NCT * 32;
  if (epi > smem) smem = epi;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_lds_kernel<T, NCT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("conv3x3_lds: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  if (NCT <= 2 && a.cin_p == 32) {
    size_t smem1 = (size_t)XBUF_BYTES + 9 * NCT * 1024;
    if (epi > smem1) smem1 = epi;
    static bool attr1_set = false;
    if (!attr1_set) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_lds_onechunk_kernel<T, (NCT <= 2 ? NCT : 1)>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) {
        cgan_set_error("conv3x3_lds: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return CGAN_ERR_HIP;
      }
      attr1_set = true;
    }
    hipLaunchKernelGGL((conv3x3_lds_onechunk_kernel<T, (NCT <= 2 ? NCT : 1)>), dim3(tiles, chunks), dim3(WAVES * 64), smem1,
                       s, a);
    return CGAN_OK;
  }
  hipLaunchKernelGGL((conv3x3_lds_kernel<T, NCT>), dim3(tiles, chunks), dim3(WAVES * 64), smem, s, a);
  return CGAN_OK;
}

template <typename T>
int launch_nct(const Conv3x3LdsArgs& a, hipStream_t s) {
  // folded taps for <= 4 input channels stored as 4 / 8 / ... channels per pixel (the packed weights keep one k-step per tap)
  if (g_c4_enabled && a.cin <= 4 && a.cin_p == 32 && a.ksteps == 9 && (a.cin_s & 3) == 0 && !a.has_res) {
    // Channel tiles per workgroup.  The workgroup is a latency chain (halo load -> 2 MFMAs per tile -> staged stores); its
    // epilogue staging (8 KiB per channel tile) decides how many share a CU.  4 x 640^2, 128 couts, same box
    // (tools/bench_conv.py --c4): 8 tiles (2 per CU) 195 us, 4 tiles 135 us, 2 tiles 135 us; at 320^2 56 / 47 / 39 us --
    // so 4 tiles (whole 128-byte lines per pixel) on large grids, 2 where the grid would otherwise leave CUs waiting.
    // (g_c4_enabled 2 / 3: force 8 / 2 tiles -- development knob)
    const long wg4 = (long)a.n * ((a.h + TH - 1) / TH) * ((a.w_ + TW - 1) / TW) * ceil_div(a.ctiles, 4);
    if (a.ctiles <= 2 || g_c4_enabled == 3 || (g_c4_enabled == 1 && wg4 < 8192)) return launch_c4<T, 2>(a, s);
    if (a.ctiles <= 4 || g_c4_enabled != 2) return launch_c4<T, 4>(a, s);
    return launch_c4<T, 8>(a, s);
  }
  // channel tiles per workgroup: the divisor of ctiles (<= 5) with the least padding
  if (g_lds_nct >= 1 && g_lds_nct <= MAX_NCT) {          // development knob
    switch (g_lds_nct) {
      case 1: return launch<T, 1>(a, s);
      case 2: return launch<T, 2>(a, s);
      case 3: return launch<T, 3>(a, s);
      case 4: return launch<T, 4>(a, s);
      default: return launch<T, 5>(a, s);
    }
  }
  int nct = a.ctiles < MAX_NCT ? a.ctiles : MAX_NCT;
  if (a.ctiles > MAX_NCT) {
    int best = MAX_NCT, waste = ceil_div(a.ctiles, MAX_NCT) * MAX_NCT - a.ctiles;
    for (int k = MAX_NCT - 1; k >= 3; --k) {
      int w = ceil_div(a.ctiles, k) * k - a.ctiles;
      if (w < waste) { waste = w; best = k; }
    }
    nct = best;
  }
  switch (nct) {
    case 1: return launch<T, 1>(a, s);
    case 2: return launch<T, 2>(a, s);
    case 3: return launch<T, 3>(a, s);
    case 4: return launch<T, 4>(a, s);
    default: return launch<T, 5>(a, s);
  }
}

}  // namespace

CGAN_DEV_ONLY(extern "C" void cgan_debug_set_conv3x3_c4(int v) { g_c4_enabled = v; })
CGAN_DEV_ONLY(extern "C" void cgan_debug_set_conv3x3_nct(int v) { g_lds_nct = v; })

bool conv3x3_lds_applicable(const CganConvDesc* d) {
  const bool pad_ok = d->pad_mode == CGAN_PAD_ZERO ? (d->pad >= 0 && d->pad <= 2)
                                                   : (d->pad == 1 && d->h_in >= 2 && d->w_in >= 2 && !d->in_upsample);
  return d->kh == 3 && d->kw == 3 && d->stride == 1 && d->dilation == 1 && pad_ok && (long)d->h_out * d->w_out >= 1024;
}

bool conv_smallcin_applicable(const CganConvDesc* d) {
  const int cs = cgan_cs(d->c_in);
  return cs == 8 && d->kh == d->kw && d->kh != 3 && d->kh >= 2 && d->kh <= 7 && (d->stride == 1 || d->stride == 2) &&
         d->dilation == 1 && d->pad_mode == CGAN_PAD_ZERO && d->pad >= 0 && d->pad <= d->kh && !d->in_upsample &&
         cgan_cs(d->c_out) <= 64 && !d->has_residual && (long)d->h_out * d->w_out >= 1024 &&
         (double)d->n * d->h_in * d->w_in * 16.0 < 4294967295.0;
}

int conv_smallcin_launch(const Conv3x3LdsArgs& a, int dtype, hipStream_t s) {
  return dtype == CGAN_F16 ? launch_smallcin_any<F16>(a, s) : launch_smallcin_any<BF16>(a, s);
}

int conv3x3_lds_launch(const Conv3x3LdsArgs& a, int dtype, hipStream_t s) {
  return dtype == CGAN_F16 ? launch_nct<F16>(a, s) : launch_nct<BF16>(a, s);
}
