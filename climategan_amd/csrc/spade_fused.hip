// Fused SPADE forward for gfx950 (reference climategan/norms.py:146-186):
//
//   seg   = nearest_resize(cond, (h, w))
//   actv  = ReLU(conv3x3(seg, w_shared) + b_shared)                    128 hidden channels
//   gamma = conv3x3(actv, w_gamma) + b_gamma ; beta = conv3x3(actv, w_beta) + b_beta
//   y     = act( (x - mean) * rstd * (1 + gamma) + beta )
//
// One workgroup (4 waves, one per SIMD; two workgroups co-resident per CU) owns a 16 x 16 pixel tile of one
// image and NCT <= 5 channel tiles (a channel tile = 16 MFMA rows = 8 gamma + 8 beta channels):
//
//   prologue  conditioning halo 20 x 20 x cond_c -> LDS, first weight stage in flight (LDS-DMA), hidden quarter 0
//   K loop    the 128 hidden channels are processed in four QUARTERS of 32 (one MFMA k-step per tap), so only
//             18*18*32 values (21 KB) of the hidden map are resident per quarter, double-buffered:
//               - gamma||beta implicit GEMM out of LDS, fp32 accumulate: M = NCT*16 rows, N = 256 pixels,
//                 12 stages = 4 quarters x 3 dx, the 3 dy taps inside a stage.  The activation (B) fragments of the
//                 6 halo rows a wave needs are read once per (quarter, dx) and reused for the three dy taps; weight
//                 (A) fragments stream L2 -> LDS by LDS-DMA, one stage (3*NCT KiB) ahead, double-buffered.
//               - between the MFMA blocks each wave computes its share of the NEXT quarter's hidden map (cond gather
//                 -> small MFMAs with the shared-conv weights held in registers, bias through a constant-one column
//                 -> ReLU -> LDS); for NCT <= 3 that work is interleaved with the MFMAs at instruction granularity
//                 (sched_group_barrier), for NCT >= 4 it runs between blocks (the interleaved form spills there).
//               - the x tile is DMA'd into the idle hidden buffer during the last quarter.
//   epilogue  one v_permlane32_swap pair brings gamma and beta of the same channel into the same lane; x is
//             normalised, activated and written back through LDS as whole 16-byte channel chunks.
// The 128-channel hidden map (105 MB/img at 640x640 in 16-bit) never touches HBM; HBM traffic is x (read) +
// y (write) + the 4-channel cond halo.  LDS fragment reads per MFMA: (3*NCT A + 6 B) / (12*NCT) = 0.35 at NCT 5.
// Conditioning with more than 3 channels (the SPADE mask decoder's 15) takes a generic, non-interleaved hidden path.
//
// MFMA operand roles: A = weights (rows = output channels), B = activations (cols = pixels), so that
// D's per-lane 4 registers are 4 consecutive channel rows of one pixel (col = lane&15, row = 4*(lane>>4)+r).
#include "cgan_common.h"

#include <type_traits>

#ifndef CGAN_SPADE_WPE
#define CGAN_SPADE_WPE 4
#endif
namespace {

constexpr int TW = 16;       // pixel-tile width == MFMA N
constexpr int TH = 16;
constexpr int HID = 128;     // hidden channels (norms.py:163)
constexpr int KS_GB = 36;    // 9 taps * 128 / 32
constexpr int MAX_NCT = 5;   // channel tiles per workgroup
constexpr int WAVES = 4;
constexpr int PT = TH / WAVES;                              // pixel-tile rows per wave
constexpr int HPH = TH + 2, HPW = TW + 2, HP = HPH * HPW;   // hidden halo (18 x 18)
constexpr int CTH = TH + 4, CTW = TW + 4;                   // cond halo (20 x 20)
constexpr int NHT = (HP + 15) / 16;                         // 21 hidden pixel tiles
constexpr int QC = 32;                                      // hidden channels per quarter
constexpr int ACTV_Q_BYTES = NHT * 16 * QC * 2;             // 21504: 21 full hidden tiles (18*18 = 324 pixels used)
constexpr int NSTAGES = 12;                                 // 4 quarters x 3 dx (3 dy taps per stage)
constexpr int NBUF = 2;                                     // weight-stage double buffer

struct SpadeParams {
  const uint16_t* x;
  const float* mean;
  const float* rstd;
  const uint16_t* cond;
  const u32x4* w_sh;    // C4 path: [8][64] fragments of the single k-step;  generic: [8][ksh][64]
  const u32x4* w_gb;    // [nt][36][64]
  const float* b_gb;    // [nt][16]  (gamma bias + 1 | beta bias), zero on pad channels
  uint16_t* y;
  uint16_t* gamma;      // training: the modulation map gamma (bias included, without the +1) [n][h][w][cs] for the backward, or null
  int n, h, w, c, cs, nt;
  int hx, wx, x_ups;
  int cond_h, cond_w, cond_c, cond_cs, ksh;
  float sy, sx;
  int tiles_y, tiles_x;
  int act;
  float slope;
  CGAN_DEV_ONLY(unsigned long long* tsbuf;)  // dev build: per-workgroup phase timestamps, or null
  CGAN_DEV_ONLY(int dbg;)                    // dev build: ablation bits: 1 skip hidden map, 2 skip main MFMAs, 8 skip stores
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// "C4" path: cond stored as 4 channels per pixel of which at most 3 are used (the Painter's RGB conditioning): the
// shared conv then is ONE k-step of 32 -- K = tap * 4 + c for taps 0..7, and the unused 4th-channel slots of taps 0..3
// carry tap 8's three channels and the constant one of the bias (half the hidden-map MFMAs of a two-k-step layout)
__host__ inline bool is_c4(int cond_c) { return cond_c <= 3; }
__host__ inline int ksh_of(int cond_c) { return is_c4(cond_c) ? 1 : ceil_div(9 * cond_c + 1, 32); }

struct PackedLayout {
  size_t w_sh, w_gb, b_gb, total;
};
__host__ inline PackedLayout packed_layout(int cs, int cond_c) {
  PackedLayout L;
  int nt = cs / 8;
  L.w_sh = 0;
  size_t wsh_bytes = (size_t)8 * ksh_of(cond_c) * 1024;
  L.w_gb = align16(L.w_sh + wsh_bytes);
  L.b_gb = align16(L.w_gb + (size_t)nt * KS_GB * 64 * 16);
  L.total = align16(L.b_gb + (size_t)nt * 16 * sizeof(float));
  return L;
}

// hidden map in LDS: [buffer][halo pixel q][4 slots of 16 B = 32 channels]; the slot is XOR-swizzled with bits
// 1..2 of q (round 6: conflict-free for ds_read_b128's lane groups at every base, see conv3x3_lds.hip's xq_addr; bits 2..3
// were a 2-way conflict on most reads)
// SWZ = false (the wave-specialised kernel): plain [pixel][slot] order.  A B-fragment address is then one per-lane
// constant plus compile-time / wave-uniform offsets -- the swizzled form needs a separate address register per (row, dx),
// which the compiler hoists out of the K loop: 18+ long-lived VGPRs that the 128-register consumers do not have.  Cost: a
// 2-way bank conflict on every B-fragment read (6 of the 21 reads of a stage; measured irrelevant next to 60 MFMAs).
template <bool SWZ>
__device__ __forceinline__ int actv_addr(int q, int slot) {
  return q * (QC * 2) + ((SWZ ? (slot ^ ((q >> 1) & 3)) : slot) << 4);
}

#define TS(i)                                                                                          \
  do {                                                                                                 \
    if (CGAN_TSBUF(p) && threadIdx.x == 0)                                                                   \
      CGAN_TSBUF(p)[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)

// wait until at most N of this wave's vector-memory ops are outstanding and all its LDS ops are done, then
// workgroup barrier.  (Raw s_barrier: __syncthreads() would drain the whole LDS-DMA ring.)
#define COUNTED_BARRIER(N)                                              \
  do {                                                                  \
    asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory");    \
    __builtin_amdgcn_s_barrier();                                       \
    asm volatile("" ::: "memory");                                      \
  } while (0)

// NW = 4: one wave per SIMD and workgroup, each owning 4 pixel rows x all NCT channel tiles AND its share of the
//         hidden-map production (two workgroups per CU -> 2 waves per SIMD, ~250 VGPRs each).
// NW = 8: WAVE SPECIALISATION -- waves 0-3 are CONSUMERS (the 4-wave kernel's MFMA stream: fragment reads + the
//         gamma||beta MFMAs, nothing else), waves 4-7 are PRODUCERS (the weight-stage LDS-DMA and the next quarter's
//         hidden map: cond gather -> small MFMAs -> ReLU -> LDS).  All 8 waves fit the 128-VGPR budget of 4 waves per
//         SIMD (NCT <= 4), so with two co-resident workgroups every SIMD has two MFMA streams and two producer streams to
//         pick from: the MFMA stream of a workgroup no longer waits on the producer chains of its own wave.
// GM: the training form that also writes gamma (p.gamma); a template parameter so that the inference kernels keep their
// register allocation (as a run-time branch it cost the NCT = 4 / 5 consumers 2 / 8 more spilled registers).
template <typename T, int NCT, bool C4, int NW, bool GM>
__global__ __launch_bounds__(NW * 64, NW == 8 ? CGAN_SPADE_WPE : 2) void spade_fused_kernel(SpadeParams p) {   // 2nd arg: waves per SIMD
  constexpr int WAVES = NW;               // shadows the namespace constant inside this kernel
  constexpr bool SPEC = NW == 8;          // wave-specialised: consumers 0-3, producers 4-7
  constexpr bool SWZ = !SPEC || NCT <= 4;  // hidden-map slot swizzle (see actv_addr): off where the VGPRs are needed
  constexpr int STAGE_BYTES = 3 * NCT * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* actv = smem;                                              // 2 * ACTV_Q_BYTES
  unsigned char* wbuf = actv + 2 * ACTV_Q_BYTES;                           // NBUF * STAGE_BYTES
  float* prm = reinterpret_cast<float*>(wbuf + NBUF * STAGE_BYTES);        // [NCT][32]: 16 bias | 8 mean | 8 rstd
  uint16_t* ctile = reinterpret_cast<uint16_t*>(prm + NCT * 32);           // CTH*CTW*cond_cs
  int* lut = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(ctile) +
                                    align16((size_t)CTH * CTW * p.cond_cs * 2));  // generic path only

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wrow = wave & 3;              // pixel-row group of this wave (rows 4 wrow .. 4 wrow + 3 of the tile)
  const int j = lane & 15;
  const int g = lane >> 4;

  int tile = blockIdx.x;
  const int txi = tile % p.tiles_x;
  tile /= p.tiles_x;
  const int tyi = tile % p.tiles_y;
  const int n = tile / p.tiles_y;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int nt0 = blockIdx.y * NCT;       // first channel tile of this workgroup
  TS(0);
  if (CGAN_TSBUF(p) && threadIdx.x == 0)
    CGAN_TSBUF(p)[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] =
        ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);

  // ---- LDS-DMA of weight stage s = (quarter q, dx): the three dy taps x NCT channel tiles = 3*NCT fragments of
  // 1 KiB -> buffer s & 1, fragment slot dy*NCT + c.  A channel tile past the end of the tensor is clamped (its
  // results are never stored).
  auto issue_stage = [&](int s) {
    const int q = s / 3, dx = s - q * 3;
    unsigned char* dstbuf = wbuf + (s & 1) * STAGE_BYTES;
#pragma unroll
    for (int i0 = 0; i0 < 3 * NCT; i0 += 4) {
      const int i = i0 + (wave & 3);      // wave-uniform; 4 issuing waves (all of them, or the producers)
      if (i < 3 * NCT) {
        const int dy = i / NCT, c = i - dy * NCT;
        const int nt = min(nt0 + c, p.nt - 1);
        const u32x4* src = p.w_gb + ((size_t)nt * KS_GB + (dy * 3 + dx) * 4 + q) * 64 + lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dstbuf + i * 1024), 16, 0, 0);
      }
    }
  };
  if (!SPEC || wave >= 4) issue_stage(0);

  // shared-conv weights of one hidden-channel quarter, kept in registers (C4 path): one fragment per 16-channel tile
  u32x4 wsh0[2];
  auto load_wsh = [&](int qq) {
#pragma unroll
    for (int c = 0; c < 2; ++c) wsh0[c] = p.w_sh[(size_t)(qq * 2 + c) * 64 + lane];
  };
  if (C4) load_wsh(0);

  // ---------------- phase 0: cond halo tile, epilogue parameters (+ K lookup table on the generic path)
  __builtin_amdgcn_s_setprio(2);
  {
    const int groups = p.cond_cs / 4;
    const int total = CTH * CTW * groups;
    for (int i = threadIdx.x; i < total; i += WAVES * 64) {
      int gq = i % groups;
      int q = i / groups;
      int hy = q / CTW, hx = q % CTW;
      int yy = ty0 - 2 + hy, xx = tx0 - 2 + hx;
      u32x2 v = (u32x2){0u, 0u};
      if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) {
        int sy = nearest_src(yy, p.sy, p.cond_h), sx = nearest_src(xx, p.sx, p.cond_w);
        v = *reinterpret_cast<const u32x2*>(p.cond + (((size_t)n * p.cond_h + sy) * p.cond_w + sx) * p.cond_cs + gq * 4);
      }
      *reinterpret_cast<u32x2*>(ctile + (size_t)q * p.cond_cs + gq * 4) = v;
    }
    for (int i = threadIdx.x; i < NCT * 32; i += WAVES * 64) {
      const int c = i >> 5, r = i & 31;
      const int nt = min(nt0 + c, p.nt - 1);
      float v;
      if (r < 16) v = p.b_gb[(size_t)nt * 16 + r];
      else if (r < 24) v = p.mean[(size_t)n * p.cs + nt * 8 + (r - 16)];
      else v = p.rstd[(size_t)n * p.cs + nt * 8 + (r - 24)];
      prm[i] = v;
    }
    if (!C4) {
      // K index -> element offset into the cond tile relative to the hidden pixel; -1: zero pad; -2: the
      // constant-one column that carries the shared-conv bias (weight column k = 9*cond_c)
      const int kmax = 9 * p.cond_c;
      for (int k = threadIdx.x; k < p.ksh * 32; k += WAVES * 64) {
        int off = -1;
        if (k < kmax) {
          int tap = k / p.cond_c, ch = k - tap * p.cond_c;
          off = ((tap / 3) * CTW + (tap % 3)) * p.cond_cs + ch;
        } else if (k == kmax) {
          off = -2;
        }
        lut[k] = off;
      }
    }
  }
  __syncthreads();
  TS(1);

  const uint32_t one = bits_of<T>(1.f);
  // C4 gather: K = (tap, c4) -> lane group g holds taps 2g and 2g+1 = two whole cond pixels (8 B each) at these pixel
  // offsets of the 20-wide cond tile.  The 4th channel of a cond pixel is padding, so those K slots (k = 8g + 3 and
  // 8g + 7) carry tap 8 (pixel offset 2*CTW + 2) and the bias column instead: g = 0: tap8.c0, tap8.c1;
  // g = 1: tap8.c2, the constant one; g = 2, 3: zero.
  const int tapoff0 = g == 0 ? 0 : (g == 1 ? 2 : (g == 2 ? CTW + 1 : 2 * CTW));
  const int tapoff1 = g == 0 ? 1 : (g == 1 ? CTW : (g == 2 ? CTW + 2 : 2 * CTW + 1));

  // One hidden tile = 16 halo pixels x the 32 channels of a quarter, in three pieces:
  //   gather_to(ht, b)        cond values of the tile -> the B fragment
  //   mma_to(b, acc)          2 small MFMAs: b x the shared weights held in wsh0
  //   finish_from(ht, dst, acc) ReLU, pack, store to the LDS hidden-map buffer dst
  u32x4 hb0 = (u32x4){0u, 0u, 0u, 0u};
  f32x4 hacc[2];
  hacc[0] = hacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto gather_to = [&](int ht, u32x4& b0) {
    const int q = ht * 16 + j;
    const bool qv = q < HP;
    const int qc = qv ? q : HP - 1;
    const int qy = qc / HPW, qx = qc - qy * HPW;
    const int yy = ty0 - 1 + qy, xx = tx0 - 1 + qx;
    // outside the image the hidden map is ZERO (the gamma/beta convs zero-pad actv): zero the whole B fragment,
    // bias column included, so the MFMA yields 0 and ReLU keeps it
    const bool inside = qv && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
    const uint16_t* cb = ctile + (qy * CTW + qx) * 4;
    const u32x2 p0 = *reinterpret_cast<const u32x2*>(cb + tapoff0 * 4);
    const u32x2 p1 = *reinterpret_cast<const u32x2*>(cb + tapoff1 * 4);
    const u32x2 p8 = *reinterpret_cast<const u32x2*>(cb + (2 * CTW + 2) * 4);
    // extras in their LOW halves: xa -> slot k = 8g + 3, xb -> slot k = 8g + 7
    const uint32_t xa = g == 0 ? p8[0] : (g == 1 ? p8[1] : 0u);
    const uint32_t xb = g == 0 ? (p8[0] >> 16) : (g == 1 ? one : 0u);
    // v_perm: bytes 0, 1 from the second operand, bytes 2, 3 = bytes 0, 1 of the first
    const uint32_t w1 = __builtin_amdgcn_perm(xa, p0[1], 0x05040100u);
    const uint32_t w3 = __builtin_amdgcn_perm(xb, p1[1], 0x05040100u);
    b0[0] = inside ? p0[0] : 0u; b0[1] = inside ? w1 : 0u;
    b0[2] = inside ? p1[0] : 0u; b0[3] = inside ? w3 : 0u;
  };
  auto mma_to = [&](const u32x4& b0, f32x4 (&ha)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) ha[c] = mfma16(as_vec8<T>(wsh0[c]), as_vec8<T>(b0), (f32x4){0.f, 0.f, 0.f, 0.f});
  };
  auto finish_from = [&](int ht, unsigned char* dst, const f32x4 (&ha)[2]) {
    const int q = ht * 16 + j;   // buffers hold 21 full tiles: pixels >= 324 are written (zeros) and never read
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      u32x2 o;
      o[0] = pack2<T>(fmaxf(ha[c][0], 0.f), fmaxf(ha[c][1], 0.f));
      o[1] = pack2<T>(fmaxf(ha[c][2], 0.f), fmaxf(ha[c][3], 0.f));
      // local channel = c*16 + 4g + r  ->  slot c*2 + (g>>1), byte (g&1)*8
      *reinterpret_cast<u32x2*>(dst + actv_addr<SWZ>(q, c * 2 + (g >> 1)) + (g & 1) * 8) = o;
    }
  };
  auto hid_gather = [&](int ht) { gather_to(ht, hb0); };
  auto hid_mma = [&]() { mma_to(hb0, hacc); };
  auto hid_finish = [&](int ht, unsigned char* dst) { finish_from(ht, dst, hacc); };
  // generic conditioning (cond_c > 4): K lookup table, weights from global memory; not split
  auto hidden_tile_generic = [&](int ht, int qq, unsigned char* dst) {
    const int q = ht * 16 + j;
    const bool qv = q < HP;
    const int qc = qv ? q : HP - 1;
    const int qy = qc / HPW, qx = qc - qy * HPW;
    const int yy = ty0 - 1 + qy, xx = tx0 - 1 + qx;
    const bool inside = qv && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
    const uint16_t* cbase = ctile + (qy * CTW + qx) * p.cond_cs;
    hacc[0] = hacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < p.ksh; ++ks) {
      u32x4 b;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o0 = lut[ks * 32 + g * 8 + 2 * i], o1 = lut[ks * 32 + g * 8 + 2 * i + 1];
        const uint32_t v0 = cbase[o0 > 0 ? o0 : 0], v1 = cbase[o1 > 0 ? o1 : 0];
        const uint32_t m = (o0 >= 0 ? 0xffffu : 0u) | (o1 >= 0 ? 0xffff0000u : 0u);
        const uint32_t k = (o0 == -2 ? one : 0u) | (o1 == -2 ? one << 16 : 0u);
        b[i] = inside ? (((v0 | (v1 << 16)) & m) | k) : 0u;
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const u32x4 a = p.w_sh[((size_t)(qq * 2 + c) * p.ksh + ks) * 64 + lane];
        hacc[c] = mfma16(as_vec8<T>(a), as_vec8<T>(b), hacc[c]);
      }
    }
    hid_finish(ht, dst);
  };
  auto hidden_tile = [&](int ht, int qq, unsigned char* dst) {
    if (C4) {
      hid_gather(ht);
      hid_mma();
      hid_finish(ht, dst);
    } else {
      hidden_tile_generic(ht, qq, dst);
    }
  };

  // ---------------- hidden map of quarter 0 (not overlapped with MFMA stages: keep all of a wave's tiles in flight)
  if (!(CGAN_DBG(p) & 1)) {
    if (C4) {
      constexpr int TPW = (NHT + WAVES - 1) / WAVES;   // 6 tiles per wave
      u32x4 gb0[TPW];
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        hid_gather(min(wave + i * WAVES, NHT - 1));
        gb0[i] = hb0;
      }
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int ht = wave + i * WAVES;
        hb0 = gb0[i];
        hid_mma();
        if (ht < NHT) hid_finish(ht, actv);
      }
    } else {
      for (int ht = wave; ht < NHT; ht += WAVES) hidden_tile(ht, 0, actv);
    }
  }
  __builtin_amdgcn_s_setprio(0);
  TS(2);

  // rows of tile nt: 0-7 = (1+gamma)[8nt..], 8-15 = beta[8nt..].  After the two permlane32 swaps of the
  // epilogue, lanes 0-31 (g=0,1) own channels 4g+{0,1} of the tile and lanes 32-63 (g=2,3) 4(g-2)+{2,3}.
  const int chan_in_tile = (g & 1) * 4 + (g >> 1) * 2;
  const int nchunk = min(NCT, p.nt - nt0);   // channel tiles that exist in this workgroup's chunk

  // Everything from here to the output stores is written once for "a wave that owns the channel tiles C0 .. C0 + CN - 1
  // of the workgroup's NCT" and instantiated for the whole set (NW = 4) or for the two halves of a wave pair (NW = 8).
  auto run = [&](auto c0_tag, auto cn_tag) {
    constexpr int C0 = decltype(c0_tag)::value;
    constexpr int CN = decltype(cn_tag)::value;
    constexpr int TPS = 2;                     // hidden tiles per producing wave and stage (21 per quarter, 3 stages x 4)

    f32x4 acc[CN][PT];
#pragma unroll
    for (int c = 0; c < CN; ++c)
#pragma unroll
      for (int t = 0; t < PT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---------------- K loop: 12 stages = (quarter q, dx), three dy taps each.
    // Stage s: barrier (weights of stage s landed, hidden map of quarter q visible) -> start the DMA of stage s+1 ->
    // B fragments of this dx -> three MFMA blocks.  A wave issues about one instruction per 4 cycles and a 16x16x32
    // MFMA occupies the matrix pipe for 16, so the stage body is ONE basic block in which the A-fragment reads of the
    // next block and the wave's share of the NEXT quarter's hidden map (gather / multiply / store) are interleaved with
    // the MFMAs at instruction granularity (sched_group_barrier pattern below).
    auto stage_body = [&](auto hid_tag, int q, int dx, int s) {
      constexpr bool HIDP = decltype(hid_tag)::value;
      const unsigned char* abuf = actv + (q & 1) * ACTV_Q_BYTES;   // this quarter's hidden map
      unsigned char* nbuf = actv + ((q + 1) & 1) * ACTV_Q_BYTES;   // next quarter's, written during this one
      const unsigned char* wb = wbuf + (s & 1) * STAGE_BYTES + lane * 16 + C0 * 1024;
      // hidden tiles of this stage (clamped: a duplicate tile stores identical values)
      const int htA = min((wave & 3) + (dx * TPS) * 4, NHT - 1), htB = min((wave & 3) + (dx * TPS + 1) * 4, NHT - 1);
      u32x4 bfr[PT + 2], a[CN];
#pragma unroll
      for (int r = 0; r < PT + 2; ++r) {
        const int qq = (wrow * PT + r) * HPW + (j + dx);
        bfr[r] = *reinterpret_cast<const u32x4*>(abuf + actv_addr<SWZ>(qq, g));
      }
#pragma unroll
      for (int c = 0; c < CN; ++c) a[c] = *reinterpret_cast<const u32x4*>(wb + c * 1024);
      if (HIDP) hid_gather(htA);
      // three MFMA blocks (dy).  The A fragment of the next block is fetched into the same registers right after the
      // last MFMA that reads them has been issued: one register set, LDS latency covered by the other channel tiles.
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int c = 0; c < CN; ++c) {
#pragma unroll
          for (int t = 0; t < PT; ++t) acc[c][t] = mfma16(as_vec8<T>(a[c]), as_vec8<T>(bfr[t + dy]), acc[c][t]);
          if (dy < 2) a[c] = *reinterpret_cast<const u32x4*>(wb + ((dy + 1) * NCT + c) * 1024);
        }
        if (HIDP) {
          if (dy == 0) {
            hid_mma();
          } else if (dy == 1) {
            hid_finish(htA, nbuf);
            if (TPS == 2) hid_gather(htB);
          } else if (TPS == 2) {
            hid_mma();
            hid_finish(htB, nbuf);
          }
        }
      }
      // issue order: one MFMA, then up to three other instructions (VALU / LDS / SALU), repeated
#pragma unroll
      for (int i = 0; i < 3 * CN * PT + 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // VALU
        __builtin_amdgcn_sched_group_barrier(0x180, 1, 0);   // DS read/write
      }
    };

    // plain stage (4-wave kernel with NCT >= 4): MFMA blocks with the hidden pieces between them, no forced interleave
    auto stage_plain = [&](int q, int dx, int s) {
      const unsigned char* abuf = actv + (q & 1) * ACTV_Q_BYTES;
      unsigned char* nbuf = actv + ((q + 1) & 1) * ACTV_Q_BYTES;
      const bool hid = !SPEC && C4 && q < 3 && !(CGAN_DBG(p) & 1);
      u32x4 bfr[PT + 2];
#pragma unroll
      for (int r = 0; r < PT + 2; ++r) {
        const int qq = (wrow * PT + r) * HPW + (j + dx);
        bfr[r] = *reinterpret_cast<const u32x4*>(abuf + actv_addr<SWZ>(qq, g));
      }
      const int htA = (wave & 3) + (dx * 2) * 4, htB = (wave & 3) + (dx * 2 + 1) * 4;   // wave-uniform
      const unsigned char* wb = wbuf + (s & 1) * STAGE_BYTES + lane * 16 + C0 * 1024;
      if (hid && htA < NHT) hid_gather(htA);
      // A fragments in flight: all CN, or (specialised consumers with 5 tiles: 80 accumulator + 24 B registers of a
      // 128-register budget) three at a time
      constexpr int AC = (SPEC && CN > 4) ? 2 : CN;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int c0 = 0; c0 < CN; c0 += AC) {
          u32x4 a[AC];
#pragma unroll
          for (int c = 0; c < AC; ++c)
            if (c0 + c < CN) a[c] = *reinterpret_cast<const u32x4*>(wb + (dy * NCT + c0 + c) * 1024);
#pragma unroll
          for (int c = 0; c < AC; ++c)
            if (c0 + c < CN) {
#pragma unroll
              for (int t = 0; t < PT; ++t)
                acc[c0 + c][t] = mfma16(as_vec8<T>(a[c]), as_vec8<T>(bfr[t + dy]), acc[c0 + c][t]);
            }
        }
        if (hid) {
          if (dy == 0) {
            if (htA < NHT) hid_mma();
          } else if (dy == 1) {
            if (htA < NHT) hid_finish(htA, nbuf);
            if (htB < NHT) hid_gather(htB);
          } else if (htB < NHT) {
            hid_mma();
            hid_finish(htB, nbuf);
          }
        }
      }
    };

    for (int q = 0; q < 4; ++q) {
      if (!SPEC && C4 && q < 3) load_wsh(q + 1);
      for (int dx = 0; dx < 3; ++dx) {
        const int s = q * 3 + dx;
        COUNTED_BARRIER(0);
        if (!SPEC && s + 1 < NSTAGES && !(CGAN_DBG(p) & 4)) issue_stage(s + 1);
        if (q == 3 && dx == 0) {
          // the last quarter has no successor: its spare hidden-map buffer (actv[0]) receives the x tile now, as
          // whole 16-byte channel chunks, lane-linear over [256 pixels][NCT chunks] (id = k*NW*64 + tid -> pixel
          // id / NCT, chunk id % NCT), so the epilogue finds x in LDS
#pragma unroll
          for (int k = 0; k < (NCT * 256 + WAVES * 64 - 1) / (WAVES * 64); ++k) {
            const int id = k * (WAVES * 64) + threadIdx.x;
            if ((k + 1) * (WAVES * 64) <= NCT * 256 || id < NCT * 256) {
              const int pix = id / NCT, cc = id - pix * NCT;
              const int yy = min(ty0 + (pix >> 4), p.h - 1), xx = min(tx0 + (pix & 15), p.w - 1);
              const int sy = p.x_ups ? (yy >> 1) : yy, sx = p.x_ups ? (xx >> 1) : xx;
              const int nt = min(nt0 + cc, p.nt - 1);
              const uint16_t* src = p.x + (((size_t)n * p.hx + sy) * p.wx + sx) * p.cs + nt * 8;
              __builtin_amdgcn_global_load_lds(
                  (const __attribute__((address_space(1))) void*)src,
                  (__attribute__((address_space(3))) void*)(actv + (k * (WAVES * 64) + wave * 64) * 16), 16, 0, 0);
            }
          }
        }
        // the interleaved body only pays where a wave has hidden-map work to weave in and VGPRs to spare: the 4-wave
        // kernel with NCT <= 3; the specialised consumers (nothing but fragment reads and MFMAs, two more waves on the
        // SIMD to cover the reads) take the plain one
        constexpr bool ILV = !SPEC && NCT <= 3;
        if (ILV) {
          if (!SPEC && C4 && q < 3 && !(CGAN_DBG(p) & 1)) stage_body(std::true_type{}, q, dx, s);
          else stage_body(std::false_type{}, q, dx, s);
        } else {
          stage_plain(q, dx, s);
        }
        if (!SPEC && !C4 && q < 3) {   // generic conditioning: not interleaved
          unsigned char* nbuf = actv + ((q + 1) & 1) * ACTV_Q_BYTES;
          for (int ht = wave + dx * TPS * 4; ht < min(NHT, (dx + 1) * TPS * 4); ht += 4)
            hidden_tile_generic(ht, q + 1, nbuf);
        }
      }
    }
    TS(5);

    // ---------------- epilogue.  The hidden-map region of LDS is free now: use it to turn the lane-linear x chunks
    // into per-lane values and the per-lane results back into lane-linear chunks.
    // x already sits in LDS (actv[0], DMA'd during the last quarter; the stage barriers since then made it visible);
    // every lane reads and rewrites only its own 4-byte slots, so no barrier is needed before the arithmetic
    __builtin_amdgcn_s_setprio(2);
    const bool lrelu_max = p.slope >= 0.f && p.slope <= 1.f;
    unsigned char* xt = actv;   // [256 px][NCT * 16 B]
    // training (p.gamma): gamma leaves through the OTHER hidden buffer in the same layout -- the backward reads it instead
    // of re-running the 128 -> C convolution over a re-materialised hidden map.  That buffer is free once every wave has
    // left the K loop: one more workgroup barrier (the producers meet it at the end of run_producer).
    unsigned char* gt = actv + ACTV_Q_BYTES;
    static_assert(MAX_NCT * 4096 <= ACTV_Q_BYTES, "a 256-pixel x NCT-tile staging block fits one hidden buffer");
    if (GM) COUNTED_BARRIER(0);
#pragma unroll
    for (int c = 0; c < CN; ++c) {
      const int cg = C0 + c;      // channel tile within the workgroup's chunk
      const int ch = (nt0 + cg) * 8 + chan_in_tile;
      const bool tile_pad = (nt0 + cg) * 8 + 8 > p.c;
      const f32x4 eb = *reinterpret_cast<const f32x4*>(prm + cg * 32 + g * 4);
      const float em0 = prm[cg * 32 + 16 + chan_in_tile], em1 = prm[cg * 32 + 17 + chan_in_tile];
      const float er0 = prm[cg * 32 + 24 + chan_in_tile], er1 = prm[cg * 32 + 25 + chan_in_tile];
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        unsigned a0 = __builtin_bit_cast(unsigned, acc[c][t][0] + eb[0]);
        unsigned a1 = __builtin_bit_cast(unsigned, acc[c][t][1] + eb[1]);
        unsigned a2 = __builtin_bit_cast(unsigned, acc[c][t][2] + eb[2]);
        unsigned a3 = __builtin_bit_cast(unsigned, acc[c][t][3] + eb[3]);
        auto s02 = __builtin_amdgcn_permlane32_swap(a0, a2, false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(a1, a3, false, false);
        const float gm0 = __builtin_bit_cast(float, (unsigned)s02[0]);
        const float bt0 = __builtin_bit_cast(float, (unsigned)s02[1]);
        const float gm1 = __builtin_bit_cast(float, (unsigned)s13[0]);
        const float bt1 = __builtin_bit_cast(float, (unsigned)s13[1]);
        const int pix = (wrow * PT + t) * 16 + j;
        uint32_t* slot = reinterpret_cast<uint32_t*>(xt + (pix * NCT + cg) * 16 + chan_in_tile * 2);
        float x0, x1;
        unpack2<T>(*slot, x0, x1);
        float o0 = (x0 - em0) * er0 * gm0 + bt0;
        float o1 = (x1 - em1) * er1 * gm1 + bt1;
        if (p.act == CGAN_ACT_LRELU) {
          if (lrelu_max) {            // wave-uniform: slope in [0, 1] -> max(o, slope * o), two instructions instead of three
            o0 = fmaxf(o0, o0 * p.slope);
            o1 = fmaxf(o1, o1 * p.slope);
          } else {
            o0 = o0 > 0.f ? o0 : o0 * p.slope;
            o1 = o1 > 0.f ? o1 : o1 * p.slope;
          }
        }
        if (tile_pad) {               // wave-uniform: only the last channel tile of a layer has pad channels
          if (ch >= p.c) o0 = 0.f;
          if (ch + 1 >= p.c) o1 = 0.f;
        }
        *slot = pack2<T>(o0, o1);   // each lane reads and rewrites only its own 4 bytes
        if (GM) {
          float g0 = gm0 - 1.f, g1 = gm1 - 1.f;      // the packed bias carries the "1 +"
          if (tile_pad) {
            if (ch >= p.c) g0 = 0.f;
            if (ch + 1 >= p.c) g1 = 0.f;
          }
          *reinterpret_cast<uint32_t*>(gt + (pix * NCT + cg) * 16 + chan_in_tile * 2) = pack2<T>(g0, g1);
        }
      }
    }
  };
  // producer waves of the specialised kernel: the same barrier sequence as the consumers' K loop, with the weight-stage
  // DMA and the hidden-map tiles of the next quarter in between
  auto run_producer = [&]() {
    const int pw = wave & 3;
    // The cond values a hidden tile is computed from do not depend on the quarter (only the shared-conv weights do):
    // gather the B fragments of this wave's six tiles ONCE and keep them in registers (48 VGPRs the producers have to
    // spare); per quarter a tile then costs 2 small MFMAs + ReLU / pack + two LDS stores.
    u32x4 gb0[6];
    if (C4) {
#pragma unroll
      for (int k = 0; k < 6; ++k) gather_to(min(pw + (k >> 1) * 8 + (k & 1) * 4, NHT - 1), gb0[k]);
    }
    for (int q = 0; q < 4; ++q) {
      if (C4 && q < 3) load_wsh(q + 1);
      for (int dx = 0; dx < 3; ++dx) {
        const int s = q * 3 + dx;
        COUNTED_BARRIER(0);
        if (s + 1 < NSTAGES && !(CGAN_DBG(p) & 4)) issue_stage(s + 1);
        if (q == 3 && dx == 0) {       // x tile -> the idle hidden buffer (all 8 waves take part; see the consumers)
#pragma unroll
          for (int k = 0; k < (NCT * 256 + WAVES * 64 - 1) / (WAVES * 64); ++k) {
            const int id = k * (WAVES * 64) + threadIdx.x;
            if ((k + 1) * (WAVES * 64) <= NCT * 256 || id < NCT * 256) {
              const int pix = id / NCT, cc = id - pix * NCT;
              const int yy = min(ty0 + (pix >> 4), p.h - 1), xx = min(tx0 + (pix & 15), p.w - 1);
              const int sy = p.x_ups ? (yy >> 1) : yy, sx = p.x_ups ? (xx >> 1) : xx;
              const int nt = min(nt0 + cc, p.nt - 1);
              const uint16_t* src = p.x + (((size_t)n * p.hx + sy) * p.wx + sx) * p.cs + nt * 8;
              __builtin_amdgcn_global_load_lds(
                  (const __attribute__((address_space(1))) void*)src,
                  (__attribute__((address_space(3))) void*)(actv + (k * (WAVES * 64) + wave * 64) * 16), 16, 0, 0);
            }
          }
        }
        if (q < 3 && !(CGAN_DBG(p) & 1)) {
          unsigned char* nbuf = actv + ((q + 1) & 1) * ACTV_Q_BYTES;
          if (C4) {
            // the two tiles of this stage in lock step (gather | gather, multiply | multiply, store | store): the
            // LDS round trip of one covers the dependent MFMA pair of the other.  A tile index past the last one is
            // clamped: the duplicate stores identical values.
            const int htA = min(pw + dx * 8, NHT - 1), htB = min(pw + dx * 8 + 4, NHT - 1);
            f32x4 ha[2], hb[2];
            // dx is a run-time loop variable: pick the register pair with selects (no dynamic VGPR indexing)
            const u32x4 a0 = dx == 0 ? gb0[0] : (dx == 1 ? gb0[2] : gb0[4]);
            const u32x4 b0 = dx == 0 ? gb0[1] : (dx == 1 ? gb0[3] : gb0[5]);
            mma_to(a0, ha);
            mma_to(b0, hb);
            finish_from(htA, nbuf, ha);
            finish_from(htB, nbuf, hb);
          } else {
            for (int ht = pw + dx * 8; ht < min(NHT, (dx + 1) * 8); ht += 4) hidden_tile_generic(ht, q + 1, nbuf);
          }
        }
      }
    }
    TS(5);
    if (GM) COUNTED_BARRIER(0);   // the consumers' "every wave has left the K loop" barrier (epilogue, training)
  };
  if (!SPEC || wave < 4) {
    run(std::integral_constant<int, 0>{}, std::integral_constant<int, NCT>{});
  } else {
    run_producer();
  }
  __syncthreads();
  const unsigned char* xt = actv;   // [256 px][NCT * 16 B], now holding the results
  if (!(CGAN_DBG(p) & 8)) {
#pragma unroll
    for (int k = 0; k < (NCT * 256 + WAVES * 64 - 1) / (WAVES * 64); ++k) {
      const int id = k * (WAVES * 64) + threadIdx.x;
      const int pix = id / NCT, cc = id - pix * NCT;
      const int yy = ty0 + (pix >> 4), xx = tx0 + (pix & 15);
      if (id < NCT * 256 && yy < p.h && xx < p.w && cc < nchunk)
        CGAN_ST_STREAM(*reinterpret_cast<const u32x4*>(xt + id * 16),
                       reinterpret_cast<u32x4*>(p.y + (((size_t)n * p.h + yy) * p.w + xx) * p.cs + (nt0 + cc) * 8));
    }
    if (GM) {
      const unsigned char* gts = actv + ACTV_Q_BYTES;
#pragma unroll
      for (int k = 0; k < (NCT * 256 + WAVES * 64 - 1) / (WAVES * 64); ++k) {
        const int id = k * (WAVES * 64) + threadIdx.x;
        const int pix = id / NCT, cc = id - pix * NCT;
        const int yy = ty0 + (pix >> 4), xx = tx0 + (pix & 15);
        if (id < NCT * 256 && yy < p.h && xx < p.w && cc < nchunk)
          *reinterpret_cast<u32x4*>(p.gamma + (((size_t)n * p.h + yy) * p.w + xx) * p.cs + (nt0 + cc) * 8) =
              *reinterpret_cast<const u32x4*>(gts + id * 16);
      }
    }
  }
  TS(6);
}

// ---- weight packing
// shared conv, C4 path (cond_c <= 3): one k-step, K = tap*4 + c for taps 0..7 and c < 3; the 4th-channel slots hold
//   tap 8 (k = 3, 7, 11) and the bias (k = 15): [8 hidden tiles][64 lanes] fragments
// shared conv, generic path: K = tap*cond_c + c, bias in column 9*cond_c -> [8][ksh][64] fragments
// gamma||beta: tile t rows 0-7 = gamma[8t+i], rows 8-15 = beta[8t+i]; K = tap*128 + hidden -> [nt][36][64]
template <typename T>
__global__ void spade_pack_kernel(const float* __restrict__ w_sh, const float* __restrict__ b_sh,
                                  const float* __restrict__ w_g, const float* __restrict__ b_g,
                                  const float* __restrict__ w_b, const float* __restrict__ b_b,
                                  uint16_t* __restrict__ p_wsh, uint16_t* __restrict__ p_wgb, float* __restrict__ p_bgb,
                                  int c, int nt, int cond_c, int ksh, int c4) {
  const int n_sh = 8 * ksh * 64;
  const int n_gb = nt * KS_GB * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_sh + n_gb; idx += gridDim.x * blockDim.x) {
    uint16_t o[8];
    u32x4* dst;
    if (idx < n_sh) {
      if (c4) {
        const int lane = idx & 63, ct = idx >> 6;
        const int hc = ct * 16 + (lane & 15);
        const int gg = lane >> 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = gg * 8 + e, tap = k >> 2, cc = k & 3;
          float v = 0.f;
          if (cc < 3) {
            v = cc < cond_c ? w_sh[((size_t)hc * cond_c + cc) * 9 + tap] : 0.f;
          } else if (gg < 2) {                       // the 4th-channel slots of taps 0..3: tap 8 and the bias
            const int x = gg * 2 + (e >> 2);         // 0, 1, 2 -> tap 8 channel x;  3 -> bias
            if (x < 3) v = x < cond_c ? w_sh[((size_t)hc * cond_c + x) * 9 + 8] : 0.f;
            else v = b_sh[hc];
          }
          o[e] = bits_of<T>(v);
        }
      } else {
        const int lane = idx & 63;
        const int ks = (idx >> 6) % ksh;
        const int ct = (idx >> 6) / ksh;
        const int hc = ct * 16 + (lane & 15);
        const int k0 = ks * 32 + (lane >> 4) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = k0 + e;
          float v = 0.f;
          if (k < 9 * cond_c) {
            const int tap = k / cond_c, cc = k - tap * cond_c;
            v = w_sh[((size_t)hc * cond_c + cc) * 9 + tap];
          } else if (k == 9 * cond_c) {
            v = b_sh[hc];  // bias rides on the constant-one K column
          }
          o[e] = bits_of<T>(v);
        }
      }
      dst = reinterpret_cast<u32x4*>(p_wsh) + idx;
    } else {
      const int i2 = idx - n_sh;
      const int lane = i2 & 63;
      const int ks = (i2 >> 6) % KS_GB;
      const int t = (i2 >> 6) / KS_GB;
      const int row = lane & 15;
      const int ch = t * 8 + (row & 7);
      const float* src = (row < 8) ? w_g : w_b;
      const int k0 = ks * 32 + (lane >> 4) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        const int tap = k / HID, hc = k - tap * HID;
        o[e] = bits_of<T>((ch < c) ? src[((size_t)ch * HID + hc) * 9 + tap] : 0.f);
      }
      dst = reinterpret_cast<u32x4*>(p_wgb) + i2;
    }
    u32x4 pk;
    pk[0] = o[0] | ((uint32_t)o[1] << 16);
    pk[1] = o[2] | ((uint32_t)o[3] << 16);
    pk[2] = o[4] | ((uint32_t)o[5] << 16);
    pk[3] = o[6] | ((uint32_t)o[7] << 16);
    *dst = pk;
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nt * 16; i += gridDim.x * blockDim.x) {
    const int t = i / 16, row = i % 16;
    const int ch = t * 8 + (row & 7);
    float v = 0.f;
    if (ch < c) v = (row < 8) ? (1.f + b_g[ch]) : b_b[ch];
    p_bgb[i] = v;
  }
}

int check(const CganSpadeDesc* d) {
  CGAN_REQUIRE(d != nullptr, "spade: null descriptor");
  CGAN_REQUIRE(d->dtype == CGAN_F16 || d->dtype == CGAN_BF16, "spade: bad dtype %d", d->dtype);
  CGAN_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->c > 0, "spade: bad x shape");
  CGAN_REQUIRE(d->cond_h > 0 && d->cond_w > 0 && d->cond_c > 0 && d->cond_c <= 64, "spade: bad cond shape");
  CGAN_REQUIRE(d->hidden == HID, "spade: hidden must be 128 (reference norms.py:163), got %d", d->hidden);
  CGAN_REQUIRE(d->ksize == 3, "spade: only kernel_size 3 is supported, got %d", d->ksize);
  CGAN_REQUIRE(d->act == CGAN_ACT_NONE || d->act == CGAN_ACT_LRELU, "spade: act must be NONE or LRELU");
  if (d->x_upsample) CGAN_REQUIRE((d->h % 2) == 0 && (d->w % 2) == 0, "spade: x_upsample needs even h/w");
  return CGAN_OK;
}

template <typename T, int NCT, bool C4, int NW, bool GM>
int launch_gm(const SpadeParams& p0, hipStream_t s) {
  SpadeParams p = p0;
  p.tiles_y = ceil_div(p.h, TH);
  p.tiles_x = ceil_div(p.w, TW);
  const int tiles = p.n * p.tiles_y * p.tiles_x;
  const int chunks = ceil_div(p.nt, NCT);
  size_t smem = (size_t)2 * ACTV_Q_BYTES + (size_t)NBUF * 3 * NCT * 1024 + (size_t)NCT * 32 * 4 + align16((size_t)CTH * CTW * p.cond_cs * 2) + (C4 ? 0 : (size_t)p.ksh * 32 * 4);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spade_fused_kernel<T, NCT, C4, NW, GM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("spade_fused_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((spade_fused_kernel<T, NCT, C4, NW, GM>), dim3(tiles, chunks), dim3(NW * 64), smem, s, p);
  return CGAN_OK;
}

template <typename T, int NCT, bool C4, int NW>
int launch(const SpadeParams& p, hipStream_t s) {
  return p.gamma ? launch_gm<T, NCT, C4, NW, true>(p, s) : launch_gm<T, NCT, C4, NW, false>(p, s);
}

template <typename T, bool C4>
int launch_nct(const SpadeParams& p, int nct, int nw, hipStream_t s) {
  if (nw == 8) {     // two waves per SIMD and workgroup (channel tiles split over a wave pair): needs >= 2 tiles
    switch (nct) {
      case 2: return launch<T, 2, C4, 8>(p, s);
      case 3: return launch<T, 3, C4, 8>(p, s);
      case 4: return launch<T, 4, C4, 8>(p, s);
      case 5: return launch<T, 5, C4, 8>(p, s);
      default: break;
    }
  }
  switch (nct) {
    case 1: return launch<T, 1, C4, 4>(p, s);
    case 2: return launch<T, 2, C4, 4>(p, s);
    case 3: return launch<T, 3, C4, 4>(p, s);
    case 4: return launch<T, 4, C4, 4>(p, s);
    default: return launch<T, 5, C4, 4>(p, s);
  }
}

// Development knobs (not part of the stable ABI): force the channel tiles per workgroup / ablation bits /
// timestamp buffer.
CGAN_KNOB(int, g_spade_variant, 0);
CGAN_KNOB(int, g_spade_waves, 8);   // default: the wave-specialised kernel wherever a workgroup has >= 2 channel tiles
CGAN_KNOB(int, g_spade_dbg, 0);
CGAN_KNOB(unsigned long long*, g_spade_tsbuf, nullptr);

}  // namespace

CGAN_DEV_ONLY(extern "C" void cgan_debug_set_spade_variant(int v) { g_spade_variant = v; })
CGAN_DEV_ONLY(extern "C" void cgan_debug_set_spade_waves(int v) { g_spade_waves = v == 4 ? 4 : 8; })
CGAN_DEV_ONLY(extern "C" void cgan_debug_set_spade_ablation(int bits) { g_spade_dbg = bits; })
CGAN_DEV_ONLY(extern "C" void cgan_debug_set_spade_tsbuf(void* p) { g_spade_tsbuf = (unsigned long long*)p; })

extern "C" size_t cgan_spade_packed_weight_bytes(const CganSpadeDesc* d) {
  if (check(d) != CGAN_OK) return 0;
  return packed_layout(cgan_cs(d->c), d->cond_c).total;
}

extern "C" int cgan_spade_pack_weights(const float* w_shared, const float* b_shared, const float* w_gamma,
                                       const float* b_gamma, const float* w_beta, const float* b_beta, void* packed,
                                       const CganSpadeDesc* d, void* stream) {
  int rc = check(d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(w_shared && b_shared && w_gamma && b_gamma && w_beta && b_beta && packed, "spade_pack_weights: null pointer");
  const int cs = cgan_cs(d->c), nt = cs / 8, ksh = ksh_of(d->cond_c);
  const bool c4 = is_c4(d->cond_c);
  PackedLayout L = packed_layout(cs, d->cond_c);
  unsigned char* base = (unsigned char*)packed;
  const int total = 8 * ksh * 64 + nt * KS_GB * 64;
  const int blocks = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == CGAN_F16)
    hipLaunchKernelGGL(spade_pack_kernel<F16>, dim3(blocks), dim3(256), 0, s, w_shared, b_shared, w_gamma, b_gamma,
                       w_beta, b_beta, (uint16_t*)(base + L.w_sh), (uint16_t*)(base + L.w_gb), (float*)(base + L.b_gb),
                       d->c, nt, d->cond_c, ksh, (int)c4);
  else
    hipLaunchKernelGGL(spade_pack_kernel<BF16>, dim3(blocks), dim3(256), 0, s, w_shared, b_shared, w_gamma, b_gamma,
                       w_beta, b_beta, (uint16_t*)(base + L.w_sh), (uint16_t*)(base + L.w_gb), (float*)(base + L.b_gb),
                       d->c, nt, d->cond_c, ksh, (int)c4);
  CGAN_CHECK_LAUNCH("spade_pack_weights");
  return CGAN_OK;
}

extern "C" int cgan_spade_fused_fwd(const void* x, const float* mean, const float* rstd, const void* cond,
                                    const void* packed, void* y, const CganSpadeDesc* d, void* stream) {
  return cgan_spade_fused_fwd_train(x, mean, rstd, cond, packed, y, nullptr, d, stream);
}

extern "C" int cgan_spade_fused_fwd_train(const void* x, const float* mean, const float* rstd, const void* cond,
                                          const void* packed, void* y, void* gamma_out, const CganSpadeDesc* d,
                                          void* stream) {
  int rc = check(d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(x && mean && rstd && cond && packed && y, "spade_fused_fwd: null pointer");
  SpadeParams p;
  const int cs = cgan_cs(d->c);
  PackedLayout L = packed_layout(cs, d->cond_c);
  const unsigned char* base = (const unsigned char*)packed;
  p.x = (const uint16_t*)x; p.mean = mean; p.rstd = rstd; p.cond = (const uint16_t*)cond;
  p.w_sh = (const u32x4*)(base + L.w_sh);
  p.w_gb = (const u32x4*)(base + L.w_gb); p.b_gb = (const float*)(base + L.b_gb);
  p.y = (uint16_t*)y; p.gamma = (uint16_t*)gamma_out;
  p.n = d->n; p.h = d->h; p.w = d->w; p.c = d->c; p.cs = cs; p.nt = cs / 8;
  p.x_ups = d->x_upsample; p.hx = d->x_upsample ? d->h / 2 : d->h; p.wx = d->x_upsample ? d->w / 2 : d->w;
  p.cond_h = d->cond_h; p.cond_w = d->cond_w; p.cond_c = d->cond_c; p.cond_cs = cgan_cond_cs(d->cond_c);
  p.ksh = ksh_of(d->cond_c);
  p.sy = (float)d->cond_h / (float)d->h; p.sx = (float)d->cond_w / (float)d->w;
  p.act = d->act; p.slope = d->act_slope; CGAN_DEV_ONLY(p.dbg = g_spade_dbg; p.tsbuf = g_spade_tsbuf;)
  hipStream_t s = (hipStream_t)stream;
  // Channel tiles per workgroup (3 .. 5 when the layer has that many).  A workgroup costs a fixed part (hidden-map
  // production, prologue, epilogue: ~2.5 tile-equivalents) plus its tiles, and the chip takes 512 workgroups per round
  // (2 per CU): minimise rounds x (k + 2.5).  On the 5^2 .. 80^2 layers of the Painter, where the grid is 1-2 rounds, this
  // picks 3 or 4 tiles instead of 5 (e.g. 320 channels at 40^2: 576 workgroups of 5 tiles = 2 rounds, 40.8 us; 720 of 4
  // tiles: 33.6 us; 640 channels at 5^2 / 10^2: 18 -> 13.7 us); large grids keep 5 (least hidden-map recomputation).
  // (tools/bench_spade.py sweep, bs 8.)
  const int kmax = p.nt < MAX_NCT ? p.nt : MAX_NCT;
  int nct = kmax;
  if (p.nt > 3) {
    const long tiles = (long)p.n * ceil_div(p.h, 16) * ceil_div(p.w, 16);
    double best = 1e30;
    for (int k = kmax; k >= 3; --k) {                       // ties keep the larger k
      const long wgs = tiles * ceil_div(p.nt, k);
      const double cost = (double)((wgs + 511) / 512) * (k + 2.5);
      if (cost < best) { best = cost; nct = k; }
    }
  }
  if (g_spade_variant >= 1 && g_spade_variant <= MAX_NCT) nct = g_spade_variant;
  const bool c4 = is_c4(d->cond_c);
  const int nw = g_spade_waves;
  if (d->dtype == CGAN_F16) rc = c4 ? launch_nct<F16, true>(p, nct, nw, s) : launch_nct<F16, false>(p, nct, nw, s);
  else rc = c4 ? launch_nct<BF16, true>(p, nct, nw, s) : launch_nct<BF16, false>(p, nct, nw, s);
  if (rc != CGAN_OK) return rc;
  CGAN_CHECK_LAUNCH("spade_fused_fwd");
  return CGAN_OK;
}
