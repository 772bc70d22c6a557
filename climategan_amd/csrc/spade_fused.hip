// Fused SPADE forward for gfx950 (reference climategan/norms.py:146-186):
//
//   seg   = nearest_resize(cond, (h, w))
//   actv  = ReLU(conv3x3(seg, w_shared) + b_shared)                    128 hidden channels
//   gamma = conv3x3(actv, w_gamma) + b_gamma ; beta = conv3x3(actv, w_beta) + b_beta
//   y     = act( (x - mean) * rstd * (1 + gamma) + beta )
//
// One workgroup (4 waves, one per SIMD; two workgroups co-resident per CU) owns a 16 x 16 pixel tile of one
// image and NCT <= 5 channel tiles (16 MFMA rows = 8 gamma + 8 beta channels each):
//   phase 0  stage the nearest-resized conditioning halo tile 20 x 20 x cond_c in LDS
//   for each HALF of the 128 hidden channels (so the LDS image is 18*18*64*2 B = 41 KB and two workgroups fit):
//     phase 1  hidden map of the 18 x 18 halo for these 64 channels: MFMA implicit GEMM with K = 9*cond_c (+1
//              constant-one column carrying the bias), ReLU, zeroed outside the image (the gamma/beta convs
//              zero-pad actv, not seg), written to LDS as [halo pixel][64 ch] with an XOR swizzle on the
//              16-byte channel slots
//     phase 2  gamma||beta implicit GEMM out of LDS, fp32 accumulate: M = NCT*16 rows, N = 256 pixels,
//              K = 9 taps x 64.  Loop order (dx, dy): the B fragments (activations) of the 6 halo rows a wave
//              needs are read once per dx and reused for the three dy taps; weights (A fragments) stream
//              L2 -> LDS by LDS-DMA, one tap per stage, double-buffered.
//   phase 3  epilogue in registers: one v_permlane32_swap pair brings gamma and beta of the same channel
//            into the same lane, which then normalises x (prefetched during phase 2) and stores 2 channels.
// The 128-channel hidden map (105 MB/img at 640x640 in 16-bit) never touches HBM; HBM traffic is
// x (read) + y (write) + the 3-channel cond halo.  LDS traffic per MFMA is (30 A + 12 B) / 120 = 0.35
// fragment reads, which keeps the kernel under the LDS-read roof (measured ~128 B/clk/CU for ds_read_b128).
//
// MFMA operand roles: A = weights (rows = output channels), B = activations (cols = pixels), so that
// D's per-lane 4 registers are 4 consecutive channel rows of one pixel (col = lane&15, row = 4*(lane>>4)+r).
#include "cgan_common.h"

namespace {

constexpr int TW = 16;       // pixel-tile width == MFMA N
constexpr int HID = 128;     // hidden channels (norms.py:163)
constexpr int KS_GB = 36;    // 9 taps * 128 / 32
constexpr int MAX_NCT = 5;   // channel tiles (16 rows = 8 gamma + 8 beta) per workgroup

struct SpadeParams {
  const uint16_t* x;
  const float* mean;
  const float* rstd;
  const uint16_t* cond;
  const u32x4* w_sh;    // [8][ksh][64]
  const float* b_sh;    // [128]
  const u32x4* w_gb;    // [nt][36][64]
  const float* b_gb;    // [nt][16]  (gamma bias + 1 | beta bias), zero on pad channels
  uint16_t* y;
  int n, h, w, c, cs, nt;
  int hx, wx, x_ups;
  int cond_h, cond_w, cond_c, cond_cs, ksh;
  float sy, sx;
  int tiles_y, tiles_x;
  int nct;              // channel tiles per workgroup (blockIdx.y selects the chunk)
  int act;
  float slope;
  unsigned long long* tsbuf;  // development: per-workgroup phase timestamps (s_memtime), or null
  int dbg;  // ablation bits (development only): 1 skip hidden map, 2 skip main MFMAs, 4 skip weight staging,
            // 8 skip epilogue stores, 16 skip epilogue prefetch, 32 skip the whole main loop
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

struct PackedLayout {
  size_t w_sh, b_sh, w_gb, b_gb, total;
};
__host__ inline PackedLayout packed_layout(int cs, int cond_c) {
  PackedLayout L;
  int ksh = ceil_div(9 * cond_c + 1, 32);
  int nt = cs / 8;
  L.w_sh = 0;
  L.b_sh = align16(L.w_sh + (size_t)8 * ksh * 64 * 16);
  L.w_gb = align16(L.b_sh + HID * sizeof(float));
  L.b_gb = align16(L.w_gb + (size_t)nt * KS_GB * 64 * 16);
  L.total = align16(L.b_gb + (size_t)nt * 16 * sizeof(float));
  return L;
}

constexpr int TH = 16;
constexpr int WAVES = 4;
constexpr int PT = TH / WAVES;                 // pixel-tile rows per wave
constexpr int HPH = TH + 2, HPW = TW + 2, HP = HPH * HPW;   // hidden halo
constexpr int CTH = TH + 4, CTW = TW + 4;                   // cond halo
constexpr int NHT = (HP + 15) / 16;
constexpr int HC = 64;                         // hidden channels resident in LDS at a time
constexpr int ACTV_BYTES = HP * HC * 2;        // 41472
constexpr int NSTAGES = 18;                    // 2 halves x 3 dx x 3 dy

__device__ __forceinline__ int actv_addr(int q, int slot) {
  // [pixel q][8 slots of 16 B], slot XOR-swizzled with bits 1..3 of q: two pixels share a 256-B bank row, so
  // 16 consecutive pixels reading the same logical slot hit 16 distinct 16-B bank groups
  return q * (HC * 2) + ((slot ^ ((q >> 1) & 7)) << 4);
}

#define TS(i)                                                                              \
  do {                                                                                     \
    if (p.tsbuf && threadIdx.x == 0)                                                       \
      p.tsbuf[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter(); \
  } while (0)

template <typename T, int NCT, bool KSH1>
__global__ __launch_bounds__(WAVES * 64, 2) void spade_fused_kernel(SpadeParams p) {
  constexpr int STAGE_BYTES = NCT * 2 * 1024;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* actv = smem;                                             // ACTV_BYTES
  unsigned char* wbuf = smem + ACTV_BYTES;                                // 2 * STAGE_BYTES
  uint16_t* ctile = reinterpret_cast<uint16_t*>(wbuf + 2 * STAGE_BYTES);  // CTH*CTW*cond_cs
  int* lut = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(ctile) +
                                    align16((size_t)CTH * CTW * p.cond_cs * 2));
  float* prm = reinterpret_cast<float*>(lut + p.ksh * 32);  // [NCT][32]: 16 bias rows | 8 mean | 8 rstd
  unsigned char* wshl = reinterpret_cast<unsigned char*>(prm + NCT * 32);  // KSH1: 8 shared-conv fragments

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15;
  const int g = lane >> 4;

  int tile = blockIdx.x;
  const int txi = tile % p.tiles_x;
  tile /= p.tiles_x;
  const int tyi = tile % p.tiles_y;
  const int n = tile / p.tiles_y;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int nt0 = blockIdx.y * NCT;       // first channel tile of this workgroup

  // LDS-DMA of weight stage s = (half, dx, dy) -> buffer s&1: NCT tiles x 2 k-steps, 1 KiB each
  auto issue_stage = [&](int s) {
    const int h = s / 9, r = s - h * 9, dx = r / 3, dy = r - dx * 3;
    const int ks0 = (dy * 3 + dx) * 4 + h * 2;
    unsigned char* dstbuf = wbuf + (s & 1) * STAGE_BYTES;
#pragma unroll
    for (int i0 = 0; i0 < NCT * 2; i0 += WAVES) {
      const int i = i0 + wave;            // fragment slot: c = i/2, kc = i%2   (wave-uniform)
      if (i < NCT * 2) {
        const int c = i >> 1, kc = i & 1;
        if (nt0 + c < p.nt) {
          const u32x4* src = p.w_gb + ((size_t)(nt0 + c) * KS_GB + ks0 + kc) * 64 + lane;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dstbuf + i * 1024), 16, 0, 0);
        }
      }
    }
  };

  TS(0);
  issue_stage(0);  // overlaps phases 0 and 1
  if (KSH1) {
    // the whole shared-conv weight matrix (128 x 32 incl. the bias column) = 8 fragments: LDS-DMA them once
#pragma unroll
    for (int i0 = 0; i0 < 8; i0 += WAVES) {
      const u32x4* src = p.w_sh + (size_t)(i0 + wave) * 64 + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(wshl + (i0 + wave) * 1024), 16, 0, 0);
    }
  }

  // ---------------- phase 0: cond halo tile + K lookup table
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
    // K index -> element offset into the cond tile relative to the hidden pixel; -1: zero pad; -2: the
    // constant-one column that carries the shared-conv bias (packed as weight column k = 9*cond_c)
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
    // epilogue parameters of this workgroup's channel tiles
    for (int i = threadIdx.x; i < NCT * 32; i += WAVES * 64) {
      const int c = i >> 5, r = i & 31;
      const int nt = min(nt0 + c, p.nt - 1);
      float v;
      if (r < 16) v = p.b_gb[(size_t)nt * 16 + r];
      else if (r < 24) v = p.mean[(size_t)n * p.cs + nt * 8 + (r - 16)];
      else v = p.rstd[(size_t)n * p.cs + nt * 8 + (r - 24)];
      prm[i] = v;
    }
  }
  __syncthreads();

  TS(1);
  // Shared-conv B fragment of a hidden pixel = 8 gathered cond values per lane.  Branch-free: every slot reads
  // cbase[max(off,0)], then (value & mask) | konst zeroes the pad slots and plants the constant one of the bias
  // column.  With cond_c = 3 (KSH1) offsets/masks are loop-invariant and live in registers.
  const uint32_t one = bits_of<T>(1.f);
  int off0[8];
  u32x4 msk0 = (u32x4){0u, 0u, 0u, 0u}, kon0 = (u32x4){0u, 0u, 0u, 0u};
  if (KSH1) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int off = lut[g * 8 + i];
      off0[i] = off > 0 ? off : 0;
      const uint32_t sh = (i & 1) * 16;
      msk0[i >> 1] |= (off >= 0 ? 0xffffu : 0u) << sh;
      kon0[i >> 1] |= (off == -2 ? one : 0u) << sh;
    }
  }

  // per-lane output pixel bookkeeping (pixel-tile row t of this wave, column j)
  // x / y tile traffic is done in whole 16-byte channel chunks, lane-linear over the tile's
  // [256 pixels][NCT chunks] so that a wave touches contiguous memory; chunk k of this thread:
  //   id = k*256 + tid -> pixel id / NCT (row-major in the 16x16 tile), channel chunk id % NCT
  const int nchunk = min(NCT, p.nt - nt0);   // channel tiles that exist in this workgroup's chunk
  // rows of tile nt: 0-7 = (1+gamma)[8nt..], 8-15 = beta[8nt..].  After the two permlane32 swaps of the
  // epilogue, lanes 0-31 (g=0,1) own channels 4g+{0,1} of the tile and lanes 32-63 (g=2,3) 4(g-2)+{2,3}.
  const int chan_in_tile = (g & 1) * 4 + (g >> 1) * 2;

  f32x4 acc[NCT][PT];
#pragma unroll
  for (int c = 0; c < NCT; ++c)
#pragma unroll
    for (int t = 0; t < PT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // epilogue operands, fetched during the second half so their latency hides behind the MFMA stages
  u32x4 xv[NCT];

  int s = 0;
  for (int h = 0; h < 2; ++h) {
    if (h) {
      TS(3);
      __syncthreads();  // every wave is done reading the previous half's hidden map
    }

    // ---------------- phase 1: hidden channels [64h, 64h+64) of the halo into LDS
    // Short, latency-bound code that shares its SIMD with the other resident workgroup's MFMA stream: run it
    // at raised priority and keep two hidden tiles in flight per wave.
    __builtin_amdgcn_s_setprio(2);
    if (!(p.dbg & 1)) {
      u32x4 wsh[4];
      if (KSH1) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          wsh[c] = *reinterpret_cast<const u32x4*>(wshl + (h * 4 + c) * 1024 + lane * 16);
      }
      // B fragment (cond gather) of hidden tile ht for K-step ks
      auto gather = [&](int ht, int ks) -> u32x4 {
        const int q = min(ht * 16 + j, HP - 1);
        const int qy = q / HPW, qx = q - qy * HPW;
        const uint16_t* cbase = ctile + (qy * CTW + qx) * p.cond_cs;
        u32x4 b;
        if (KSH1) {
          uint32_t e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] = cbase[off0[i]];
#pragma unroll
          for (int i = 0; i < 4; ++i) b[i] = ((e[2 * i] | (e[2 * i + 1] << 16)) & msk0[i]) | kon0[i];
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int o0 = lut[ks * 32 + g * 8 + 2 * i], o1 = lut[ks * 32 + g * 8 + 2 * i + 1];
            const uint32_t v0 = cbase[o0 > 0 ? o0 : 0], v1 = cbase[o1 > 0 ? o1 : 0];
            const uint32_t m = (o0 >= 0 ? 0xffffu : 0u) | (o1 >= 0 ? 0xffff0000u : 0u);
            const uint32_t k = (o0 == -2 ? one : 0u) | (o1 == -2 ? one << 16 : 0u);
            b[i] = ((v0 | (v1 << 16)) & m) | k;
          }
        }
        return b;
      };
      // ReLU, zero outside the image, pack and store the 4 channel tiles of hidden tile ht
      auto finish = [&](int ht, const f32x4 (&hacc)[4]) {
        const int q = ht * 16 + j;
        const bool qv = q < HP;
        const int qc = qv ? q : HP - 1;
        const int qy = qc / HPW, qx = qc - qy * HPW;
        const int yy = ty0 - 1 + qy, xx = tx0 - 1 + qx;
        const bool inside = qv && yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = hacc[c][r];
            v[r] = (inside && t > 0.f) ? t : 0.f;
          }
          u32x2 o;
          o[0] = pack2<T>(v[0], v[1]);
          o[1] = pack2<T>(v[2], v[3]);
          if (qv) *reinterpret_cast<u32x2*>(actv + actv_addr(qc, c * 2 + (g >> 1)) + (g & 1) * 8) = o;
        }
      };
      constexpr int TILES_PER_WAVE = (NHT + WAVES - 1) / WAVES;   // 6
#pragma unroll
      for (int it = 0; it < TILES_PER_WAVE; it += 2) {
        const int htA = wave + it * WAVES, htB = wave + (it + 1) * WAVES;   // wave-uniform
        f32x4 accA[4], accB[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) accA[c] = accB[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int nks = KSH1 ? 1 : p.ksh;
        for (int ks = 0; ks < nks; ++ks) {
          const u32x4 bA = gather(htA, ks);
          const u32x4 bB = gather(htB, ks);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const u32x4 a = KSH1 ? wsh[c] : p.w_sh[((size_t)(h * 4 + c) * p.ksh + ks) * 64 + lane];
            accA[c] = mfma16(as_vec8<T>(a), as_vec8<T>(bA), accA[c]);
            accB[c] = mfma16(as_vec8<T>(a), as_vec8<T>(bB), accB[c]);
          }
        }
        if (htA < NHT) finish(htA, accA);
        if (htB < NHT) finish(htB, accB);
      }
    }
    __builtin_amdgcn_s_setprio(0);

    TS(2 + 2 * h);
    // ---------------- phase 2: 9 taps x 2 k-steps of this half
    for (int dx = 0; dx < 3; ++dx) {
      u32x4 bfr[PT + 2][2];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        // stage s has landed (vmcnt(0) before the barrier) and the other buffer is free; the first barrier of
        // a half also publishes the hidden map
        __syncthreads();
        if (s + 1 < NSTAGES) issue_stage(s + 1);
        if (dy == 0) {
          if (h == 1 && dx == 0) {
#pragma unroll
            for (int k = 0; k < NCT; ++k) {
              const int id = k * (WAVES * 64) + threadIdx.x;
              const int pix = id / NCT, cc = id - pix * NCT;
              const int yy = min(ty0 + (pix >> 4), p.h - 1), xx = min(tx0 + (pix & 15), p.w - 1);
              const int sy = p.x_ups ? (yy >> 1) : yy, sx = p.x_ups ? (xx >> 1) : xx;
              const int nt = min(nt0 + cc, p.nt - 1);
              xv[k] = *reinterpret_cast<const u32x4*>(p.x + (((size_t)n * p.hx + sy) * p.wx + sx) * p.cs + nt * 8);
            }
          }
#pragma unroll
          for (int r = 0; r < PT + 2; ++r) {
            const int q = (wave * PT + r) * HPW + (j + dx);
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
              bfr[r][kc] = *reinterpret_cast<const u32x4*>(actv + actv_addr(q, kc * 4 + g));
          }
        }
        const unsigned char* wb = wbuf + (s & 1) * STAGE_BYTES + lane * 16;
        if (!(p.dbg & 2)) {
#pragma unroll
          for (int kc = 0; kc < 2; ++kc) {
            u32x4 a[NCT];
#pragma unroll
            for (int c = 0; c < NCT; ++c) a[c] = *reinterpret_cast<const u32x4*>(wb + (c * 2 + kc) * 1024);
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
              for (int t = 0; t < PT; ++t)
                acc[c][t] = mfma16(as_vec8<T>(a[c]), as_vec8<T>(bfr[t + dy][kc]), acc[c][t]);
          }
        }
        ++s;
      }
    }
  }

  TS(5);
  // ---------------- phase 3: epilogue.  The hidden-map region of LDS is free now: use it to turn the
  // lane-linear x chunks into per-lane values and the per-lane results back into lane-linear chunks.
  __builtin_amdgcn_s_setprio(2);
  __syncthreads();
  unsigned char* xt = actv;   // [256 px][NCT * 16 B]
#pragma unroll
  for (int k = 0; k < NCT; ++k) {
    const int id = k * (WAVES * 64) + threadIdx.x;
    *reinterpret_cast<u32x4*>(xt + id * 16) = xv[k];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < NCT; ++c) {
    const int nt = nt0 + c;
    const int ch = nt * 8 + chan_in_tile;
    const f32x4 eb = *reinterpret_cast<const f32x4*>(prm + c * 32 + g * 4);
    const float em0 = prm[c * 32 + 16 + chan_in_tile], em1 = prm[c * 32 + 17 + chan_in_tile];
    const float er0 = prm[c * 32 + 24 + chan_in_tile], er1 = prm[c * 32 + 25 + chan_in_tile];
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
      const int pix = (wave * PT + t) * 16 + j;
      uint32_t* slot = reinterpret_cast<uint32_t*>(xt + (pix * NCT + c) * 16 + chan_in_tile * 2);
      float x0, x1;
      unpack2<T>(*slot, x0, x1);
      float o0 = (x0 - em0) * er0 * gm0 + bt0;
      float o1 = (x1 - em1) * er1 * gm1 + bt1;
      if (p.act == CGAN_ACT_LRELU) {
        o0 = o0 > 0.f ? o0 : o0 * p.slope;
        o1 = o1 > 0.f ? o1 : o1 * p.slope;
      }
      if (ch >= p.c) o0 = 0.f;
      if (ch + 1 >= p.c) o1 = 0.f;
      *slot = pack2<T>(o0, o1);   // same lane wrote... reads and rewrites only its own 4 bytes
    }
  }
  __syncthreads();
  if (!(p.dbg & 8)) {
#pragma unroll
    for (int k = 0; k < NCT; ++k) {
      const int id = k * (WAVES * 64) + threadIdx.x;
      const int pix = id / NCT, cc = id - pix * NCT;
      const int yy = ty0 + (pix >> 4), xx = tx0 + (pix & 15);
      if (yy < p.h && xx < p.w && cc < nchunk)
        *reinterpret_cast<u32x4*>(p.y + (((size_t)n * p.h + yy) * p.w + xx) * p.cs + (nt0 + cc) * 8) =
            *reinterpret_cast<const u32x4*>(xt + id * 16);
    }
  }
  TS(6);
}

// ---- weight packing
// shared conv: A rows = hidden channel, K = tap*cond_c + c   -> [8][ksh][64] fragments
// gamma||beta:  tile t rows 0-7 = gamma[8t+i], rows 8-15 = beta[8t+i]; K = tap*128 + hidden -> [nt][36][64]
template <typename T>
__global__ void spade_pack_kernel(const float* __restrict__ w_sh, const float* __restrict__ b_sh,
                                  const float* __restrict__ w_g, const float* __restrict__ b_g,
                                  const float* __restrict__ w_b, const float* __restrict__ b_b,
                                  uint16_t* __restrict__ p_wsh, float* __restrict__ p_bsh, uint16_t* __restrict__ p_wgb,
                                  float* __restrict__ p_bgb, int c, int nt, int cond_c, int ksh) {
  const int n_sh = 8 * ksh * 64;
  const int n_gb = nt * KS_GB * 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n_sh + n_gb; idx += gridDim.x * blockDim.x) {
    uint16_t o[8];
    u32x4* dst;
    if (idx < n_sh) {
      int lane = idx & 63;
      int ks = (idx >> 6) % ksh;
      int ct = (idx >> 6) / ksh;
      int hc = ct * 16 + (lane & 15);
      int k0 = ks * 32 + (lane >> 4) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int k = k0 + e;
        float v = 0.f;
        if (k < 9 * cond_c) {
          int tap = k / cond_c, cc = k - tap * cond_c;
          v = w_sh[((size_t)hc * cond_c + cc) * 9 + tap];
        } else if (k == 9 * cond_c) {
          v = b_sh[hc];  // bias rides on the constant-one K column
        }
        o[e] = bits_of<T>(v);
      }
      dst = reinterpret_cast<u32x4*>(p_wsh) + idx;
    } else {
      int i2 = idx - n_sh;
      int lane = i2 & 63;
      int ks = (i2 >> 6) % KS_GB;
      int t = (i2 >> 6) / KS_GB;
      int row = lane & 15;
      int ch = t * 8 + (row & 7);
      const float* src = (row < 8) ? w_g : w_b;
      int k0 = ks * 32 + (lane >> 4) * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int k = k0 + e;
        int tap = k / HID, hc = k - tap * HID;
        float v = (ch < c) ? src[((size_t)ch * HID + hc) * 9 + tap] : 0.f;
        o[e] = bits_of<T>(v);
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
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HID; i += gridDim.x * blockDim.x) p_bsh[i] = b_sh[i];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nt * 16; i += gridDim.x * blockDim.x) {
    int t = i / 16, row = i % 16;
    int ch = t * 8 + (row & 7);
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

template <typename T, int NCT, bool KSH1>
int launch(const SpadeParams& p0, hipStream_t s) {
  SpadeParams p = p0;
  p.tiles_y = ceil_div(p.h, TH);
  p.tiles_x = ceil_div(p.w, TW);
  p.nct = NCT;
  const int tiles = p.n * p.tiles_y * p.tiles_x;
  const int chunks = ceil_div(p.nt, NCT);
  size_t smem = (size_t)ACTV_BYTES + 2 * NCT * 2 * 1024 + align16((size_t)CTH * CTW * p.cond_cs * 2) +
                (size_t)p.ksh * 32 * 4 + (size_t)NCT * 32 * 4 + (KSH1 ? 8 * 1024 : 0);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spade_fused_kernel<T, NCT, KSH1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("spade_fused_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((spade_fused_kernel<T, NCT, KSH1>), dim3(tiles, chunks), dim3(WAVES * 64), smem, s, p);
  return CGAN_OK;
}

template <typename T, bool KSH1>
int launch_nct(const SpadeParams& p, int nct, hipStream_t s) {
  switch (nct) {
    case 1: return launch<T, 1, KSH1>(p, s);
    case 2: return launch<T, 2, KSH1>(p, s);
    case 3: return launch<T, 3, KSH1>(p, s);
    case 4: return launch<T, 4, KSH1>(p, s);
    default: return launch<T, 5, KSH1>(p, s);
  }
}

// Development knobs (not part of the stable ABI): force the channel tiles per workgroup / ablation bits.
int g_spade_variant = 0;
int g_spade_dbg = 0;
unsigned long long* g_spade_tsbuf = nullptr;

}  // namespace

// Development/benchmark knob (not part of the stable ABI): choose the fused-SPADE tile variant.
extern "C" void cgan_debug_set_spade_variant(int v) { g_spade_variant = v; }
extern "C" void cgan_debug_set_spade_ablation(int bits) { g_spade_dbg = bits; }
extern "C" void cgan_debug_set_spade_tsbuf(void* p) { g_spade_tsbuf = (unsigned long long*)p; }

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
  const int cs = cgan_cs(d->c), nt = cs / 8, ksh = ceil_div(9 * d->cond_c + 1, 32);
  PackedLayout L = packed_layout(cs, d->cond_c);
  unsigned char* base = (unsigned char*)packed;
  const int total = 8 * ksh * 64 + nt * KS_GB * 64;
  const int blocks = ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048;
  hipStream_t s = (hipStream_t)stream;
  if (d->dtype == CGAN_F16)
    hipLaunchKernelGGL(spade_pack_kernel<F16>, dim3(blocks), dim3(256), 0, s, w_shared, b_shared, w_gamma, b_gamma,
                       w_beta, b_beta, (uint16_t*)(base + L.w_sh), (float*)(base + L.b_sh),
                       (uint16_t*)(base + L.w_gb), (float*)(base + L.b_gb), d->c, nt, d->cond_c, ksh);
  else
    hipLaunchKernelGGL(spade_pack_kernel<BF16>, dim3(blocks), dim3(256), 0, s, w_shared, b_shared, w_gamma, b_gamma,
                       w_beta, b_beta, (uint16_t*)(base + L.w_sh), (float*)(base + L.b_sh),
                       (uint16_t*)(base + L.w_gb), (float*)(base + L.b_gb), d->c, nt, d->cond_c, ksh);
  CGAN_CHECK_LAUNCH("spade_pack_weights");
  return CGAN_OK;
}

extern "C" int cgan_spade_fused_fwd(const void* x, const float* mean, const float* rstd, const void* cond,
                                    const void* packed, void* y, const CganSpadeDesc* d, void* stream) {
  int rc = check(d);
  if (rc != CGAN_OK) return rc;
  CGAN_REQUIRE(x && mean && rstd && cond && packed && y, "spade_fused_fwd: null pointer");
  SpadeParams p;
  const int cs = cgan_cs(d->c);
  PackedLayout L = packed_layout(cs, d->cond_c);
  const unsigned char* base = (const unsigned char*)packed;
  p.x = (const uint16_t*)x; p.mean = mean; p.rstd = rstd; p.cond = (const uint16_t*)cond;
  p.w_sh = (const u32x4*)(base + L.w_sh); p.b_sh = (const float*)(base + L.b_sh);
  p.w_gb = (const u32x4*)(base + L.w_gb); p.b_gb = (const float*)(base + L.b_gb);
  p.y = (uint16_t*)y;
  p.n = d->n; p.h = d->h; p.w = d->w; p.c = d->c; p.cs = cs; p.nt = cs / 8;
  p.x_ups = d->x_upsample; p.hx = d->x_upsample ? d->h / 2 : d->h; p.wx = d->x_upsample ? d->w / 2 : d->w;
  p.cond_h = d->cond_h; p.cond_w = d->cond_w; p.cond_c = d->cond_c; p.cond_cs = cgan_cond_cs(d->cond_c);
  p.ksh = ceil_div(9 * d->cond_c + 1, 32);
  p.sy = (float)d->cond_h / (float)d->h; p.sx = (float)d->cond_w / (float)d->w;
  p.act = d->act; p.slope = d->act_slope; p.dbg = g_spade_dbg; p.tsbuf = g_spade_tsbuf;
  hipStream_t s = (hipStream_t)stream;
  // channel tiles per workgroup: as many as divide nt with the least padded work (nt = 3 -> 3, 5/10/20.. -> 5)
  int nct = p.nt < MAX_NCT ? p.nt : MAX_NCT;
  if (p.nt > MAX_NCT) {
    int best = MAX_NCT, waste = ceil_div(p.nt, MAX_NCT) * MAX_NCT - p.nt;
    for (int k = MAX_NCT - 1; k >= 3; --k) {
      int w = ceil_div(p.nt, k) * k - p.nt;
      if (w < waste) { waste = w; best = k; }
    }
    nct = best;
  }
  if (g_spade_variant >= 1 && g_spade_variant <= MAX_NCT) nct = g_spade_variant;
  const bool ksh1 = p.ksh == 1;
  if (d->dtype == CGAN_F16) rc = ksh1 ? launch_nct<F16, true>(p, nct, s) : launch_nct<F16, false>(p, nct, s);
  else rc = ksh1 ? launch_nct<BF16, true>(p, nct, s) : launch_nct<BF16, false>(p, nct, s);
  if (rc != CGAN_OK) return rc;
  CGAN_CHECK_LAUNCH("spade_fused_fwd");
  return CGAN_OK;
}
