// Fused SPADE forward for gfx950 (reference climategan/norms.py:146-186):
//
//   seg   = nearest_resize(cond, (h, w))
//   actv  = ReLU(conv3x3(seg, w_shared) + b_shared)                    128 hidden channels
//   gamma = conv3x3(actv, w_gamma) + b_gamma ; beta = conv3x3(actv, w_beta) + b_beta
//   y     = act( (x - mean) * rstd * (1 + gamma) + beta )
//
// One workgroup (4 waves) owns a TH x 16 pixel tile of one image:
//   phase 0  stage the nearest-resized conditioning halo tile (TH+4) x 20 x cond_c in LDS
//   phase 1  hidden map for the (TH+2) x 18 halo: one MFMA implicit GEMM (K = 9*cond_c, padded to 32),
//            bias + ReLU, zeroed outside the image (the gamma/beta convs zero-pad actv, not seg), written
//            to LDS as [halo pixel][128 ch] 16-bit with an XOR swizzle on the 16-byte channel slots
//   phase 2  gamma||beta implicit GEMM straight out of LDS: M = 2*Cs output rows, N = tile pixels,
//            K = 9 taps x 128 = 1152 (36 MFMA k-steps), fp32 accumulate
//   phase 3  epilogue in registers: rows are packed so that a 16-row MFMA tile holds gamma[8t..8t+7] and
//            beta[8t..8t+7]; one v_permlane32_swap pair brings gamma and beta of the same channel into the
//            same lane, which then normalises x and stores 2 channels.
// The 128-channel hidden map (105 MB/img at 640x640 in 16-bit) never touches HBM; HBM traffic is
// x (read) + y (write) + the 3-channel cond halo.
//
// MFMA operand roles: A = weights (rows = output channels), B = activations (cols = pixels), so that
// D's per-lane 4 registers are 4 consecutive channel rows of one pixel (col = lane&15, row = 4*(lane>>4)+r).
#include "cgan_common.h"

namespace {

constexpr int TW = 16;       // pixel-tile width == MFMA N
constexpr int HID = 128;     // hidden channels (norms.py:163)
constexpr int KS_GB = 36;    // 9 taps * 128 / 32
constexpr int CTC = 5;       // channel tiles (16 rows = 8 gamma + 8 beta) accumulated per pass

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
  int nt_per_split;
  int act;
  float slope;
};

__host__ __device__ inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

struct PackedLayout {
  size_t w_sh, b_sh, w_gb, b_gb, total;
};
__host__ inline PackedLayout packed_layout(int cs, int cond_c) {
  PackedLayout L;
  int ksh = ceil_div(9 * cond_c, 32);
  int nt = cs / 8;
  L.w_sh = 0;
  L.b_sh = align16(L.w_sh + (size_t)8 * ksh * 64 * 16);
  L.w_gb = align16(L.b_sh + HID * sizeof(float));
  L.b_gb = align16(L.w_gb + (size_t)nt * KS_GB * 64 * 16);
  L.total = align16(L.b_gb + (size_t)nt * 16 * sizeof(float));
  return L;
}

template <typename T, int TH>
__global__ __launch_bounds__(256) void spade_fused_kernel(SpadeParams p) {
  constexpr int HPH = TH + 2, HPW = TW + 2, HP = HPH * HPW;  // hidden halo
  constexpr int CTH = TH + 4, CTW = TW + 4;                  // cond halo
  constexpr int PT = TH / 4;                                 // pixel tiles (rows) per wave
  constexpr int NHT = (HP + 15) / 16;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* actv = smem;                                             // HP * 256 B
  uint16_t* ctile = reinterpret_cast<uint16_t*>(smem + (size_t)HP * 256); // CTH*CTW*cond_cs
  int* lut = reinterpret_cast<int*>(smem + (size_t)HP * 256 + align16((size_t)CTH * CTW * p.cond_cs * 2));

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15;
  const int g = lane >> 4;

  int tile = blockIdx.x;
  const int txi = tile % p.tiles_x;
  tile /= p.tiles_x;
  const int tyi = tile % p.tiles_y;
  const int n = tile / p.tiles_y;
  const int ty0 = tyi * TH, tx0 = txi * TW;

  // ---------------- phase 0: cond halo tile + K lookup table
  {
    const int groups = p.cond_cs / 4;
    const int total = CTH * CTW * groups;
    for (int i = threadIdx.x; i < total; i += 256) {
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
    const int kmax = 9 * p.cond_c;
    for (int k = threadIdx.x; k < p.ksh * 32; k += 256) {
      int off = -1;
      if (k < kmax) {
        int tap = k / p.cond_c, ch = k - tap * p.cond_c;
        off = ((tap / 3) * CTW + (tap % 3)) * p.cond_cs + ch;
      }
      lut[k] = off;
    }
  }
  __syncthreads();

  // ---------------- phase 1: hidden map into LDS
  for (int ht = wave; ht < NHT; ht += 4) {
    const int q = ht * 16 + j;           // hidden halo pixel of this lane (as MFMA column)
    const bool qv = q < HP;
    const int qy = qv ? q / HPW : 0, qx = qv ? q % HPW : 0;
    const uint16_t* cbase = ctile + (size_t)(qy * CTW + qx) * p.cond_cs;
    f32x4 acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < p.ksh; ++ks) {
      uint16_t e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int off = lut[ks * 32 + g * 8 + i];
        e[i] = (off >= 0 && qv) ? cbase[off] : (uint16_t)0;
      }
      u32x4 b;
      b[0] = e[0] | ((uint32_t)e[1] << 16);
      b[1] = e[2] | ((uint32_t)e[3] << 16);
      b[2] = e[4] | ((uint32_t)e[5] << 16);
      b[3] = e[6] | ((uint32_t)e[7] << 16);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        u32x4 a = p.w_sh[((size_t)c * p.ksh + ks) * 64 + lane];
        acc[c] = mfma16(as_vec8<T>(a), as_vec8<T>(b), acc[c]);
      }
    }
    if (qv) {
      const int yy = ty0 - 1 + qy, xx = tx0 - 1 + qx;
      const bool inside = yy >= 0 && yy < p.h && xx >= 0 && xx < p.w;
      const int key = q & 15;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int ch = c * 16 + g * 4;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = acc[c][r] + p.b_sh[ch + r];
          v[r] = (inside && t > 0.f) ? t : 0.f;
        }
        u32x2 o;
        o[0] = pack2<T>(v[0], v[1]);
        o[1] = pack2<T>(v[2], v[3]);
        const int slot = (c * 2 + (g >> 1)) ^ key;
        *reinterpret_cast<u32x2*>(actv + (size_t)q * 256 + slot * 16 + (g & 1) * 8) = o;
      }
    }
  }
  __syncthreads();

  // ---------------- phase 2/3: gamma||beta GEMM out of LDS + SPADE epilogue
  const int nt_begin = blockIdx.y * p.nt_per_split;
  const int nt_end = min(p.nt, nt_begin + p.nt_per_split);

  // per-lane pixel (column j of pixel-tile row pr) bookkeeping
  int prow[PT];
#pragma unroll
  for (int t = 0; t < PT; ++t) prow[t] = wave * PT + t;

  for (int nt0 = nt_begin; nt0 < nt_end; nt0 += CTC) {
    f32x4 acc[CTC][PT];
#pragma unroll
    for (int c = 0; c < CTC; ++c)
#pragma unroll
      for (int t = 0; t < PT; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const int ks = tap * 4 + kc;
        u32x4 a[CTC];
#pragma unroll
        for (int c = 0; c < CTC; ++c) {
          a[c] = (u32x4){0u, 0u, 0u, 0u};
          if (nt0 + c < nt_end) a[c] = p.w_gb[((size_t)(nt0 + c) * KS_GB + ks) * 64 + lane];
        }
        u32x4 b[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) {
          const int q = (prow[t] + dy) * HPW + (j + dx);
          const int slot = (kc * 4 + g) ^ (q & 15);
          b[t] = *reinterpret_cast<const u32x4*>(actv + (size_t)q * 256 + slot * 16);
        }
#pragma unroll
        for (int c = 0; c < CTC; ++c) {
          if (nt0 + c < nt_end) {
#pragma unroll
            for (int t = 0; t < PT; ++t) acc[c][t] = mfma16(as_vec8<T>(a[c]), as_vec8<T>(b[t]), acc[c][t]);
          }
        }
      }
    }

    // ---- epilogue.  Rows of tile nt: 0-7 = (1+gamma)[8nt..], 8-15 = beta[8nt..].
    // lanes 0-31 (g=0,1) hold gamma rows 4g+r, lanes 32-63 (g=2,3) beta rows 4(g-2)+r of the same pixel j.
    // swap(a0,a2), swap(a1,a3):  lo lanes -> (gamma,beta) of channels 4g+{0,1};  hi lanes -> 4(g-2)+{2,3}.
    const int chan_in_tile = (g & 1) * 4 + (g >> 1) * 2;
#pragma unroll
    for (int c = 0; c < CTC; ++c) {
      const int nt = nt0 + c;
      if (nt >= nt_end) continue;
      const float* bias = p.b_gb + (size_t)nt * 16 + g * 4;
      const float b0 = bias[0], b1 = bias[1], b2 = bias[2], b3 = bias[3];
      const int ch = nt * 8 + chan_in_tile;
      const float m0 = p.mean[(size_t)n * p.cs + ch], m1 = p.mean[(size_t)n * p.cs + ch + 1];
      const float r0 = p.rstd[(size_t)n * p.cs + ch], r1 = p.rstd[(size_t)n * p.cs + ch + 1];
#pragma unroll
      for (int t = 0; t < PT; ++t) {
        unsigned a0 = __builtin_bit_cast(unsigned, acc[c][t][0] + b0);
        unsigned a1 = __builtin_bit_cast(unsigned, acc[c][t][1] + b1);
        unsigned a2 = __builtin_bit_cast(unsigned, acc[c][t][2] + b2);
        unsigned a3 = __builtin_bit_cast(unsigned, acc[c][t][3] + b3);
        auto s02 = __builtin_amdgcn_permlane32_swap(a0, a2, false, false);
        auto s13 = __builtin_amdgcn_permlane32_swap(a1, a3, false, false);
        const float gm0 = __builtin_bit_cast(float, (unsigned)s02[0]);
        const float bt0 = __builtin_bit_cast(float, (unsigned)s02[1]);
        const float gm1 = __builtin_bit_cast(float, (unsigned)s13[0]);
        const float bt1 = __builtin_bit_cast(float, (unsigned)s13[1]);
        const int yy = ty0 + prow[t], xx = tx0 + j;
        if (yy < p.h && xx < p.w) {
          const int sy = p.x_ups ? (yy >> 1) : yy, sx = p.x_ups ? (xx >> 1) : xx;
          const uint32_t xv = *reinterpret_cast<const uint32_t*>(
              p.x + (((size_t)n * p.hx + sy) * p.wx + sx) * p.cs + ch);
          float x0, x1;
          unpack2<T>(xv, x0, x1);
          float o0 = (x0 - m0) * r0 * gm0 + bt0;
          float o1 = (x1 - m1) * r1 * gm1 + bt1;
          o0 = act_apply(o0, p.act, p.slope);
          o1 = act_apply(o1, p.act, p.slope);
          if (ch >= p.c) o0 = 0.f;
          if (ch + 1 >= p.c) o1 = 0.f;
          *reinterpret_cast<uint32_t*>(p.y + (((size_t)n * p.h + yy) * p.w + xx) * p.cs + ch) = pack2<T>(o0, o1);
        }
      }
    }
  }
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

template <typename T, int TH>
int launch(const SpadeParams& p0, hipStream_t s) {
  SpadeParams p = p0;
  p.tiles_y = ceil_div(p.h, TH);
  p.tiles_x = ceil_div(p.w, TW);
  const int tiles = p.n * p.tiles_y * p.tiles_x;
  // split the channel tiles over blockIdx.y when the pixel tiles alone cannot fill the chip
  int chunks = ceil_div(p.nt, CTC);
  int split = 1;
  while (split < chunks && tiles * split < 512) ++split;
  p.nt_per_split = ceil_div(chunks, split) * CTC;
  split = ceil_div(p.nt, p.nt_per_split);
  constexpr int HP = (TH + 2) * (TW + 2);
  size_t smem = (size_t)HP * 256 + align16((size_t)(TH + 4) * (TW + 4) * p.cond_cs * 2) + (size_t)p.ksh * 32 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spade_fused_kernel<T, TH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("spade_fused_fwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((spade_fused_kernel<T, TH>), dim3(tiles, split), dim3(256), smem, s, p);
  return CGAN_OK;
}

}  // namespace

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
  const int cs = cgan_cs(d->c), nt = cs / 8, ksh = ceil_div(9 * d->cond_c, 32);
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
  p.ksh = ceil_div(9 * d->cond_c, 32);
  p.sy = (float)d->cond_h / (float)d->h; p.sx = (float)d->cond_w / (float)d->w;
  p.act = d->act; p.slope = d->act_slope;
  hipStream_t s = (hipStream_t)stream;
  // small images: 8-row tiles (more workgroups, less padding waste); otherwise 16-row tiles
  const bool small = d->h <= 8 || ((long)d->n * ceil_div(d->h, 16) * ceil_div(d->w, 16) < 256);
  if (d->dtype == CGAN_F16) rc = small ? launch<F16, 8>(p, s) : launch<F16, 16>(p, s);
  else rc = small ? launch<BF16, 8>(p, s) : launch<BF16, 16>(p, s);
  if (rc != CGAN_OK) return rc;
  CGAN_CHECK_LAUNCH("spade_fused_fwd");
  return CGAN_OK;
}
