// 1x1 / stride-1 convolutions with a SHORT K (cin <= 256) and many output channels (the ResNet bottlenecks' expand layers
// 256 -> 1024, 128 -> 512, 64 -> 256 and the data gradients of the reduce layers): the activation tile stays RESIDENT in
// LDS while the workgroup walks ALL output channels (round 3).
//
// Why.  With a (BC couts x BP pixels) block tile both operands are re-fetched through the L2 -> LDS path once per block of
// the other dimension.  256 -> 1024 at 8 x 80 x 80 (26 MB in, 105 MB out, HBM bound 16 us, MFMA bound 11 us) moves 415 MB
// through that path with the 128 x 128 tile of conv_gemm.hip and runs 70 us: K is only 8 k-steps, so a workgroup is all
// prologue (first fill from HBM) and epilogue, and the fill is 3 x the HBM traffic.  Here
//   * a workgroup (8 waves, one per CU) owns 256 pixels: their cin_s channels (<= 128 KiB) are fetched ONCE by LDS-DMA, in
//     the conflict-free image of conv_gemm_big.hip's pixel stage (8 pixels x 128 B per 1-KiB piece: whole cache lines);
//   * the weights never touch LDS: they are packed in MFMA fragment order, so the A fragment of (cout tile, k-step) is one
//     contiguous 1-KiB global_load_dwordx4 per wave, straight into registers, prefetched PF k-steps ahead (L2 hits: the
//     whole weight matrix is <= 512 KiB).  Waves are 4 (cout) x 2 (pixel): 64 couts x 128 pixels each per 256-cout block,
//     one B-fragment LDS read per four MFMAs, A fragments shared by two waves only (2 x redundancy on the smallest operand);
//   * after the one barrier that publishes the activation tile the waves never synchronise again: each walks the cout
//     blocks on its own, and the stores of one wave's epilogue overlap the MFMAs of the others;
//   * epilogue per (cout block, pixel tile): rounded in registers, transposed through a wave-private 2.3-KiB staging row
//     block, stored as 16-byte chunks = one full 128-byte line per pixel and wave.
//     Optional BatchNorm statistics from the fp32 accumulators as in conv_gemm_kernel (chunk = a wave's 128 pixels).
// L2 -> CU traffic of the example: 26 MB of activations + 200 workgroups x 2 x 512 KiB of weights = 236 MB (was 415).
#include "conv_gemm.h"
#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __attribute__((aligned(16))) unsigned int g_xres_zeros[4];

__device__ __forceinline__ float row16_sum_x(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// KC = 64-channel chunks of K (cin_s = 64 KC); PF = k-steps (32 channels) of A fragments in flight ahead of the MFMAs
template <typename T, int KC, int PF>
__global__ __launch_bounds__(512, 1) void conv1x1_xres_kernel(ConvGemmArgs p, int npb, unsigned long long* ts) {
  constexpr int WC = 4, WP = 8;                   // per wave: 64 couts x 128 pixels
  constexpr int NKS = 2 * KC;                     // k-steps
  constexpr int XCHUNK = 32 * 1024;               // 256 pixels x 64 channels
  constexpr int ROWB = WC * 32 + 16;              // staging row: 64 couts x 2 B + pad
  constexpr int STG = 16 * ROWB;                  // one pixel tile
  static_assert(NKS % PF == 0, "the prefetch ring must divide the k-steps (static register indices)");
  static_assert(KC * XCHUNK + 8 * STG <= 160 * 1024, "LDS");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int pblk = xcd * ((npb + 7) >> 3) + slot;
  if (pblk >= npb) return;
  // development aid (tools/ts_xres.py): per-wave cycle stamps, 16 slots per wave; null in production
  auto stamp = [&](int i) {
    if (ts && lane == 0 && i < 16) ts[((size_t)blockIdx.x * 8 + wave) * 16 + i] = __builtin_readcyclecounter();
  };
  stamp(0);

  // ---- the activation tile: KC x 32 pieces of 1 KiB, piece id -> (chunk id >> 5, pixel tile (id & 31) >> 1, half id & 1)
  {
    const long zero_off = reinterpret_cast<const unsigned char*>(g_xres_zeros) - reinterpret_cast<const unsigned char*>(p.x);
    const int q = lane >> 3;
#pragma unroll
    for (int m = 0; m < KC * 4; ++m) {
      const int id = wave + 8 * m;
      const int cc2 = id >> 5, t = (id & 31) >> 1, half = id & 1;
      const int kchunk = (lane & 7) ^ (4 * half + ((q >> 1) & 3));
      const int pix = (pblk * 16 + t) * 16 + 8 * half + q;
      const long off = pix < p.npix ? ((long)pix * p.cin_s + cc2 * 64 + kchunk * 8) * 2 : zero_off;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(p.x) + off),
          (__attribute__((address_space(3))) void*)(smem + id * 1024), 16, 0, 0);
    }
  }

  const int wc = wave & 3, wp = wave >> 2;
  const int j16 = lane & 15, g = lane >> 4;
  const int ncb = (p.ctiles + 15) >> 4;
  int b_off[2];
  {
    const int q = j16 & 7, h = j16 >> 3;
#pragma unroll
    for (int c = 0; c < 2; ++c) b_off[c] = wp * WP * 2048 + 1024 * h + 128 * q + 16 * ((4 * c + g) ^ (4 * h + ((q >> 1) & 3)));
  }
  // A fragments: packed [cout tile][k-step][lane]; this wave's tiles of cout block cb: cb * 16 + wc * 4 + c.  Buffer loads:
  // one VGPR (16 * lane) for every weight address of the kernel, the rest is wave-uniform (SGPR offset)
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.w), 0, p.ctiles * p.ksteps * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.npix * p.cout_s * 2, 0x00020000);
  const int lane16 = lane * 16;
  u32x4 a[PF][WC];
  auto load_a = [&](auto slot_tag, int cb, int ks) {
    constexpr int sl = decltype(slot_tag)::value;
#pragma unroll
    for (int c = 0; c < WC; ++c) {
      const int ct = min(cb * 16 + wc * WC + c, p.ctiles - 1);
      a[sl][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane16, (ct * p.ksteps + ks) * 1024, 0);
    }
  };
  // the first PF k-steps' weights are on their way while the activations land
  static_for<0, PF>([&](auto i) { load_a(i, 0, decltype(i)::value); });

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                   // the tile is complete and visible to every wave
  stamp(1);

  unsigned char* stg = smem + KC * XCHUNK + wave * STG;
  const int st_rd = (lane >> 3) * ROWB + (lane & 7) * 16;              // staged row / 16-byte chunk this lane stores
  const int st_voff = ((lane >> 3) * p.cout_s + (lane & 7) * 8) * 2;
  const int pix_wave = (pblk * 16 + wp * WP) * 16;

  // B fragments: two half sets (pixel tiles 0-3 / 4-7 of the wave), each read from LDS half a k-step before its MFMAs, into
  // the registers the half before last has just released.
  u32x4 b_lo[4], b_hi[4];
  auto read_b = [&](u32x4* bh, int ks, int half) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      bh[t] = *reinterpret_cast<const u32x4*>(smem + (ks >> 1) * XCHUNK + b_off[ks & 1] + (half * 4 + t) * 2048);   // (ks is static)
  };

  for (int cb = 0; cb < ncb; ++cb) {
    read_b(b_lo, 0, 0);                            // (kept across the epilogue it was spilled)
    f32x4 acc[WC][WP];
#pragma unroll
    for (int c = 0; c < WC; ++c)
#pragma unroll
      for (int t = 0; t < WP; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto kstep = [&](auto ks_tag) {
      constexpr int ks = decltype(ks_tag)::value;
      constexpr int sl = ks % PF;
      read_b(b_hi, ks, 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < WC; ++c) acc[c][t] = mfma16(as_vec8<T>(a[sl][c]), as_vec8<T>(b_lo[t]), acc[c][t]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ks + 1 < NKS) read_b(b_lo, ks + 1, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < WC; ++c) acc[c][4 + t] = mfma16(as_vec8<T>(a[sl][c]), as_vec8<T>(b_hi[t]), acc[c][4 + t]);
      __builtin_amdgcn_sched_barrier(0);
      // refill this slot with the k-step PF ahead; the next cout block's first PF k-steps are requested from the middle of
      // the epilogue (half of the accumulators are dead by then: prefetched across the whole epilogue they were spilled)
      if constexpr (ks + PF < NKS) load_a(std::integral_constant<int, sl>{}, cb, ks + PF);
      __builtin_amdgcn_sched_barrier(0);
    };
    static_for<0, NKS>(kstep);
    stamp(2 + 2 * cb);

    // ---- epilogue of this cout block: lane holds couts ct * 16 + 4 g + {0..3} of pixel (tile t, j16).  The kernel takes
    // only convs without bias / activation / pad channels (the bottleneck convs, whose BatchNorm follows, and the data
    // gradients): convert and store.  FULL (wave-uniform: no pixel / cout tail in this wave's tile) stores unguarded.
    const int cout_base = (cb * 16 + wc * WC) * 16;
    const bool full = pix_wave + WP * 16 <= p.npix && cout_base + WC * 16 <= p.cout_s;
    if (p.stats && pix_wave < p.npix) {     // (a trailing wave of the last block owns no chunk: npix % 128 == 0 only)
      // (mean, M2) of this wave's 128 pixels per channel; pairs of channels on the packed-fp32 VALU path
      const int chunk = pblk * 2 + wp;
      constexpr float inv_cnt = 1.f / (float)(WP * 16);
#pragma unroll
      for (int c = 0; c < WC; ++c) {
        // of the values as stored (rounded), chunk mean first, then M2 = sum (v - mean)^2: see conv_gemm_kernel
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
        float mean[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) mean[r] = row16_sum_x(s0[r >> 1][r & 1]) * inv_cnt;
        const f32x2 m2[2] = {(f32x2){mean[0], mean[1]}, (f32x2){mean[2], mean[3]}};
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
        float o[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[2 * r] = mean[r];
          o[2 * r + 1] = row16_sum_x(s1[r >> 1][r & 1]);
        }
        const int ch = cout_base + c * 16 + 4 * g;
        if (j16 == 0 && ch < p.cout_s) {
          float* dst = p.stats + ((size_t)chunk * p.cout_s + ch) * 2;
          *reinterpret_cast<f32x4*>(dst) = (f32x4){o[0], o[1], o[2], o[3]};
          *reinterpret_cast<f32x4*>(dst + 4) = (f32x4){o[4], o[5], o[6], o[7]};
        }
      }
    }
    auto emit = [&](auto full_tag) {
      constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
      for (int t = 0; t < WP; ++t) {
#pragma unroll
        for (int c = 0; c < WC; ++c) {
          const f32x4 v = acc[c][t];
          *reinterpret_cast<u32x2*>(stg + j16 * ROWB + c * 32 + g * 8) = (u32x2){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const u32x4 o = *reinterpret_cast<const u32x4*>(stg + st_rd + it * 8 * ROWB);
          const int pix0 = pix_wave + t * 16 + it * 8;                   // wave-uniform: the store's SGPR offset
          if (FULL || (pix0 + (lane >> 3) < p.npix && cout_base + (lane & 7) * 8 < p.cout_s))
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, st_voff, (pix0 * p.cout_s + cout_base) * 2, 2);   // aux 2 = nt (streamed: cgan_common.h)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the staged rows are read before the next tile overwrites
        if (t == WP / 2 - 1 && cb + 1 < ncb) {
          __builtin_amdgcn_sched_barrier(0);
          static_for<0, PF>([&](auto i) { load_a(i, cb + 1, decltype(i)::value); });
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (full) emit(std::true_type{});
    else emit(std::false_type{});
    stamp(3 + 2 * cb);
  }
}

CGAN_KNOB(unsigned long long*, g_xres_ts, nullptr);

template <typename T, int KC, int PF>
int launch_xres(const ConvGemmArgs& a, hipStream_t s) {
  constexpr size_t smem = (size_t)KC * 32 * 1024 + 8 * 16 * (4 * 32 + 16);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_xres_kernel<T, KC, PF>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("conv1x1_xres: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  const int npb = ceil_div(a.npix, 256);
  hipLaunchKernelGGL((conv1x1_xres_kernel<T, KC, PF>), dim3(ceil_div(npb, 8) * 8), dim3(512), smem, s, a, npb, g_xres_ts);
  return CGAN_OK;
}

template <typename T>
int launch_any(const ConvGemmArgs& a, hipStream_t s) {
  switch (a.cin_s >> 6) {
    case 1: return launch_xres<T, 1, 2>(a, s);
    case 2: return launch_xres<T, 2, 2>(a, s);
    case 3: return launch_xres<T, 3, 2>(a, s);
    default: return launch_xres<T, 4, 2>(a, s);
  }
}

}  // namespace

CGAN_DEV_ONLY(extern "C" void cgan_debug_set_xres_tsbuf(void* p) { g_xres_ts = (unsigned long long*)p; })

// 1x1, stride 1, no padding, no residual / bias / activation / pad channels, whole 64-channel chunks up to 256 channels, 32-bit byte offsets
bool conv1x1_xres_ok(const ConvGemmArgs& a) {
  return a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && (a.cin_s & 63) == 0 && a.cin_s <= 256 && !a.has_res &&
         !a.bias && a.act == CGAN_ACT_NONE && a.cout == a.cout_s &&
         a.h_in == a.h_out && a.w_in == a.w_out && (long)a.npix * a.cin_s < (1L << 30) &&
         (long)a.npix * a.cout_s < (1L << 30);
}

int conv1x1_xres_launch(const ConvGemmArgs& a, int dtype, hipStream_t s) {
  return dtype == CGAN_F16 ? launch_any<F16>(a, s) : launch_any<BF16>(a, s);
}
