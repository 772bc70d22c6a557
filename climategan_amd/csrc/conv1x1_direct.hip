// 1x1 / stride-1 convolutions with <= 256 output channels (the ResNet bottlenecks' reduce layers 1024 -> 256, 512 -> 128,
// 256 -> 64 and their data gradients' twins, ASPP's 2048 -> 256 ...): a GEMM  Y[pixel][cout] = X[pixel][k] W[cout][k]^T
// whose streamed operand is the ACTIVATION matrix (105 MB for 1024 channels at 8 x 80 x 80) and whose weights are small.
//
// conv_gemm.hip moves both operands through LDS by LDS-DMA and re-fetches each once per block of the other dimension;
// measured (round 3, rocprofv3 + launch events) that fill path delivers ~30-40 GB/s per CU whatever the tile, so these
// layers ran at 2.0-2.4 TB/s of algorithmic traffic (1024 -> 256: 55 us against a 21 us HBM bound).  Here:
//   * a workgroup (8 waves) owns 256 pixels and ALL output channels: X is read exactly once, by the wave that uses it,
//     straight into MFMA B-fragment registers (lane (j, g) of a 16-pixel tile loads the 16 bytes "channels 32 ks + 8 g .."
//     of pixel j: no LDS round trip, no sharing needed -- each pixel belongs to one wave);
//   * only the weights go through LDS (they are shared by the eight waves): K = 64 per stage, register-staged
//     (global_load -> ds_write_b128, one stage ahead) into a double buffer; the packed fragment order makes every piece a
//     contiguous 1-KiB read;
//   * per stage a wave issues its 4 B-fragment loads of the NEXT stage and its weight loads of the stage after next, then
//     runs 2 k-steps x CT cout tiles x 2 pixel tiles MFMAs; one barrier per stage.  Plain loads only (no LDS-DMA), so the
//     compiler's own counted s_waitcnt keeps every load in flight until its first use.
// Epilogue as conv_gemm_kernel (fp32 staging through LDS, 16-byte stores, bias / residual / activation in fp32).
#include "conv_gemm.h"

namespace {

// CTW cout tiles per wave x WAVES_C wave rows = all couts; 8 / WAVES_C wave columns x WP pixel tiles x 16 = pixels per
// workgroup.  <8, 2, 4>: 256 couts (a wave: 128 couts x 64 pixels, one A fragment read per FOUR MFMAs; every pixel's B
// fragments are loaded by the two waves of its column -- the second load is an L1 / L2 hit); <8, 1, 2>: 128 couts;
// <4, 1, 2>: 64 couts.  (The first version gave every wave all 16 cout tiles x 2 pixel tiles: one LDS read per two MFMAs at
// 256 registers, no room to prefetch the A fragments -- LDS latency bound, 1024 -> 256 at 8 x 80^2 68 us against 55.)
template <typename T, int CTW, int WAVES_C, int WP>
__global__ __launch_bounds__(512, 2) void conv1x1_allc_kernel(ConvGemmArgs p) {
  constexpr int NWAVE = 8, CT = CTW * WAVES_C, NPG = NWAVE / WAVES_C;
  constexpr int PIECES = CT * 2;                                   // 1-KiB weight pieces per K = 64 stage
  constexpr int PPW = (PIECES + NWAVE - 1) / NWAVE;                // per wave
  constexpr int STAGE_BYTES = PIECES * 1024;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j16 = lane & 15, g = lane >> 4;
  const int wc = wave % WAVES_C, wpg = wave / WAVES_C;
  const int pix0 = (blockIdx.x * NPG + wpg) * (WP * 16);
  const int n_st = p.cin_s >> 6;

  // B fragments straight from global memory: pixel (tile t, j), channels 64 s + 32 h + 8 g ..
  const u32x4* bsrc[WP];
#pragma unroll
  for (int t = 0; t < WP; ++t) {
    const int pix = min(pix0 + t * 16 + j16, p.npix - 1);          // rows past the end are computed and never stored
    bsrc[t] = reinterpret_cast<const u32x4*>(p.x + (size_t)pix * p.cin_s + g * 8);
  }
  auto load_b = [&](u32x4 (*b)[WP], int s) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < WP; ++t) b[h][t] = bsrc[t][s * 8 + h * 4];    // + (64 s + 32 h) channels = (8 s + 4 h) x 16 B
  };
  // weight pieces of this wave: id = wave + NWAVE m -> (cout tile id >> 1, k-half id & 1)
  const u32x4* wsrc[PPW];
#pragma unroll
  for (int m = 0; m < PPW; ++m) {
    const int id = wave + NWAVE * m;
    wsrc[m] = p.w + ((size_t)min(id >> 1, p.ctiles - 1) * p.ksteps + (id & 1)) * 64 + lane;
  }
  u32x4 wreg[PPW];
  auto load_w = [&](int s) {
#pragma unroll
    for (int m = 0; m < PPW; ++m)
      if (wave + NWAVE * m < PIECES) wreg[m] = wsrc[m][(size_t)s * 128];
  };
  auto store_w = [&](int slot) {
#pragma unroll
    for (int m = 0; m < PPW; ++m)
      if (wave + NWAVE * m < PIECES)
        *reinterpret_cast<u32x4*>(smem + slot * STAGE_BYTES + (wave + NWAVE * m) * 1024 + lane * 16) = wreg[m];
  };

  f32x4 acc[CTW][WP];
#pragma unroll
  for (int c = 0; c < CTW; ++c)
#pragma unroll
    for (int t = 0; t < WP; ++t) acc[c][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  u32x4 b0[2][WP], b1[2][WP];
  // ---- prologue: W(0) in slot 0, W(1) on its way, B(0) on its way
  load_w(0);
  load_b(b0, 0);
  store_w(0);
  if (n_st > 1) load_w(1);
  __syncthreads();

  auto stage = [&](int s, u32x4 (*bc)[WP], u32x4 (*bn)[WP]) {
    const int slot = s & 1;
    if (s + 1 < n_st) {
      store_w(slot ^ 1);                       // W(s + 1): loaded during stage s - 1; the slot was read in stage s - 1
      load_b(bn, s + 1);
    }
    if (s + 2 < n_st) load_w(s + 2);
    const unsigned char* wl = smem + slot * STAGE_BYTES + wc * CTW * 2048 + lane * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(wl + (c * 2 + h) * 1024);
#pragma unroll
        for (int t = 0; t < WP; ++t) acc[c][t] = mfma16(as_vec8<T>(a), as_vec8<T>(bc[h][t]), acc[c][t]);
      }
    __syncthreads();                           // W(s + 1) visible; everyone is done with slot s & 1
  };
  for (int s = 0; s < n_st; s += 2) {
    stage(s, b0, b1);
    stage(s + 1, b1, b0);                      // n_st is even (cin_s % 128 == 0)
  }

  // ---- epilogue: one pixel tile at a time through LDS (fp32 rows of CT*16 couts + 16 B pad), 16-byte stores
  constexpr int ROWB = CTW * 64 + 16;
  unsigned char* stg = smem + wave * (16 * ROWB);
  const int cout_base = wc * CTW * 16;
  // (the store path shared with conv_gemm_kernel, one pixel tile per pass: conv_gemm.h)
  conv_gemm_staged_store<T, CTW, WP, 1>(acc, p, stg, pix0, cout_base, p.bias, lane, j16, g);
}

template <typename T, int CTW, int WAVES_C, int WP>
int launch_allc(const ConvGemmArgs& a, hipStream_t s) {
  constexpr int CT = CTW * WAVES_C, PB = (8 / WAVES_C) * WP * 16;
  constexpr size_t stage2 = (size_t)2 * CT * 2 * 1024, epi = (size_t)8 * 16 * (CTW * 64 + 16);
  constexpr size_t smem = stage2 > epi ? stage2 : epi;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1x1_allc_kernel<T, CTW, WAVES_C, WP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) {
      cgan_set_error("conv1x1_direct: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return CGAN_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((conv1x1_allc_kernel<T, CTW, WAVES_C, WP>), dim3(ceil_div(a.npix, PB)), dim3(512), smem, s, a);
  return CGAN_OK;
}

template <typename T>
int launch_any(const ConvGemmArgs& a, hipStream_t s) {
  if (a.ctiles <= 4) return launch_allc<T, 4, 1, 2>(a, s);
  if (a.ctiles <= 8) return launch_allc<T, 8, 1, 2>(a, s);
  return launch_allc<T, 8, 2, 4>(a, s);
}

}  // namespace

// 1x1, stride 1, no padding, whole 128-channel chunks (the stage loop is unrolled by two K = 64 stages), <= 256 couts
bool conv1x1_allc_ok(const ConvGemmArgs& a) {
  return a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad == 0 && (a.cin_s & 127) == 0 && a.ctiles <= 16 &&
         a.h_in == a.h_out && a.w_in == a.w_out;
}

int conv1x1_allc_launch(const ConvGemmArgs& a, int dtype, hipStream_t s) {
  return dtype == CGAN_F16 ? launch_any<F16>(a, s) : launch_any<BF16>(a, s);
}
