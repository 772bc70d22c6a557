// conv2d backward-weight (NHWC 16-bit activations, fp32 OIHW gradient), MFMA 16x16x32, gfx950.
//
//   dW[co][ci][ky][kx] += sum over output pixels  dY[pix][co] * X[pix shifted by tap (ky,kx)][ci]
//
// GEMM view per tap: M = co, N = ci, K = output pixels.  Both operands are K-strided in NHWC memory (a lane needs 8
// PIXELS of one channel), so tiles are staged in LDS row-major [pixel][64 channels] by LDS-DMA (8 pixels x 128 B per
// 1-KiB piece: whole cache lines) and read with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group
// reads a [4 pixels][16 channels] block and each lane receives one channel's 4 pixels) -- two of them make one
// MFMA operand fragment, no shuffles.
//
// Workgroup = 4 independent waves: wave w owns pixels [32w, 32w+32) of every 128-pixel chunk (its own DMA, its own
// double buffer, no barriers in the loop) and accumulates the full 64 co x 64 ci tile of one tap; the four partial
// tiles are summed through LDS at the end and added to dW with fp32 atomics (grid = pixel splits x taps x channel
// blocks, so several workgroups contribute to each weight).
#include "cgan_common.h"
#include <type_traits>

namespace {


struct WgradArgs {
  const uint16_t* x;
  const uint16_t* dy;
  float* dw;
  unsigned x_bytes, dy_bytes;   // extents for the buffer descriptors (out-of-range DMA lanes read zeros)
  float* ws;   // non-null: partial tiles go to ws[split][tile][16 fragments][64 lanes] (f32x4) for wgrad_reduce_kernel
  float* bpart;  // non-null: the bias gradient rides along -- the workgroups of N tile 0 multiply their dy fragments with a
                 // constant-one B fragment as well (sum over pixels = one more GEMM column) and store bpart[split][cout_s]
  int n, h_in, w_in, cin, cin_s;
  int cout, cout_s;
  int kh, kw, stride, pad, dil;
  int h_out, w_out, npix;
  int nchunks, ci_blocks, co_blocks, splits, per_xcd;
  int coop, co_pairs, n_pairs;     // cooperative 128 x 128 kernel: pairs of co blocks x pairs of N tiles
  int fold, cpt, tpt, tap_slots;   // tap folding for cin_s <= 32 (see wgrad_plan): 16-byte chunks per tap, taps per 64-wide tile, tap groups
  CGAN_DEV_ONLY(int dbg;)     // dev build: ablation bits (tools/bench_wgrad.py): 1 = skip the atomics, 2 = skip the MFMAs
  int reflect; // 1: nn.ReflectionPad2d(pad) in front of the conv (index math instead of the zero page)
  int x_ups;   // 1: x is stored at (h_in/2, w_in/2) and read through the folded nearest x2 upsample
  CGAN_DEV_ONLY(unsigned long long* ts;)   // dev build (tools/ts_wgrad.py): per-wave phase tick sums of the cooperative kernel
};
#ifdef CGAN_DEV
#define CGAN_WTS(p) ((p).ts)
#else
#define CGAN_WTS(p) ((unsigned long long*)nullptr)
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));
// LDS-DMA of 16 bytes per lane through a buffer descriptor, HIDDEN from the compiler (round 6).  hipcc (ROCm 7.2) treats
// __builtin_amdgcn_raw_ptr_buffer_load_lds as a store to LDS that any later ds_read may alias and puts ``s_waitcnt vmcnt(0)``
// in front of the first fragment read after it -- i.e. behind the pieces of the NEXT chunk that were issued a moment before:
// every chunk waited for its successor's L2 / HBM round trip before multiplying (the disassembly showed the wait in all
// three kernels of this file; the SQ counters showed the waves parked 0.52-0.54 of their life).  As an asm statement the
// load is outside the compiler's counters: the kernels' own counted waits (already there) are the only ones.  M0 (the LDS
// destination) is saved and restored around the load; the s_nop covers the M0 write -> LDS-DMA hazard.
typedef unsigned wg_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wg_rsrc_t wg_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  return (wg_rsrc_t){(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ void wg_dma16(wg_rsrc_t rs, const unsigned char* lds_dst, unsigned voff) {
  const unsigned lds = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) const unsigned char*)lds_dst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(lds), "s"(rs)
               : "memory");
}


// An 8-pixel DMA piece (8 rows x 128 B) sits on a pitch of 1024 + 128 B (round 6): the two 16-lane groups a transposing read
// services together read rows k and 8 + k of the slab -- 1024 B apart they start on the SAME banks whatever the slot
// swizzle does (the swizzle depends on the row within a piece only), a 2-way conflict on every fragment read (SQ counters,
// round 5: bank-conflict cycles = 0.49 of the LDS-active cycles).  One row of padding per piece rotates the odd pieces by
// 32 banks: rows k and 8 + k then use complementary halves of the 64 banks for every slot pair.
constexpr int PIECE = 1024 + 128;
constexpr int SLAB_BYTES = 4 * PIECE;       // 32 pixels x 64 channels x 2 B, in four padded pieces
constexpr int WAVE_LDS = 4 * SLAB_BYTES;    // {dy, x} x double buffer

// LDS slabs are [pixel row][8 slots of 16 B]; row r keeps channel chunk c in slot c ^ swz(r): un-swizzled, the 16 rows a
// transposing read touches all start on the same banks (row pitch 128 B = one full bank cycle).  With swz the 4 rows x
// 32 B of a 16-lane group cover all 32 banks.  (Measured: 0-4 % -- the kernel is bound by the L2 -> LDS DMA path and by
// per-workgroup overheads, not by LDS reads; kept because it is free.)
__device__ __forceinline__ int swz(int r) { return ((r & 3) << 1) | ((r >> 2) & 1); }

__device__ __forceinline__ u32x4 tr_frag(const unsigned char* slab, int tile, int lane) {
  // rows 8g..8g+7 (pixels) of channel tile `tile`: two transposing reads of [4 pixels][16 channels] blocks
  const int i = lane & 15, g = lane >> 4;
  const int k = i >> 2;                                   // row within the 4-row block
  const int chunk = tile * 2 + ((i & 3) >> 1);
  const unsigned char* a0 = slab + g * PIECE + k * 128 + ((chunk ^ (k << 1)) << 4) + (i & 1) * 8;          // rows 8g + k
  const unsigned char* a1 = slab + g * PIECE + (4 + k) * 128 + ((chunk ^ ((k << 1) | 1)) << 4) + (i & 1) * 8;   // rows 8g + 4 + k
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a1);
  u32x4 r;
  r[0] = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  r[1] = (uint16_t)lo[2] | ((uint32_t)(uint16_t)lo[3] << 16);
  r[2] = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
  r[3] = (uint16_t)hi[2] | ((uint32_t)(uint16_t)hi[3] << 16);
  return r;
}

// the all-ones B fragment of the bias column (bf16 1.0 = 0x3f80, fp16 1.0 = 0x3c00)
template <typename T>
__device__ __forceinline__ u32x4 ones_frag() {
  constexpr uint32_t one = std::is_same<T, BF16>::value ? 0x3f803f80u : 0x3c003c00u;
  return (u32x4){one, one, one, one};
}
// accb[a]: rows = output channels co0 + 16 a + 4 g + r, every column the same sum: column 0's lanes store the row
__device__ __forceinline__ void store_bias_row(float* __restrict__ row, const f32x4 (&accb)[4], int na, int co0, int cout_s,
                                               int lane) {
  if ((lane & 15) != 0) return;
  const int g = lane >> 4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int co = co0 + a * 16 + 4 * g;
    if (a < na && co < cout_s) *reinterpret_cast<f32x4*>(row + co) = accb[a];
  }
}

// Column `col` (0..63) of the N tile (tap slot `slot`, ci block `cib`) -> (tap, ci).  Normal layout: one tap per slot,
// 64 consecutive input channels.  Folded layout (cin_s <= 32): a tile row holds `tpt` taps x `cpt` 16-byte channel chunks,
// so a 3-channel 3x3 conv needs 2 tiles per co block instead of 9 that are 7/8 zero padding.
__device__ __forceinline__ bool wgrad_column(int fold, int cpt, int tpt, int taps, int cin, int slot, int cib, int col,
                                             int& tap, int& ci) {
  if (!fold) {
    tap = slot;
    ci = cib * 64 + col;
    return ci < cin;
  }
  const int q = col >> 3, tl = q / cpt;
  tap = slot * tpt + tl;
  ci = (q - tl * cpt) * 8 + (col & 7);
  return tl < tpt && tap < taps && ci < cin;
}

// MODE 0: zero padding; 1: x stored at half resolution and read through the folded nearest x2 upsample; 2: reflect padding
// UNI: w_out % 8 == 0, so the 8 pixels of a DMA piece lie in one output row and the piece's coordinates are wave-uniform:
// they live in scalar registers and advance on the scalar unit; a lane only adds its constant part.
template <typename T, int MODE, bool UNI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Work items are ordered (pixel split, ci block, tap, co block) with the co block fastest, and each XCD (blockIdx.x
  // & 7: one L2) takes one CONTIGUOUS range of them, so the co blocks and taps of one (pixel split, ci block) run
  // together on one XCD and walk the same chunks: their x / dy slabs are L2 hits instead of HBM re-reads.
  const int taps_n = p.kh * p.kw;
  int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  const int tiles_n = p.tap_slots * p.ci_blocks * p.co_blocks;
  if ((int)(blockIdx.x >> 3) >= p.per_xcd || item >= tiles_n * p.splits) return;
  const int cob = item % p.co_blocks;
  item /= p.co_blocks;
  const int slot = item % p.tap_slots;    // tap (normal layout) or group of tpt taps (folded layout)
  item /= p.tap_slots;
  const int cib = item % p.ci_blocks;
  const int split = item / p.ci_blocks;
  const int co0 = cob * 64;
  unsigned char* wl = smem + wave * WAVE_LDS;

  const int prow = lane >> 3;          // pixel within an 8-pixel DMA piece
  const int qs = (lane & 7) ^ swz(prow);   // channel chunk this lane stages (LDS slot lane & 7 of row prow, swizzled)
  const int q8 = qs * 8;               // first channel of this lane's 16-byte chunk
  const bool co_ok = co0 + q8 < p.cout_s;
  // this lane's 16-byte chunk of an x row: (tap, first channel)
  int ky, kx, cch;
  bool ci_ok;
  if (p.fold) {
    const int q = qs, tl = q / p.cpt;
    const int tap_l = slot * p.tpt + tl;
    ci_ok = tl < p.tpt && tap_l < taps_n;
    const int tcl = ci_ok ? tap_l : 0;
    ky = tcl / p.kw;
    kx = tcl - ky * p.kw;
    cch = (q - tl * p.cpt) * 8;
  } else {
    ky = slot / p.kw;
    kx = slot - ky * p.kw;
    cch = cib * 64 + q8;
    ci_ok = cch < p.cin_s;
  }
  // ---- addressing of the DMA pieces.  Everything is 32-bit and incremental: the first version recomputed 64-bit
  // offsets ((n h + iy) w + ix) cin per piece, ~35 instructions with quarter-rate 64-bit multiplies behind exec-masked
  // branches -- 8 pieces per chunk cost several times the chunk's 16 MFMAs and bound the kernel (ablation: without the
  // DMA *and its address math* it ran twice as fast).  Byte offsets fit 32 bits (host check: tensors < 4 GiB).
  // Buffer descriptors: a lane whose pixel / tap / channel chunk does not exist sends an out-of-range offset and the
  // hardware writes zeros to its LDS slot -- no zero page, no 64-bit pointer selects.
  const wg_rsrc_t rs_dy = wg_make_rsrc(p.dy, p.dy_bytes);
  const wg_rsrc_t rs_x = wg_make_rsrc(p.x, p.x_bytes);
  const unsigned cin_b = (unsigned)p.cin_s * 2u, cout_b = (unsigned)p.cout_s * 2u;
  const unsigned cch2 = (unsigned)cch * 2u;
  const int tap_y = ky * p.dil - p.pad, tap_x = kx * p.dil - p.pad;
  const int hx = MODE == 1 ? (p.h_in >> 1) : p.h_in, wx = MODE == 1 ? (p.w_in >> 1) : p.w_in;   // stored extent of x
  const unsigned row_b = (unsigned)wx * cin_b;                                                    // bytes per stored row

  // State of this lane's 4 pieces for the NEXT chunk to issue: linear pixel, (ox, oy) pre-multiplied by the stride,
  // n * (stored rows per image), dy byte offset.  Advanced by the uniform chunk stride with carries.
  int c_pix[4], c_sx[4], c_sy[4], c_nh[4];
  unsigned c_dy[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pix = split * 128 + wave * 32 + i * 8 + prow;
    c_pix[i] = pix;
    const int r = pix / p.w_out;
    c_sx[i] = (pix - r * p.w_out) * p.stride;
    const int nn = r / p.h_out;
    c_sy[i] = (r - nn * p.h_out) * p.stride;
    c_nh[i] = nn * hx;
    c_dy[i] = (unsigned)pix * cout_b + (unsigned)(co0 + q8) * 2u;
  }
  const int step = p.splits * 128;                                   // pixels between two chunks of this workgroup
  const int step_r = step / p.w_out;
  const int step_sx = (step - step_r * p.w_out) * p.stride;
  const int step_n = step_r / p.h_out;
  const int step_sy = (step_r - step_n * p.h_out) * p.stride;
  const int step_nh = step_n * hx;
  const int wrap_x = p.w_out * p.stride, wrap_y = p.h_out * p.stride;
  const unsigned step_dy = (unsigned)step * cout_b;

  // ---- UNI: per-piece state from wave-uniform values only (row 0 of the piece), per-lane constants on top.  The four
  // pieces' coordinates live in the lanes with (lane & 3) == i: ONE vector update per chunk advances all of them and the
  // issue loop fetches a piece's values with v_readlane (the first version kept them in scalar registers: ~28 scalar
  // instructions of carries per piece, more issue time than the chunk's 16 MFMAs -- see the cooperative kernel)
  int v_sx, v_sy, v_nh;
  {
    const int pix = split * 128 + wave * 32 + (lane & 3) * 8;
    const int r = pix / p.w_out;
    v_sx = (pix - r * p.w_out) * p.stride;
    const int nn = r / p.h_out;
    v_sy = (r - nn * p.h_out) * p.stride;
    v_nh = nn * hx;
  }
  unsigned u_dy0 = (unsigned)(split * 128 + wave * 32) * cout_b;
  const int lx = prow * p.stride + tap_x;
  const int ly = ci_ok ? tap_y : -(1 << 20);                    // a lane without a tap / channel chunk is always out of bounds
  const unsigned lane_dyc = co_ok ? (unsigned)prow * cout_b + (unsigned)(co0 + q8) * 2u : 0x80000000u;   // dy_bytes < 2^31
  const unsigned lane_xc = (unsigned)((tap_y * wx + lx) * (int)cin_b) + cch2;                            // MODE 0

  // DMA of this wave's 32-pixel slab of the next chunk into buffer b: 4 pieces of dy, 4 pieces of (tap-shifted) x
  auto issue = [&](int b) {
    unsigned char* dst_dy = wl + b * 2 * SLAB_BYTES;
    unsigned char* dst_x = dst_dy + SLAB_BYTES;
    if constexpr (UNI) {
      const unsigned v_off = (unsigned)((v_nh + v_sy) * wx + v_sx) * cin_b;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // pixels past the end: the dy offset is out of range by itself (zeros), which also neutralises whatever x holds
        wg_dma16(rs_dy, dst_dy + i * PIECE, u_dy0 + (unsigned)(i * 8) * cout_b + lane_dyc);
        int iy = __builtin_amdgcn_readlane(v_sy, i) + ly, ix = __builtin_amdgcn_readlane(v_sx, i) + lx;
        if (MODE == 2) {      // nn.ReflectionPad2d: mirror without repeating the border (a dead lane stays out of range)
          iy = iy < 0 ? -iy : (iy >= p.h_in ? 2 * p.h_in - 2 - iy : iy);
          ix = ix < 0 ? -ix : (ix >= p.w_in ? 2 * p.w_in - 2 - ix : ix);
        }
        const bool xv = ((unsigned)iy < (unsigned)p.h_in) & ((unsigned)ix < (unsigned)p.w_in);
        unsigned off;
        if (MODE == 1) {
          off = (unsigned)(__builtin_amdgcn_readlane(v_nh, i) + (iy >> 1)) * row_b + (unsigned)(ix >> 1) * cin_b + cch2;
        } else if (MODE == 2) {
          off = (unsigned)(__builtin_amdgcn_readlane(v_nh, i) + iy) * row_b + (unsigned)ix * cin_b + cch2;
        } else {
          off = (unsigned)__builtin_amdgcn_readlane((int)v_off, i) + lane_xc;   // piece part + lane constant
        }
        wg_dma16(rs_x, dst_x + i * PIECE, xv ? off : 0xffffffffu);
      }
      u_dy0 += step_dy;
      int sx = v_sx + step_sx, sy = v_sy + step_sy, nh = v_nh + step_nh;
      const bool cx = sx >= wrap_x;
      sx = cx ? sx - wrap_x : sx;
      sy = cx ? sy + p.stride : sy;
      const bool cy = sy >= wrap_y;
      sy = cy ? sy - wrap_y : sy;
      nh = cy ? nh + hx : nh;
      v_sx = sx; v_sy = sy; v_nh = nh;
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool pv = c_pix[i] < p.npix;
      wg_dma16(rs_dy, dst_dy + i * PIECE, (pv && co_ok) ? c_dy[i] : 0xffffffffu);
      int iy = c_sy[i] + tap_y, ix = c_sx[i] + tap_x;
      if (MODE == 2) {
        iy = iy < 0 ? -iy : (iy >= p.h_in ? 2 * p.h_in - 2 - iy : iy);
        ix = ix < 0 ? -ix : (ix >= p.w_in ? 2 * p.w_in - 2 - ix : ix);
      }
      const bool xv = pv && ci_ok && (unsigned)iy < (unsigned)p.h_in && (unsigned)ix < (unsigned)p.w_in;
      const unsigned row = (unsigned)(c_nh[i] + (MODE == 1 ? (iy >> 1) : iy));
      const unsigned col = (unsigned)(MODE == 1 ? (ix >> 1) : ix);
      const unsigned off = row * row_b + col * cin_b + cch2;
      wg_dma16(rs_x, dst_x + i * PIECE, xv ? off : 0xffffffffu);
      // advance to the following chunk
      c_pix[i] += step;
      c_dy[i] += step_dy;
      int sx = c_sx[i] + step_sx, sy = c_sy[i] + step_sy, nh = c_nh[i] + step_nh;
      const bool cx = sx >= wrap_x;
      sx = cx ? sx - wrap_x : sx;
      sy = cx ? sy + p.stride : sy;
      const bool cy = sy >= wrap_y;
      sy = cy ? sy - wrap_y : sy;
      nh = cy ? nh + hx : nh;
      c_sx[i] = sx; c_sy[i] = sy; c_nh[i] = nh;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool do_bias = p.bpart != nullptr && slot == 0 && cib == 0;      // block-uniform
  f32x4 accb[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int c = split, buf = 0;
  if (c < p.nchunks) issue(0);
  for (; c < p.nchunks; c += p.splits) {
    const int cn = c + p.splits;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // earlier fragment reads of the other buffer are done
    if (cn < p.nchunks) {
      issue(buf ^ 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned char* sdy = wl + buf * 2 * SLAB_BYTES;
    const unsigned char* sx = sdy + SLAB_BYTES;
    u32x4 fa[4], fb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) fa[a] = tr_frag(sdy, a, lane);
#pragma unroll
    for (int b = 0; b < 4; ++b) fb[b] = tr_frag(sx, b, lane);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(as_vec8<T>(fa[a]), as_vec8<T>(fb[b]), acc[a][b]);
    if (do_bias) {
#pragma unroll
      for (int a = 0; a < 4; ++a) accb[a] = mfma16(as_vec8<T>(fa[a]), as_vec8<T>(ones_frag<T>()), accb[a]);
    }
    buf ^= 1;
  }
  // bias rows: one per (pixel split, wave) -- a wave covers its own 32 pixels of every chunk
  if (do_bias) store_bias_row(p.bpart + ((size_t)split * 4 + wave) * p.cout_s, accb, 4, co0, p.cout_s, lane);

  // ---- sum the four waves' partial tiles through LDS, then fp32 atomics into dW (OIHW)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      *reinterpret_cast<f32x4*>(smem + ((wave * 16 + a * 4 + b) * 64 + lane) * 16) = acc[a][b];
  __syncthreads();
  const int j = lane & 15, g = lane >> 4;
  f32x4* wst = nullptr;
  if (p.ws) {
    const int tile = (slot * p.ci_blocks + cib) * p.co_blocks + cob;
    wst = reinterpret_cast<f32x4*>(p.ws) + ((size_t)split * tiles_n + tile) * 1024;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int a = wave;   // this wave finalises co tile `wave`
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(smem + ((w * 16 + a * 4 + b) * 64 + lane) * 16);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    if (wst) {
      wst[(a * 4 + b) * 64 + lane] = s;   // fragment order: 16 B per lane, coalesced
      continue;
    }
    int tap, ci;
    const bool col_ok = wgrad_column(p.fold, p.cpt, p.tpt, taps_n, p.cin, slot, cib, b * 16 + j, tap, ci);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + a * 16 + 4 * g + r;
      if (co < p.cout && col_ok && !(CGAN_DBG(p) & 1)) atomicAdd(p.dw + ((size_t)co * p.cin + ci) * taps_n + tap, s[r]);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Cooperative variant for large-pixel layers (w_out % 8 == 0; zero or reflect padding, folded upsample): one workgroup = a
// 128 co x 128 column tile (2 co blocks x 2 N tiles), the four waves each own a 64 x 64 quadrant and SHARE the staged
// slabs: per 64-pixel chunk wave 0 / 1 stage the dy slab of co half 0 / 1, wave 2 / 3 the x slab of N tile 0 / 1 (8 KiB
// each), then every wave reads one dy and one x slab: 32 KiB of L2 -> LDS traffic per 128 MFMAs instead of per 64.
// One barrier per chunk; a wave's quadrant is a complete tile (no cross-wave reduction).  Needs 4x the pixel splits of
// the single-wave kernel for the same number of workgroups, i.e. 4x the partial-tile workspace: only layers with
// enough pixels per split use it (wgrad_plan).
// CH = pixels per stage (64 or 32); a sub-slab is CH pixels x 64 channels x 2 B; a stage = dy half 0 | dy half 1 |
// x tile 0 | x tile 1; two stages: 64 KiB (2 workgroups per CU) or 32 KiB (4 per CU) of LDS.

// G = 4 (round 5): the same kernel on a 4 x 4 grid of waves -- a 256 co x 256 column tile per workgroup, 16 waves (four per
// SIMD, one workgroup per CU, 128 KiB of LDS for two 64-pixel stages): every staged sub-slab now feeds FOUR quadrants instead
// of two, i.e. half the L2 -> LDS bytes and half the LDS-DMA pieces per MFMA (4 pieces per wave and chunk instead of 8: the
// pieces' issue time was as long as the chunk's MFMAs, tools/ts_wgrad.py), and four waves per SIMD to cover one another's
// piece issue and transposing reads.  Needs co and N block counts divisible by four (256-channel multiples on both sides:
// ResNet layer3 / layer4, ASPP, the decoders' 512-channel convs).
template <typename T, int MODE, int CH = 64, bool TS = false, bool BIAS = false, int G = 2>
__global__ __launch_bounds__(64 * G * G, G == 4 ? 1 : (CH == 64 ? 2 : 4)) void conv_wgrad_coop_kernel(WgradArgs p) {
  constexpr int CHUNK2 = CH, SUB2 = (CH / 8) * PIECE, STAGE2 = 2 * G * SUB2;
  constexpr int PPS = CH / 8, PW = PPS / G;      // pieces per sub-slab, pieces of each operand a wave stages
  static_assert(PW >= 1, "a wave stages at least one piece of each operand");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int taps_n = p.kh * p.kw;
  int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= p.per_xcd || item >= p.co_pairs * p.n_pairs * p.splits) return;
  const int cop = item % p.co_pairs;
  item /= p.co_pairs;
  const int np = item % p.n_pairs;
  const int split = item / p.n_pairs;
  const int cpairs = p.fold ? 1 : (p.ci_blocks + G - 1) / G;    // N-tile groups per tap slot (normal layout)
  // N tile (slot, cib) of group member m
  auto n_tile = [&](int m, int& slot, int& cib) {
    if (p.fold) { slot = np * G + m; cib = 0; return slot < p.tap_slots; }
    slot = np / cpairs;
    cib = (np - slot * cpairs) * G + m;
    return cib < p.ci_blocks;
  };

  // ---- staging: wave (mem, half) stages pieces half*4 .. half*4+3 (8 pixels each) of dy sub-slab `mem` (co half) AND of x
  // sub-slab `mem` (N tile).  (The first version gave whole sub-slabs to the waves, dy to waves 0 / 1 and x to waves 2 / 3:
  // an x piece cost ~4x a dy piece in issue time -- ~28 scalar instructions of coordinate carries per piece -- and the dy
  // waves spent 40 % of their life at the barrier waiting for the x waves, tools/ts_wgrad.py.)
  const int prow = lane >> 3, qs = (lane & 7) ^ swz(prow), q8 = qs * 8;
  const int mem = wave / G, half = wave % G;     // sub-slab (co block of the group / N tile of the group), its part
  const unsigned cin_b = (unsigned)p.cin_s * 2u, cout_b = (unsigned)p.cout_s * 2u;
  const int hx = MODE == 1 ? (p.h_in >> 1) : p.h_in, wx = MODE == 1 ? (p.w_in >> 1) : p.w_in;
  const unsigned row_b = (unsigned)wx * cin_b;
  const wg_rsrc_t rs_x = wg_make_rsrc(p.x, p.x_bytes);
  const wg_rsrc_t rs_dy = wg_make_rsrc(p.dy, p.dy_bytes);
  unsigned dy_c = 0x80000000u;                  // dy lane constant (or the always-out-of-range marker)
  {
    const int co = (cop * G + mem) * 64 + q8;
    if ((cop * G + mem) < p.co_blocks && co < p.cout_s) dy_c = (unsigned)prow * cout_b + (unsigned)co * 2u;
  }
  int lx, ly;                                   // x: the lane's tap offset (lx far out of range for a dead lane)
  unsigned cch2, x_c;
  {
    int slot, cib, ky, kx, cch;
    bool ok = n_tile(mem, slot, cib);
    if (p.fold) {
      const int tl = qs / p.cpt, tap_l = slot * p.tpt + tl;
      ok = ok && tl < p.tpt && tap_l < taps_n;
      const int tcl = ok ? tap_l : 0;
      ky = tcl / p.kw;
      kx = tcl - ky * p.kw;
      cch = (qs - tl * p.cpt) * 8;
    } else {
      ky = slot / p.kw;
      kx = slot - ky * p.kw;
      cch = cib * 64 + q8;
      ok = ok && cch < p.cin_s;
    }
    const int tap_y = ky * p.dil - p.pad, tap_x = kx * p.dil - p.pad;
    const int lxr = prow * p.stride + tap_x;
    ly = tap_y;
    lx = ok ? lxr : -(1 << 20);
    cch2 = (unsigned)cch * 2u;
    x_c = (unsigned)((tap_y * wx + lxr) * (int)cin_b) + cch2;     // MODE 0: added to the piece's pixel offset
  }

  // coordinates of the 8 pieces of the next chunk to stage, piece i in lanes with (lane & 7) == i (every lane keeps one
  // piece's state and one vector update per chunk advances all eight; the issue loop reads its pieces with v_readlane)
  int v_sx, v_sy, v_nh;
  {
    const int pix = split * CHUNK2 + (lane & (PPS - 1)) * 8;
    const int r = pix / p.w_out;
    v_sx = (pix - r * p.w_out) * p.stride;
    const int nn = r / p.h_out;
    v_sy = (r - nn * p.h_out) * p.stride;
    v_nh = nn * hx;
  }
  unsigned u_dy = (unsigned)(split * CHUNK2) * cout_b;
  const int step = p.splits * CHUNK2;
  const int step_r = step / p.w_out;
  const int step_sx = (step - step_r * p.w_out) * p.stride;
  const int step_n = step_r / p.h_out;
  const int step_sy = (step_r - step_n * p.h_out) * p.stride;
  const int step_nh = step_n * hx;
  const int wrap_x = p.w_out * p.stride, wrap_y = p.h_out * p.stride;
  const unsigned step_dy = (unsigned)step * cout_b;

  // piece k of the chunk to stage into buffer b: k < PW the wave's dy pieces, k >= PW its x pieces.  All are issued
  // right after the chunk barrier, before the fragment reads: spreading them between the MFMA rows was tried and is slower
  // (the MFMAs queue behind a piece's issue stall either way, and the pieces land later: l3 3x3 115 -> 125 us).
  unsigned v_off = 0;
  auto issue_piece = [&](int b, int k) {
    unsigned char* dst = smem + b * STAGE2 + mem * SUB2 + half * (PW * PIECE);
    if (k < PW) {
      const unsigned off = u_dy + (unsigned)((half * PW + k) * 8) * cout_b + dy_c;
      wg_dma16(rs_dy, dst + k * PIECE, off);
      return;
    }
    const int j = k - PW, i = half * PW + j;
    if (j == 0) v_off = (unsigned)((v_nh + v_sy) * wx + v_sx) * cin_b;
    int iy = __builtin_amdgcn_readlane(v_sy, i) + ly, ix = __builtin_amdgcn_readlane(v_sx, i) + lx;
    if (MODE == 2) {          // nn.ReflectionPad2d (a dead lane stays out of range)
      iy = iy < 0 ? -iy : (iy >= p.h_in ? 2 * p.h_in - 2 - iy : iy);
      ix = ix < 0 ? -ix : (ix >= p.w_in ? 2 * p.w_in - 2 - ix : ix);
    }
    const bool xv = ((unsigned)iy < (unsigned)p.h_in) & ((unsigned)ix < (unsigned)p.w_in);   // (no short-circuit branches)
    unsigned off;
    if (MODE == 1) off = (unsigned)(__builtin_amdgcn_readlane(v_nh, i) + (iy >> 1)) * row_b + (unsigned)(ix >> 1) * cin_b + cch2;
    else if (MODE == 2) off = (unsigned)(__builtin_amdgcn_readlane(v_nh, i) + iy) * row_b + (unsigned)ix * cin_b + cch2;
    else off = (unsigned)__builtin_amdgcn_readlane((int)v_off, i) + x_c;
    wg_dma16(rs_x, dst + G * SUB2 + j * PIECE, xv ? off : 0xffffffffu);
  };
  auto advance = [&]() {
    u_dy += step_dy;
    int sx = v_sx + step_sx, sy = v_sy + step_sy, nh = v_nh + step_nh;
    const bool cx = sx >= wrap_x;
    sx = cx ? sx - wrap_x : sx;
    sy = cx ? sy + p.stride : sy;
    const bool cy = sy >= wrap_y;
    sy = cy ? sy - wrap_y : sy;
    nh = cy ? nh + hx : nh;
    v_sx = sx; v_sy = sy; v_nh = nh;
  };

  // ---- compute role: quadrant (co block wc of the group, N tile wn)
  const int wc = wave % G, wn = wave / G;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // (BIAS is a template parameter: the four extra accumulators do not fit the 128-register budget of the 32-pixel-stage
  // variant, and a run-time flag would make every instance pay for them)
  const bool do_bias = BIAS && p.bpart != nullptr && np == 0 && wn == 0;      // wave-uniform: the quadrants of N tile 0
  f32x4 accb[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nchunks = (p.npix + CHUNK2 - 1) / CHUNK2;
  int c = split, buf = 0;
  unsigned long long t_a = 0, t_b = 0, ts_sum[4] = {0, 0, 0, 0}, t_start = 0;
  if (TS) t_start = __builtin_readcyclecounter();
  if (c < nchunks) {
#pragma unroll
    for (int k = 0; k < 2 * PW; ++k) issue_piece(0, k);
    advance();
  }
  for (; c < nchunks; c += p.splits) {
    if (TS) t_a = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own pieces landed, own fragment reads done
    if (TS) { t_b = __builtin_readcyclecounter(); ts_sum[0] += t_b - t_a; }
    __syncthreads();                                              // everyone's pieces landed; the other buffer is free
    if (TS) { t_a = __builtin_readcyclecounter(); ts_sum[1] += t_a - t_b; }
    if (c + p.splits < nchunks) {
#pragma unroll
      for (int k = 0; k < 2 * PW; ++k) issue_piece(buf ^ 1, k);
      advance();
    }
    if (TS) { t_b = __builtin_readcyclecounter(); ts_sum[2] += t_b - t_a; }
    const unsigned char* sdy = smem + buf * STAGE2 + wc * SUB2;
    const unsigned char* sx = smem + buf * STAGE2 + (G + wn) * SUB2;
#pragma unroll
    for (int ks = 0; ks < CH / 32; ++ks) {
      u32x4 fa[4], fb[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) fa[a] = tr_frag(sdy + ks * 4 * PIECE, a, lane);
#pragma unroll
      for (int b = 0; b < 4; ++b) fb[b] = tr_frag(sx + ks * 4 * PIECE, b, lane);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(as_vec8<T>(fa[a]), as_vec8<T>(fb[b]), acc[a][b]);
      if (BIAS && do_bias) {
#pragma unroll
        for (int a = 0; a < 4; ++a) accb[a] = mfma16(as_vec8<T>(fa[a]), as_vec8<T>(ones_frag<T>()), accb[a]);
      }
    }
    buf ^= 1;
    if (TS) { t_a = __builtin_readcyclecounter(); ts_sum[3] += t_a - t_b; }
  }
  if (BIAS && do_bias && cop * G + wc < p.co_blocks)
    store_bias_row(p.bpart + (size_t)split * p.cout_s, accb, 4, (cop * G + wc) * 64, p.cout_s, lane);
  if (TS && CGAN_WTS(p) && lane == 0) {
    unsigned long long* o = CGAN_WTS(p) + ((size_t)blockIdx.x * (G * G) + wave) * 8;
    o[0] = t_start;
    o[1] = __builtin_readcyclecounter();
    for (int i = 0; i < 4; ++i) o[2 + i] = ts_sum[i];
    o[6] = (unsigned long long)((nchunks - split + p.splits - 1) / p.splits);
  }

  // ---- the quadrant is a complete 64 x 64 tile of this pixel split
  const int cob = cop * G + wc;
  int slot_c, cib_c;
  const bool nt_ok = n_tile(wn, slot_c, cib_c);
  if (cob >= p.co_blocks || !nt_ok) return;
  if (p.ws) {
    const int tiles_n = p.tap_slots * p.ci_blocks * p.co_blocks;
    const int tile = (slot_c * p.ci_blocks + cib_c) * p.co_blocks + cob;
    f32x4* wst = reinterpret_cast<f32x4*>(p.ws) + ((size_t)split * tiles_n + tile) * 1024;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) wst[(a * 4 + b) * 64 + lane] = acc[a][b];
    return;
  }
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    int tap, ci;
    const bool col_ok = wgrad_column(p.fold, p.cpt, p.tpt, taps_n, p.cin, slot_c, cib_c, b * 16 + j, tap, ci);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cob * 64 + a * 16 + 4 * g + r;
        if (co < p.cout && col_ok && !(CGAN_DBG(p) & 1)) atomicAdd(p.dw + ((size_t)co * p.cin + ci) * taps_n + tap, acc[a][b][r]);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Spatially tiled variant for 3 x 3 / stride 1 / pad 1 layers on large maps (the SPADE gamma|beta gradients: 128 hidden
// channels at 160^2 ... 640^2).  The kernels above treat every tap as its own N tile, so each tap's workgroups stream
// their own shifted copy of x through L2 -> LDS: 9 x 419 MB of the 6.2 GB a 128 -> 40 call at 4 x 640^2 moved.  Here a
// workgroup owns a (64 co) x (64 ci x 9 taps) block of dW -- 36 accumulator tiles per wave -- and walks 4 x 32 pixel
// tiles: per tile it stages the dy tile (128 px) and the x tile WITH ITS HALO (6 rows x 40 px) once, and all nine taps
// read their shifted windows from that one copy (the transposing LDS read takes a row address per lane, so a shifted
// window costs nothing): 46 DMA pieces per 576 MFMAs instead of 8 per 32.
// wave w owns N tiles 9w .. 9w+8 of the 36 (N tile j = tap j / 4, 16-channel group j % 4).
constexpr int TL_W = 32, TL_H = 4, TL_XP = 40;                 // pixel tile, LDS pitch (pixels) of a halo row
constexpr int TL_XH_BYTES = (TL_H + 2) * TL_XP * 128;           // 30720
constexpr int TL_DY_BYTES = TL_H * TL_W * 128;                 // 16384

__device__ __forceinline__ u32x4 tr_frag_at(const unsigned char* slab, int q0, int tile, int lane) {
  // the 32 pixel rows q0 .. q0+31 (any q0) of channel tile `tile`; a row q keeps chunk c in slot c ^ swz(q & 7)
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

template <typename T, int NA = 4, bool BIAS = false>   // NA: 16-row co tiles per block that exist (a 40-channel gradient: 3)
__global__ __launch_bounds__(256, 2) void conv_wgrad_tile3x3_kernel(WgradArgs p) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* xh = smem;
  unsigned char* dyt = smem + TL_XH_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int item = (blockIdx.x & 7) * p.per_xcd + (blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= p.per_xcd || item >= p.co_blocks * p.ci_blocks * p.splits) return;
  const int cob = item % p.co_blocks;
  item /= p.co_blocks;
  const int cib = item % p.ci_blocks;
  const int split = item / p.ci_blocks;

  const int prow = lane >> 3, qs = (lane & 7) ^ swz(prow);
  const unsigned cin_b = (unsigned)p.cin_s * 2u, cout_b = (unsigned)p.cout_s * 2u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.dy), 0, p.dy_bytes, 0x00020000);
  const bool ci_ok = cib * 64 + qs * 8 < p.cin_s, co_ok = cob * 64 + qs * 8 < p.cout_s;
  const unsigned x_c = (unsigned)(cib * 64 + qs * 8) * 2u, dy_c = (unsigned)(cob * 64 + qs * 8) * 2u;

  const int tw = p.w_out / TL_W, th = p.h_out / TL_H;
  const int ntiles = p.n * th * tw;

  // staging of one spatial tile: x rows {wave, wave + 4} of the 6 halo rows (5 pieces each), dy pieces: waves 0 / 1 two
  // each, waves 2 / 3 six each (12 / 12 / 11 / 11 pieces per wave)
  auto stage = [&](int t) {
    const int tc = t % tw;
    const int r0 = t / tw;
    const int tr = r0 % th, img = r0 / th;
    const int ty0 = tr * TL_H, tx0 = tc * TL_W;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = wave + 4 * rr;
      if (r < TL_H + 2) {
        const int iy = ty0 - 1 + r;
        const unsigned row_off = (unsigned)((img * p.h_in + iy) * p.w_in) * cin_b + x_c;
#pragma unroll
        for (int i5 = 0; i5 < 5; ++i5) {
          const int ix = tx0 - 4 + 8 * i5 + prow;
          const bool ok = ci_ok & ((unsigned)iy < (unsigned)p.h_in) & ((unsigned)ix < (unsigned)p.w_in);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(xh + (r * TL_XP + 8 * i5) * 128), 16,
                                                   ok ? row_off + (unsigned)ix * cin_b : 0xffffffffu, 0, 0, 0);
        }
      }
    }
    const int j0 = wave < 2 ? wave * 2 : 4 + (wave - 2) * 6, nj = wave < 2 ? 2 : 6;
#pragma unroll
    for (int jj = 0; jj < 6; ++jj) {
      if (jj < nj) {
        const int j = j0 + jj, r = j >> 2, c8 = (j & 3) * 8;
        const unsigned off = (unsigned)((img * p.h_out + ty0 + r) * p.w_out + tx0 + c8 + prow) * cout_b + dy_c;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (__attribute__((address_space(3))) void*)(dyt + (r * TL_W + c8) * 128), 16,
                                                 co_ok ? off : 0xffffffffu, 0, 0, 0);
      }
    }
  };

  f32x4 acc[NA][9];
#pragma unroll
  for (int a = 0; a < NA; ++a)
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[a][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const bool do_bias = BIAS && p.bpart != nullptr && cib == 0 && wave == 0;   // (template parameter: NA = 4 has no registers left)
  f32x4 accb[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int t = split; t < ntiles; t += p.splits) {
    __syncthreads();                                              // the previous tile's fragment reads are done
    stage(t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int ty = 0; ty < TL_H; ++ty) {
      u32x4 fa[NA];
#pragma unroll
      for (int a = 0; a < NA; ++a) fa[a] = tr_frag_at(dyt, ty * TL_W, a, lane);
      if (BIAS && do_bias) {
#pragma unroll
        for (int a = 0; a < NA; ++a) accb[a] = mfma16(as_vec8<T>(fa[a]), as_vec8<T>(ones_frag<T>()), accb[a]);
      }
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const int jn = wave * 9 + i, tap = jn >> 2, grp = jn & 3;
        const int ky = tap / 3, kx = tap - ky * 3;
        const u32x4 fb = tr_frag_at(xh, (ty + ky) * TL_XP + 3 + kx, grp, lane);
#pragma unroll
        for (int a = 0; a < NA; ++a) acc[a][i] = mfma16(as_vec8<T>(fa[a]), as_vec8<T>(fb), acc[a][i]);
      }
    }
  }

  if (BIAS && do_bias) store_bias_row(p.bpart + (size_t)split * p.cout_s, accb, NA, cob * 64, p.cout_s, lane);
  // partial tiles to the workspace in the layout wgrad_reduce_kernel sums: [split][64 x 64 tile (tap, cib, cob)][a * 4 + b][lane]
  const int tiles_n = 9 * p.ci_blocks * p.co_blocks;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int jn = wave * 9 + i, tap = jn >> 2, grp = jn & 3;
    const int tile = (tap * p.ci_blocks + cib) * p.co_blocks + cob;
    f32x4* wst = reinterpret_cast<f32x4*>(p.ws) + ((size_t)split * tiles_n + tile) * 1024;
#pragma unroll
    for (int a = 0; a < 4; ++a) wst[(a * 4 + grp) * 64 + lane] = a < NA ? acc[a < NA ? a : 0][i] : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
}

// Second stage of the workspace path: dW += sum over pixel splits of the partial tiles.  A block = 32 consecutive
// f32x4 of a tile x 8 split lanes: each thread sums every 8th split, the 8 lanes meet in LDS, lane 0 does the
// read-modify-write of dW.  No atomics (the first version finished its split groups with up-to-16-way contended
// cross-XCD atomics, which cost more than the 65 MB of partial tiles they followed).
// (KL = 1 for layers with <= 4 splits: 256 elements per block, no LDS step.)
// The blocks past ``main_blocks`` finish the bias gradient the same way: dbias[ch] += the channel_sum_kernel blocks' partial
// rows, summed in block order (one thread per channel).
template <int KL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const f32x4* __restrict__ ws, float* __restrict__ dw,
                                                           int splits, int taps, int tap_slots, int ci_blocks,
                                                           int co_blocks, int cout, int cin, int fold, int cpt,
                                                           int tpt, int main_blocks, const float* __restrict__ bpart,
                                                           int bias_rows, int cout_s, float* __restrict__ dbias) {
  if ((int)blockIdx.x >= main_blocks) {
    // 16 channels x 16 row lanes per block: lane rl adds rows rl, rl + 16, ... (four loads in flight), the 16 lanes of a
    // channel meet in LDS and are added in lane order -- a fixed order whatever the launch looks like
    __shared__ float bsum[16][17];
    const int c16 = (int)threadIdx.x & 15, rl = (int)threadIdx.x >> 4;
    const int ch = ((int)blockIdx.x - main_blocks) * 16 + c16;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (ch < cout) {
      int r = rl;
      for (; r + 48 < bias_rows; r += 64) {
        a0 += bpart[(size_t)r * cout_s + ch];
        a1 += bpart[(size_t)(r + 16) * cout_s + ch];
        a2 += bpart[(size_t)(r + 32) * cout_s + ch];
        a3 += bpart[(size_t)(r + 48) * cout_s + ch];
      }
      for (; r < bias_rows; r += 16) a0 += bpart[(size_t)r * cout_s + ch];
    }
    bsum[rl][c16] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && ch < cout) {
      float acc = 0.f;
#pragma unroll
      for (int l = 0; l < 16; ++l) acc += bsum[l][c16];
      dbias[ch] += acc;
    }
    return;
  }
  constexpr int EL = 256 / KL;
  __shared__ f32x4 red[KL][EL];
  const int tiles_n = tap_slots * ci_blocks * co_blocks;
  const int el = threadIdx.x % EL, kl = threadIdx.x / EL;
  const int e = blockIdx.x * EL + el;
  f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (e < tiles_n * 1024) {
    const size_t stride = (size_t)tiles_n * 1024;
#pragma unroll 4
    for (int sp = kl; sp < splits; sp += KL) {
      const f32x4 v = ws[(size_t)sp * stride + e];
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
  }
  if (KL > 1) {
    red[kl][el] = s;
    __syncthreads();
    if (kl != 0) return;
#pragma unroll
    for (int l = 1; l < KL; ++l) {
      const f32x4 v = red[l][el];
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
  }
  if (e >= tiles_n * 1024) return;
  int tile = e >> 10;
  const int frag = (e >> 6) & 15, lane = e & 63;
  const int cob = tile % co_blocks;
  tile /= co_blocks;
  const int cib = tile % ci_blocks, slot = tile / ci_blocks;
  int tap, ci;
  if (!wgrad_column(fold, cpt, tpt, taps, cin, slot, cib, (frag & 3) * 16 + (lane & 15), tap, ci)) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = cob * 64 + (frag >> 2) * 16 + 4 * (lane >> 4) + r;
    if (co >= cout) continue;
    dw[((size_t)co * cin + ci) * taps + tap] += s[r];
  }
}

// per-channel sum over pixels of an NHWC tensor (bias gradient): per-thread partial sums over a strided pixel subset,
// reduced across the block's pixel lanes in LDS, then one row of per-block sums to ``part`` (plain stores; the split-reduce
// kernel adds the rows in block order: run-to-run identical) -- or, without a workspace, ONE fp32 atomic per channel per block
template <typename T>
__global__ __launch_bounds__(256) void channel_sum_kernel(const uint16_t* __restrict__ x, float* __restrict__ out,
                                                          long npix, int cs, int c, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float sm_cs[];   // [ppb][groups * 8]
  const int groups = cs / 8;
  const int ppb = blockDim.x / groups;               // pixel lanes per block
  const int gi = threadIdx.x % groups, pl = threadIdx.x / groups;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (pl < ppb) {
#pragma unroll 4
    for (long pix = (long)blockIdx.x * ppb + pl; pix < npix; pix += (long)gridDim.x * ppb) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(x + pix * cs + gi * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a, b;
        unpack2<T>(v[e], a, b);
        s[2 * e] += a;
        s[2 * e + 1] += b;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) sm_cs[(pl * groups + gi) * 8 + e] = s[e];
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < groups * 8; ch += blockDim.x) {
    if (ch >= c) continue;
    float acc = 0.f;
    for (int l = 0; l < ppb; ++l) acc += sm_cs[l * groups * 8 + ch];
    if (part) part[(size_t)blockIdx.x * cs + ch] = acc;
    else atomicAdd(out + ch, acc);
  }
}

CGAN_KNOB(int, g_wgrad_target, 0);
CGAN_KNOB(int, g_wgrad_dbg, 0);
CGAN_KNOB(int, g_wgrad_coop_min_pix, 32768);
CGAN_KNOB(int, g_wgrad_coop_chunk, 0);
CGAN_KNOB(int, g_wgrad_slots, 512);
CGAN_KNOB(int, g_wgrad_tile, 1);
CGAN_KNOB(int, g_wgrad_coop_g, 0);          // dev: 0 = automatic, 2 = never the 16-wave 256 x 256 tile, 4 = wherever it applies
CGAN_KNOB(int, g_wgrad_bias_fused, 1);      // dev: 0 = the separate channel-sum pass for every bias gradient
CGAN_KNOB(int, g_wgrad_ws_cost_pct, 100);   // dev: the planner's cost of a partial tile through the workspace, in % of the fitted value
CGAN_KNOB(unsigned long long*, g_wgrad_ts, nullptr);
}  // namespace

#ifdef CGAN_DEV
extern "C" void cgan_debug_set_wgrad_tsbuf(void* p) { g_wgrad_ts = (unsigned long long*)p; }
extern "C" void cgan_debug_set_wgrad_bias_fused(int v) { g_wgrad_bias_fused = v; }
extern "C" void cgan_debug_set_wgrad_ws_cost(int pct) { g_wgrad_ws_cost_pct = pct > 0 ? pct : 100; }
extern "C" void cgan_debug_set_wgrad_tile3x3(int v) { g_wgrad_tile = v; }   // 0: never the spatially tiled 3 x 3 kernel, 2: wherever it applies
extern "C" void cgan_debug_set_wgrad_slots(int v) { g_wgrad_slots = v > 0 ? v : 512; }   // resident workgroups the planner assumes
extern "C" void cgan_debug_set_wgrad_coop_chunk(int v) { g_wgrad_coop_chunk = (v == 32 || v == 64) ? v : 0; }   // 0: automatic

extern "C" void cgan_debug_set_wgrad_coop_min_pixels(int v) { g_wgrad_coop_min_pix = v; }
extern "C" void cgan_debug_set_wgrad_coop_g(int v) { g_wgrad_coop_g = (v == 2 || v == 4) ? v : 0; }

extern "C" void cgan_debug_set_wgrad(int target_workgroups, int dbg) {
  g_wgrad_target = target_workgroups;          // > 0: target number of workgroups; < 0: -target pixel splits, as given
  g_wgrad_dbg = dbg;
}
#endif

constexpr int BIAS_MAX_BLOCKS = 512;      // rows of the separate channel-sum pass
constexpr int BIAS_MAX_ROWS = 1024;       // rows of bias partials the workspace holds (fused: one per pixel split [x wave])

// Tiling of one weight-gradient call: N tiles (tap slots x ci blocks), M tiles (co blocks), pixel splits.
struct WgradPlan {
  int fold, cpt, tpt, tap_slots, ci_blocks, co_blocks, splits;
  int coop, co_pairs, n_pairs, chunk;   // chunk: pixels per stage of the cooperative kernel (64 / 32)
  int g;                                // wave grid of the cooperative kernel: 2 (128 x 128 tile) or 4 (256 x 256, 16 waves)
  int tile;                             // the spatially tiled 3 x 3 kernel (conv_wgrad_tile3x3_kernel)
  long tiles() const { return (long)tap_slots * ci_blocks * co_blocks; }
};

static WgradPlan wgrad_plan(const CganConvDesc* d, bool have_ws = true) {
  WgradPlan pl;
  const int taps = d->kh * d->kw, cin_s = cgan_cs(d->c_in);
  pl.co_blocks = ceil_div(cgan_cs(d->c_out), 64);
  // cin_s <= 32 with several taps: fold taps into the 64-wide N tile (cpt 16-byte channel chunks per tap, tpt taps per
  // tile) instead of padding every tap's few channels to 64
  pl.fold = (cin_s <= 32 && taps > 1 && g_wgrad_dbg != 4) ? 1 : 0;
  pl.cpt = pl.fold ? cin_s / 8 : 8;
  pl.tpt = pl.fold ? 8 / pl.cpt : 1;
  pl.tap_slots = pl.fold ? ceil_div(taps, pl.tpt) : taps;
  pl.ci_blocks = pl.fold ? 1 : ceil_div(cin_s, 64);
  const long npix = (long)d->n * d->h_out * d->w_out;
  // cooperative 128 x 128 kernel: needs 2 co blocks and 2 N tiles to pair, rows of whole 8-pixel pieces, and either so many pixels that 4x the splits still leaves long pixel ranges or so many tiles that the
  // splits stay few (the partial-tile workspace grows with the splits)
  pl.g = 2;
  pl.co_pairs = (pl.co_blocks + 1) / 2;
  pl.n_pairs = pl.fold ? (pl.tap_slots + 1) / 2 : pl.tap_slots * ((pl.ci_blocks + 1) / 2);
  // (odd block counts would leave a quarter of the quadrants idle: 128 -> 160 at 4 x 320^2 is 6 % slower that way)
  pl.coop = ((pl.co_blocks % 2) == 0 && ((pl.fold ? pl.tap_slots : pl.ci_blocks) % 2) == 0 && (d->w_out % 8) == 0 &&
             (g_wgrad_dbg & 8) == 0 && (double)npix * cgan_cs(d->c_out) * 2.0 < 1.9e9 &&
             (npix >= (long)g_wgrad_coop_min_pix || pl.tiles() >= 256))
                ? 1 : 0;
  // spatially tiled 3 x 3 kernel: large maps of whole 4 x 32 pixel tiles, 64-channel input blocks, partial tiles through
  // the workspace only; where the cooperative kernel cannot pair blocks (SPADE gamma|beta 128 -> 40 at 4 x 640^2: 438 ->
  // 287 us, 128 -> 160 at 4 x 320^2: 292 -> 217 us; with paired blocks the cooperative kernel is as fast: 128 -> 80 548
  // against 586 us).  Knob 2 forces it wherever it applies (tests).
  pl.tile = ((g_wgrad_tile == 2 || (g_wgrad_tile == 1 && !pl.coop)) && have_ws && !pl.fold && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 &&
             d->dilation == 1 && d->pad_mode == CGAN_PAD_ZERO && !d->in_upsample && (d->w_out % TL_W) == 0 &&
             (d->h_out % TL_H) == 0 && (cin_s % 64) == 0 && npix >= 65536 && (g_wgrad_dbg & (8 | 16)) == 0) ? 1 : 0;
  if (pl.tile) {
    pl.coop = 0;
    pl.chunk = TL_W * TL_H;
    const long pairs = (long)pl.co_blocks * pl.ci_blocks, ntiles = npix / (TL_W * TL_H);
    long sp = g_wgrad_target < 0 ? -g_wgrad_target : (long)g_wgrad_slots / pairs;
    if (sp > ntiles) sp = ntiles;
    pl.splits = sp < 1 ? 1 : (int)sp;
    return pl;
  }
  const long tiles = pl.coop ? (long)pl.co_pairs * pl.n_pairs : pl.tiles();
  // Pixel splits.  The single-wave kernel and the cooperative kernel with 64-pixel stages keep 64 KiB of LDS per
  // workgroup, so 2 workgroups x 256 CUs = 512 run at a time and equal workgroups finish together: a grid of 513 takes
  // twice as long as one of 512 (tools/sweep_wgrad_splits.py: SPADE gamma|beta gradient 128 -> 80 at 4 x 640^2, 56 splits
  // = 504 workgroups 623 us, 57 splits 1101 us; the first planner aimed at "about 1024 / 2048 workgroups" and landed just
  // past a multiple as often as not: layer3 3x3 118 -> 84 us, layer2 3x3 61 -> 32 us with the splits below).
  // Candidates: the largest split counts that still fit m rounds; cost in chunk times = rounds x (chunks per workgroup
  // + ~2 for prologue / tile store) + workgroups x (partial tile written to and read back from the workspace), fitted
  // to that sweep.
  auto plan_splits = [&](int chunk, long slots, int& nchunks_out) {
    const int nchunks = (int)((npix + chunk - 1) / chunk);
    nchunks_out = nchunks;
    long sp_best = 1;
    if (g_wgrad_target > 0) {
      sp_best = (g_wgrad_target + tiles - 1) / tiles;
    } else if (g_wgrad_target < 0) {
      sp_best = -g_wgrad_target;
    } else {
      const double t_ws = (pl.coop ? 0.024 : 0.008) * (g_wgrad_ws_cost_pct / 100.0);
      double best = 1e30;
      for (int m = 0; m <= 8; ++m) {
        long sp = m == 0 ? 1 : (slots * m) / tiles;
        if (sp < 1) continue;
        if (sp > nchunks) sp = nchunks;
        const long wgs = sp * tiles;
        const double cost = (double)((wgs + slots - 1) / slots) * ((double)((nchunks + sp - 1) / sp) + 2.0) + (double)wgs * t_ws;
        if (cost < best) { best = cost; sp_best = sp; }
      }
    }
    if (sp_best > nchunks) sp_best = nchunks;
    return sp_best < 1 ? 1L : sp_best;
  };
  int nchunks = 0;
  pl.chunk = pl.coop ? 64 : 128;
  // 16-wave 256 x 256 tile (round 5): block counts divisible by four on both sides; one workgroup per CU, so the splits are
  // sized to 256 slots; taken where every workgroup still walks a handful of chunks (its 256 KiB partial tile and its
  // first fill are not amortised below that)
  if (pl.coop && g_wgrad_coop_g != 2 && (pl.co_blocks % 4) == 0 && ((pl.fold ? pl.tap_slots : pl.ci_blocks) % 4) == 0 &&
      g_wgrad_coop_chunk == 0) {
    const long tiles4 = (long)(pl.co_blocks / 4) * (pl.fold ? pl.tap_slots / 4 : (long)pl.tap_slots * (pl.ci_blocks / 4));
    const int nch = (int)((npix + 63) / 64);
    long sp = g_wgrad_target < 0 ? -g_wgrad_target : (g_wgrad_slots / 2) / tiles4;
    if (sp < 1) sp = 1;
    if (sp > nch) sp = nch;
    if (g_wgrad_coop_g == 4 || (tiles4 <= g_wgrad_slots / 2 && nch / sp >= 8)) {
      pl.g = 4;
      pl.co_pairs = pl.co_blocks / 4;
      pl.n_pairs = pl.fold ? pl.tap_slots / 4 : pl.tap_slots * (pl.ci_blocks / 4);
      pl.chunk = 64;
      pl.splits = (int)sp;
      return pl;
    }
  }
  long splits = plan_splits(pl.chunk, g_wgrad_slots, nchunks);
  // cooperative kernel with 32-pixel stages (32 KiB of LDS, 4 workgroups per CU = 1024 at a time): pays on long pixel
  // ranges (ASPP 3x3 2048 -> 256: 612 -> 583 us, SPADE gamma|beta 128 -> 80 at 4 x 640^2: 586 -> 550 us, layer4) and
  // costs on short ones (layer3 1x1: 45 -> 53 us, layer2 3x3: 31 -> 40 us): taken from ~100 64-pixel chunks per
  // workgroup on
  if (pl.coop && (g_wgrad_coop_chunk == 32 || (g_wgrad_coop_chunk == 0 && nchunks / splits >= 96))) {
    pl.chunk = 32;
    splits = plan_splits(32, 2 * g_wgrad_slots, nchunks);
  }
  if (splits > nchunks) splits = nchunks;
  pl.splits = splits < 1 ? 1 : (int)splits;
  return pl;
}

extern "C" size_t cgan_conv2d_bwd_weight_workspace_bytes(const CganConvDesc* d) {
  if (!d || d->n <= 0 || d->h_out <= 0 || d->w_out <= 0 || d->c_in <= 0 || d->c_out <= 0 || d->kh <= 0 || d->kw <= 0)
    return 0;
  const WgradPlan pl = wgrad_plan(d);
  // partial tiles | up to BIAS_MAX_BLOCKS rows of per-block channel sums (the bias gradient's deterministic reduction)
  return (size_t)pl.tiles() * (size_t)pl.splits * 64 * 64 * sizeof(float) + (size_t)BIAS_MAX_ROWS * cgan_cs(d->c_out) * sizeof(float);
}

extern "C" int cgan_conv2d_nhwc_bwd_weight(const void* x, const void* dy, float* dw_oihw, float* dbias,
                                           const CganConvDesc* d, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  CGAN_REQUIRE(d != nullptr && x && dy && dw_oihw, "conv2d_nhwc_bwd_weight: null pointer");
  CGAN_REQUIRE(d->dtype == CGAN_F16 || d->dtype == CGAN_BF16, "conv2d_nhwc_bwd_weight: bad dtype %d", d->dtype);
  CGAN_REQUIRE(d->pad_mode == CGAN_PAD_ZERO || d->pad_mode == CGAN_PAD_REFLECT, "conv2d_nhwc_bwd_weight: bad pad mode");
  if (d->pad_mode == CGAN_PAD_REFLECT)
    CGAN_REQUIRE(d->pad < d->h_in && d->pad < d->w_in && !d->in_upsample,
                 "conv2d_nhwc_bwd_weight: reflect padding must be < input size (and is not combined with in_upsample)");
  if (d->in_upsample) CGAN_REQUIRE((d->h_in % 2) == 0 && (d->w_in % 2) == 0, "conv2d_nhwc_bwd_weight: in_upsample needs even h_in/w_in");
  CGAN_REQUIRE(d->n > 0 && d->h_in > 0 && d->w_in > 0 && d->c_in > 0 && d->c_out > 0 && d->kh > 0 && d->kw > 0 &&
                   d->stride > 0 && d->dilation > 0 && d->pad >= 0,
               "conv2d_nhwc_bwd_weight: bad shape");
  const int eh = (d->h_in + 2 * d->pad - d->dilation * (d->kh - 1) - 1) / d->stride + 1;
  const int ew = (d->w_in + 2 * d->pad - d->dilation * (d->kw - 1) - 1) / d->stride + 1;
  CGAN_REQUIRE(eh == d->h_out && ew == d->w_out, "conv2d_nhwc_bwd_weight: h_out/w_out inconsistent");
  WgradArgs a;
  a.x = (const uint16_t*)x; a.dy = (const uint16_t*)dy; a.dw = dw_oihw;
  a.n = d->n; a.h_in = d->h_in; a.w_in = d->w_in; a.cin = d->c_in; a.cin_s = cgan_cs(d->c_in);
  a.cout = d->c_out; a.cout_s = cgan_cs(d->c_out);
  a.kh = d->kh; a.kw = d->kw; a.stride = d->stride; a.pad = d->pad; a.dil = d->dilation;
  a.h_out = d->h_out; a.w_out = d->w_out;
  const long npix = (long)d->n * d->h_out * d->w_out;
  CGAN_REQUIRE(npix < (1L << 31) - 256, "conv2d_nhwc_bwd_weight: too many pixels");
  a.npix = (int)npix;
  a.x_ups = d->in_upsample;
  a.reflect = d->pad_mode == CGAN_PAD_REFLECT;
  a.nchunks = ceil_div(a.npix, 128);
  const WgradPlan pl = wgrad_plan(d, workspace != nullptr);
  a.ci_blocks = pl.ci_blocks; a.co_blocks = pl.co_blocks; a.splits = pl.splits;
  a.fold = pl.fold; a.cpt = pl.cpt; a.tpt = pl.tpt; a.tap_slots = pl.tap_slots;
  a.coop = pl.coop; a.co_pairs = pl.co_pairs; a.n_pairs = pl.n_pairs;
  const int taps = d->kh * d->kw;
  CGAN_DEV_ONLY(a.dbg = g_wgrad_dbg; a.ts = g_wgrad_ts;)
  const long items = (long)a.splits * (pl.tile ? (long)pl.co_blocks * pl.ci_blocks
                                              : (pl.coop ? (long)pl.co_pairs * pl.n_pairs : pl.tiles()));
  a.per_xcd = (int)((items + 7) / 8);
  CGAN_REQUIRE(items < (1L << 30), "conv2d_nhwc_bwd_weight: grid too large");
  const unsigned gx = (unsigned)a.per_xcd * 8;
  a.ws = nullptr;
  a.bpart = nullptr;
  int bias_rows = 0;
  if (workspace) {
    CGAN_REQUIRE(workspace_bytes >= cgan_conv2d_bwd_weight_workspace_bytes(d),
                 "conv2d_nhwc_bwd_weight: workspace too small (%zu bytes)", workspace_bytes);
    a.ws = (float*)workspace;
    // the bias gradient inside the weight-gradient kernel (a constant-one GEMM column in the workgroups of N tile 0): one
    // partial row per pixel split (x 4 waves in the single-wave-tile kernel), as long as the rows fit; otherwise, and
    // without a workspace, the separate channel-sum pass below
    const int rows = a.splits * ((pl.tile || pl.coop) ? 1 : 4);
    // (not in the two variants without registers to spare: the cooperative kernel's 32-pixel stages, the tiled kernel with
    // four co tiles)
    const bool variant_ok = pl.tile ? (pl.co_blocks == 1 && ceil_div(cgan_cs(d->c_out), 16) < 4)
                                    : (!pl.coop || (pl.chunk == 64 && pl.g == 2));
    if (dbias && rows <= BIAS_MAX_ROWS && variant_ok && g_wgrad_bias_fused && CGAN_WTS(a) == nullptr) {
      a.bpart = a.ws + (size_t)pl.tiles() * (size_t)pl.splits * 64 * 64;
      bias_rows = rows;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t smem = 4 * WAVE_LDS;   // 72 KiB: also holds the 4 x 16 KiB partial tiles of the final reduction
  // 32-bit byte offsets inside the kernel
  CGAN_REQUIRE((double)d->n * (a.x_ups ? d->h_in / 2 : d->h_in) * (a.x_ups ? d->w_in / 2 : d->w_in) * a.cin_s * 2.0 < 4294967295.0 &&
                   (double)a.npix * a.cout_s * 2.0 < 4294967295.0,
               "conv2d_nhwc_bwd_weight: activation tensors of 4 GiB or more are not supported");
  a.x_bytes = (unsigned)((size_t)d->n * (a.x_ups ? d->h_in / 2 : d->h_in) * (a.x_ups ? d->w_in / 2 : d->w_in) * a.cin_s * 2);
  a.dy_bytes = (unsigned)((size_t)a.npix * a.cout_s * 2);
  const int mode = a.x_ups ? 1 : (a.reflect ? 2 : 0);
  // uniform-piece addressing: pieces must not straddle output rows, and the "always out of range" lane constant needs
  // the dy offsets (one chunk stride past the end included) below 2^31
  const bool uni = (d->w_out % 8) == 0 && g_wgrad_dbg != 16 &&
                   ((double)a.npix + (double)a.splits * 128.0 + 128.0) * a.cout_s * 2.0 < 2147483648.0;
// more than 64 KiB of dynamic LDS needs the function attribute, once per kernel instance and device
#define CGAN_BIG_LDS(KERNEL)                                                                                           \
  do {                                                                                                                 \
    static unsigned long long done_mask = 0;                                                                           \
    int dev_ = 0;                                                                                                      \
    (void)hipGetDevice(&dev_);                                                                                         \
    if (!((done_mask >> (dev_ & 63)) & 1ull)) {                                                                        \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&KERNEL),                                      \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                     \
      if (e_ != hipSuccess) {                                                                                          \
        cgan_set_error("conv2d_nhwc_bwd_weight: hipFuncSetAttribute failed: %s", hipGetErrorString(e_));               \
        return CGAN_ERR_HIP;                                                                                           \
      }                                                                                                                \
      done_mask |= 1ull << (dev_ & 63);                                                                                \
    }                                                                                                                  \
  } while (0)
#define WGRAD_LAUNCH(TT, MM, UU) do { CGAN_BIG_LDS((conv_wgrad_kernel<TT, MM, UU>)); hipLaunchKernelGGL((conv_wgrad_kernel<TT, MM, UU>), dim3(gx), dim3(256), smem, s, a); } while (0)
#define WGRAD_MODE(TT)                                                          \
  do {                                                                          \
    if (mode == 2) { if (uni) WGRAD_LAUNCH(TT, 2, true); else WGRAD_LAUNCH(TT, 2, false); }     \
    else if (mode == 1) { if (uni) WGRAD_LAUNCH(TT, 1, true); else WGRAD_LAUNCH(TT, 1, false); } \
    else { if (uni) WGRAD_LAUNCH(TT, 0, true); else WGRAD_LAUNCH(TT, 0, false); }                \
  } while (0)
  if (pl.tile) {
    const size_t smem3 = TL_XH_BYTES + TL_DY_BYTES;
    // a single co block of <= 48 channels: only the co tiles that exist (the 16-row tiles past cout_s would multiply zeros)
    const int na = pl.co_blocks == 1 ? ceil_div(a.cout_s, 16) : 4;
#define TILE_LAUNCH(TT, NN, BB) hipLaunchKernelGGL((conv_wgrad_tile3x3_kernel<TT, NN, BB>), dim3(gx), dim3(256), smem3, s, a)
#define TILE_NB(TT, NN) do { if (a.bpart) TILE_LAUNCH(TT, NN, true); else TILE_LAUNCH(TT, NN, false); } while (0)
#define TILE_NA(TT) do { if (na == 1) TILE_NB(TT, 1); else if (na == 2) TILE_NB(TT, 2); else if (na == 3) TILE_NB(TT, 3); else TILE_LAUNCH(TT, 4, false); } while (0)
    if (d->dtype == CGAN_F16) TILE_NA(F16); else TILE_NA(BF16);
#undef TILE_NA
#undef TILE_NB
#undef TILE_LAUNCH
  } else if (pl.coop && pl.g == 4) {
    const size_t smem4 = (size_t)2 * 8 * 8 * PIECE;      // two stages of 4 dy + 4 x sub-slabs of 8 padded pieces: 144 KiB
#define COOP4_LAUNCH(TT, MM)                                                                                           \
  do {                                                                                                                 \
    CGAN_BIG_LDS((conv_wgrad_coop_kernel<TT, MM, 64, false, false, 4>));                                               \
    hipLaunchKernelGGL((conv_wgrad_coop_kernel<TT, MM, 64, false, false, 4>), dim3(gx), dim3(1024), smem4, s, a);      \
  } while (0)
#define COOP4_MODE(TT) do { if (mode == 1) COOP4_LAUNCH(TT, 1); else if (mode == 2) COOP4_LAUNCH(TT, 2); else COOP4_LAUNCH(TT, 0); } while (0)
#ifdef CGAN_DEV
    if (CGAN_WTS(a) && d->dtype == CGAN_BF16 && mode == 0) {        // tools/ts_wgrad.py: the phase stamps of the 16-wave tile
      CGAN_BIG_LDS((conv_wgrad_coop_kernel<BF16, 0, 64, true, false, 4>));
      hipLaunchKernelGGL((conv_wgrad_coop_kernel<BF16, 0, 64, true, false, 4>), dim3(gx), dim3(1024), smem4, s, a);
    } else
#endif
    if (d->dtype == CGAN_F16) COOP4_MODE(F16);
    else COOP4_MODE(BF16);
#undef COOP4_MODE
#undef COOP4_LAUNCH
  } else if (pl.coop) {
    const size_t smem2 = (size_t)2 * 4 * (pl.chunk / 8) * PIECE;      // 72 KiB (64-pixel stages) / 36 KiB (32-pixel)
#define COOP_LAUNCH(TT, MM, CC, BB) do { CGAN_BIG_LDS((conv_wgrad_coop_kernel<TT, MM, CC, false, BB>)); hipLaunchKernelGGL((conv_wgrad_coop_kernel<TT, MM, CC, false, BB>), dim3(gx), dim3(256), smem2, s, a); } while (0)
#define COOP_MODE(TT)                                                                       \
  do {                                                                                      \
    if (pl.chunk == 32) { if (mode == 1) COOP_LAUNCH(TT, 1, 32, false); else if (mode == 2) COOP_LAUNCH(TT, 2, 32, false); else COOP_LAUNCH(TT, 0, 32, false); } \
    else if (a.bpart) { if (mode == 1) COOP_LAUNCH(TT, 1, 64, true); else if (mode == 2) COOP_LAUNCH(TT, 2, 64, true); else COOP_LAUNCH(TT, 0, 64, true); }            \
    else { if (mode == 1) COOP_LAUNCH(TT, 1, 64, false); else if (mode == 2) COOP_LAUNCH(TT, 2, 64, false); else COOP_LAUNCH(TT, 0, 64, false); }            \
  } while (0)
#ifdef CGAN_DEV
    if (CGAN_WTS(a) && d->dtype == CGAN_BF16 && mode == 0) {
      if (pl.chunk == 32) hipLaunchKernelGGL((conv_wgrad_coop_kernel<BF16, 0, 32, true>), dim3(gx), dim3(256), smem2, s, a);
      else { CGAN_BIG_LDS((conv_wgrad_coop_kernel<BF16, 0, 64, true>)); hipLaunchKernelGGL((conv_wgrad_coop_kernel<BF16, 0, 64, true>), dim3(gx), dim3(256), smem2, s, a); }
    } else
#endif
    if (d->dtype == CGAN_F16) COOP_MODE(F16);
    else COOP_MODE(BF16);
#undef COOP_MODE
#undef COOP_LAUNCH
  } else if (d->dtype == CGAN_F16) WGRAD_MODE(F16); else WGRAD_MODE(BF16);
#undef WGRAD_MODE
#undef WGRAD_LAUNCH
  CGAN_CHECK_LAUNCH("conv2d_nhwc_bwd_weight");
  float* bpart = a.bpart;
  if (dbias && !a.bpart) {
    const int cs = a.cout_s;
    const int threads = 256;
    const int ppb = threads / (cs / 8) > 0 ? threads / (cs / 8) : 1;
    CGAN_REQUIRE(cs / 8 <= threads, "conv2d_nhwc_bwd_weight: too many channels for the bias reduction");
    long want = (npix + (long)ppb * 16 - 1) / ((long)ppb * 16);
    const int grid = (int)(want < 1 ? 1 : (want > BIAS_MAX_BLOCKS ? BIAS_MAX_BLOCKS : want));
    const size_t smem_b = (size_t)ppb * (cs / 8) * 8 * sizeof(float);
    if (a.ws) {       // per-block rows behind the partial tiles, summed in order by the reduce kernel below
      bpart = a.ws + (size_t)pl.tiles() * (size_t)pl.splits * 64 * 64;
      bias_rows = grid;
    }
    if (d->dtype == CGAN_F16)
      hipLaunchKernelGGL(channel_sum_kernel<F16>, dim3(grid), dim3(threads), smem_b, s, (const uint16_t*)dy, dbias, npix,
                         cs, d->c_out, bpart);
    else
      hipLaunchKernelGGL(channel_sum_kernel<BF16>, dim3(grid), dim3(threads), smem_b, s, (const uint16_t*)dy, dbias, npix,
                         cs, d->c_out, bpart);
    CGAN_CHECK_LAUNCH("conv2d_nhwc_bwd_weight(bias)");
  }
  if (a.ws) {
    const int elems = (int)pl.tiles() * 1024;
    const int bias_blocks = bpart ? ceil_div(d->c_out, 16) : 0;
    if (a.splits <= 4) {
      const int mb = ceil_div(elems, 256);
      hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3(mb + bias_blocks), dim3(256), 0, s, (const f32x4*)a.ws, a.dw,
                         a.splits, taps, a.tap_slots, a.ci_blocks, a.co_blocks, a.cout, a.cin, a.fold, a.cpt, a.tpt, mb,
                         (const float*)bpart, bias_rows, a.cout_s, dbias);
    } else {
      const int mb = ceil_div(elems, 32);
      hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3(mb + bias_blocks), dim3(256), 0, s, (const f32x4*)a.ws, a.dw,
                         a.splits, taps, a.tap_slots, a.ci_blocks, a.co_blocks, a.cout, a.cin, a.fold, a.cpt, a.tpt, mb,
                         (const float*)bpart, bias_rows, a.cout_s, dbias);
    }
    CGAN_CHECK_LAUNCH("conv2d_nhwc_bwd_weight(reduce)");
  }
  return CGAN_OK;
}
