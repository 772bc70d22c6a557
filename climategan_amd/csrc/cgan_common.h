// Shared device/host helpers for libcgan_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "climategan_hip.h"

// ------------------------------------------------------------------------------------------------
// error reporting (host)
// ------------------------------------------------------------------------------------------------
void cgan_set_error(const char* fmt, ...);

#define CGAN_REQUIRE(cond, ...)            \
  do {                                     \
    if (!(cond)) {                         \
      cgan_set_error(__VA_ARGS__);         \
      return CGAN_ERR_BAD_ARG;             \
    }                                      \
  } while (0)

#define CGAN_CHECK_LAUNCH(name)                                                    \
  do {                                                                             \
    hipError_t e__ = hipGetLastError();                                            \
    if (e__ != hipSuccess) {                                                       \
      cgan_set_error("%s: HIP launch failed: %s", name, hipGetErrorString(e__));   \
      return CGAN_ERR_HIP;                                                         \
    }                                                                              \
  } while (0)

// ------------------------------------------------------------------------------------------------
// the conv epilogues' "residual" slot
// ------------------------------------------------------------------------------------------------
// has_res = 1: v += r (residual add; the second gradient contribution of a data-gradient conv).  has_res = 2: v = r > 0 ? v : 0
// -- the ReLU derivative taken from the activation's OUTPUT r, for a data-gradient conv whose result is the gradient of a
// ReLU's output (cgan_conv2d_nhwc_bwd_data_relu): the separate act_bwd pass (two reads + one write of the map) disappears.
__device__ __forceinline__ float cgan_res_apply(float v, float r, int mode) { return mode == 2 ? (r > 0.f ? v : 0.f) : v + r; }
// has_res = 3 (round 6): both at once, from two tensors -- v = m > 0 ? v + r : 0 (cgan_conv2d_nhwc_bwd_data_add_relu: the
// data gradient of a bottleneck's first conv + the skip branch's gradient, times the derivative of the ReLU whose output the
// conv read: what BatchNorm's backward of the previous block would otherwise form in a pass of its own).  Only the kernels
// with the shared store path of conv_gemm.h take it (conv_gemm_res2_ok); ``res2`` carries m.
__device__ __forceinline__ float cgan_res_apply3(float v, float r, float m) { return m > 0.f ? v + r : 0.f; }

// ------------------------------------------------------------------------------------------------
// development knobs
// ------------------------------------------------------------------------------------------------
// Kernel-selection / ablation / in-kernel-timestamp knobs (cgan_debug_set_*) exist only in the CGAN_DEV build
// (libcgan_hip_dev.so: `make dev`; tools/ and the tests that run every kernel variant on the same cases load that one).
// In the product library they are compile-time constants: no process-global mutable state behind the ABI, no debug
// fields in the kernels' parameter structs, the branches on them fold away.
#ifdef CGAN_DEV
#define CGAN_KNOB(type, name, init) type name = init
#define CGAN_DEV_ONLY(...) __VA_ARGS__
#define CGAN_DBG(p) ((p).dbg)
#define CGAN_TSBUF(p) ((p).tsbuf)
#else
#define CGAN_KNOB(type, name, init) [[maybe_unused]] constexpr type name = init
#define CGAN_DEV_ONLY(...)
#define CGAN_DBG(p) 0
#define CGAN_TSBUF(p) ((unsigned long long*)nullptr)
#endif

// ------------------------------------------------------------------------------------------------
// streaming accesses
// ------------------------------------------------------------------------------------------------
// The element-wise / normalisation passes touch every byte of a map once per launch and the maps of the train step are far
// larger than L2 and the memory-side cache: their loads and stores carry the non-temporal hint (global_load / _store ... nt),
// round 6, last session.  CGAN_NO_NT (build flag): plain accesses (same-box A/B).
#ifndef CGAN_NO_NT
#define CGAN_LD_STREAM(p) __builtin_nontemporal_load(p)
#define CGAN_ST_STREAM(v, p) __builtin_nontemporal_store(v, p)
#else
#define CGAN_LD_STREAM(p) (*(p))
#define CGAN_ST_STREAM(v, p) (*(p) = (v))
#endif

// ------------------------------------------------------------------------------------------------
// 16-bit element types and MFMA wrappers
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct F16 {
  using scalar = _Float16;
  using vec8 = f16x8;
  static constexpr int id = CGAN_F16;
};
struct BF16 {
  using scalar = __bf16;
  using vec8 = bf16x8;
  static constexpr int id = CGAN_BF16;
};

// ------------------------------------------------------------------------------------------------
// split-precision maps (csrc/pair.hip, cgan_conv2d_nhwc_fwd_pair)
// ------------------------------------------------------------------------------------------------
// A value is carried as NC 16-bit components (v = c0 + c1 [+ c2], each the rounding of what the previous ones left).  A conv
// multiplies NB K-blocks per pixel, K-block b being component xcomp(b) of the input against component wcomp(b) of the weights
// along their input channels, so that ONE ordinary K loop accumulates every product of (sum x_i)(sum w_j) down to the type's
// precision floor.  A map STORES each component once: NS = NC channel blocks per pixel (round 6; rounds 4-5 stored all NB
// K-blocks, i.e. every component up to three times: twice the bytes per activation) -- the conv kernels read K-block b from
// storage block xcomp(b):
//   fp16 (11-bit mantissa, narrow exponent): NC = 2, blocks (c0 | c1 | c0) x (w0 | w0 | w1): 22 bits where the low part
//        stays a normal number (|v| >~ 0.1), an absolute floor of 2^-24 below;
//   bf16 (8-bit mantissa, fp32's exponent):  NC = 3, blocks (c0 | c1 | c2 | c0 | c1 | c0) x (w0 | w0 | w0 | w1 | w1 | w2):
//        every product down to 2^-16 of the leading one, 24 bits at any magnitude = the fp32 reference's arithmetic.
template <typename T> struct Split;
template <> struct Split<F16> {
  static constexpr int NC = 2, NB = 3, NS = 2;
  __host__ __device__ static constexpr int xcomp(int b) { return b == 1 ? 1 : 0; }
  __host__ __device__ static constexpr int wcomp(int b) { return b == 2 ? 1 : 0; }
};
template <> struct Split<BF16> {
  static constexpr int NC = 3, NB = 6, NS = 3;
  __host__ __device__ static constexpr int xcomp(int b) { return b == 1 || b == 4 ? 1 : (b == 2 ? 2 : 0); }
  __host__ __device__ static constexpr int wcomp(int b) { return b < 3 ? 0 : (b < 5 ? 1 : 2); }
};
static inline int cgan_split_blocks(int dtype) { return dtype == CGAN_BF16 ? 6 : 3; }      // K blocks a conv multiplies
static inline int cgan_split_store_blocks(int dtype) { return dtype == CGAN_BF16 ? 3 : 2; }  // blocks a map STORES (round 6)

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ float to_f32(typename T::scalar v) {
  return (float)v;
}
template <typename T>
__device__ __forceinline__ typename T::scalar from_f32(float v) {
  return (typename T::scalar)v;
}
template <typename T>
__device__ __forceinline__ uint16_t bits_of(float v) {
  typename T::scalar s = (typename T::scalar)v;
  return __builtin_bit_cast(uint16_t, s);
}
template <typename T>
__device__ __forceinline__ float f32_of_bits(uint16_t b) {
  return (float)__builtin_bit_cast(typename T::scalar, b);
}
// two fp32 -> one packed pair of T, as ONE vector conversion (gfx950: v_cvt_pk_bf16_f32 / v_cvt_pkrtz-free
// v_cvt_pk_f16_f32, round to nearest even like the scalar casts): written as two scalar conversions + shift + or the
// compiler emitted four instructions per pair in every conv / norm epilogue
template <typename T>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef typename T::scalar s2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, s2));
}
template <typename T>
__device__ __forceinline__ void unpack2(uint32_t p, float& lo, float& hi) {
  lo = f32_of_bits<T>((uint16_t)(p & 0xffffu));
  hi = f32_of_bits<T>((uint16_t)(p >> 16));
}
template <typename T>
__device__ __forceinline__ typename T::vec8 as_vec8(u32x4 v) {
  return __builtin_bit_cast(typename T::vec8, v);
}

// NONE / ReLU / LeakyReLU share one select with a (uniform) negative-side factor: a per-value switch over the five
// activations compiled to ~4 scalar branches per VALUE in the conv epilogues (hundreds per workgroup, a visible share of
// the short-K 1x1 layers).  `+ 0.f` keeps ReLU's negative side at +0 (v * 0 alone would be -0).
static_assert(CGAN_ACT_NONE == 0 && CGAN_ACT_RELU == 1 && CGAN_ACT_LRELU == 2, "act_apply relies on the enum order");
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  if (act <= CGAN_ACT_LRELU) {
    const float ns = act == CGAN_ACT_NONE ? 1.f : (act == CGAN_ACT_RELU ? 0.f : slope);
    return v > 0.f ? v : v * ns + 0.f;
  }
  return act == CGAN_ACT_TANH ? tanhf(v) : 1.f / (1.f + __expf(-v));
}

// the same on N values with the activation decided once per wave (uniform branches): NONE costs nothing; ReLU and LeakyReLU
// with 0 <= slope <= 1 a fused multiply-add and a max per value -- max(v, slope v + 0) is the select's value for every v:
// NaN propagates (NaN * s + 0 = NaN), the negative side of ReLU is +0 as in act_apply; everything else the general form
template <int N>
__device__ __forceinline__ void act_apply_n(float (&v)[N], int act, float slope) {
  if (act == CGAN_ACT_NONE) return;
  const float ns = act == CGAN_ACT_RELU ? 0.f : slope;
  if (act <= CGAN_ACT_LRELU && ns >= 0.f && ns <= 1.f) {
#pragma unroll
    for (int r = 0; r < N; ++r) v[r] = fmaxf(v[r], __builtin_fmaf(v[r], ns, 0.f));
  } else {
#pragma unroll
    for (int r = 0; r < N; ++r) v[r] = act_apply(v[r], act, slope);
  }
}

// F.interpolate(mode="nearest") legacy source index: min(floor(dst * scale), in - 1), scale = in / out in f32
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
