// Development aid (tools/check_pk_opsel_concurrent.py): a kernel made of the packed-fp32 instruction pair the vectoriser
// formed in the first version of the wildfire blur -- v_pk_mul_f32 with one half of source 0 broadcast (op_sel_hi:[0,1]) and
// v_pk_add_f32 with crossed halves -- on register pairs whose other half holds junk, repeated `iters` times, so that its
// result can be compared alone and next to another stream's MFMA / LDS-DMA kernels (R5 DESIGN 4.6).
#include "cgan_common.h"

#ifdef CGAN_DEV      // dev build only (libcgan_hip_dev.so): the product library has no development entry point

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void pk_opsel_kernel(const float* __restrict__ x, const unsigned* __restrict__ junk,
                                                       const float* __restrict__ taps, float* __restrict__ out, long n, int iters, int mode) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    f32x2 acc = {0.f, 0.f};
    for (int k = 0; k < iters; ++k) {
      const long j = (i + 977L * k) % n;
      f32x2 src0 = {x[j], __builtin_bit_cast(float, junk[j])};
      const f32x2 t = {taps[2 * (k & 255)], taps[2 * (k & 255) + 1]};
      f32x2 prod;
      if (mode == 0) {            // the blur's pair: broadcast half, crossed halves
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(prod) : "v"(src0), "v"(t));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(acc) : "v"(acc), "v"(prod));
      } else if (mode == 1) {     // plain packed operations on fully defined pairs (what the statistics epilogues use)
        const f32x2 both = {src0[0], src0[0]};
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(prod) : "v"(both), "v"(t));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc) : "v"(acc), "v"(prod));
      } else {                    // the same arithmetic with scalar instructions
        float p0, p1, a0 = acc[0], a1 = acc[1];
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(src0[0]), "v"(t[0]));
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(src0[0]), "v"(t[1]));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a0) : "v"(a0), "v"(p0));
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(a1) : "v"(a1), "v"(p1));
        acc = (f32x2){a0, a1};
      }
    }
    out[2 * i] = acc[0];
    out[2 * i + 1] = acc[1];
  }
}
}  // namespace

extern "C" int cgan_debug_pk_opsel(const float* x, const unsigned* junk, const float* taps, float* out, int64_t n,
                                   int32_t iters, int32_t mode, void* stream) {
  CGAN_REQUIRE(x && junk && taps && out && n > 0 && iters > 0, "debug_pk_opsel: bad arguments");
  hipLaunchKernelGGL(pk_opsel_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, x, junk, taps, out, (long)n, iters, mode);
  CGAN_CHECK_LAUNCH("debug_pk_opsel");
  return CGAN_OK;
}
#endif  // CGAN_DEV
