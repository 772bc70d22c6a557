// Micro-test (GPU box): does v_pk_mul_f32 / v_pk_add_f32 with op_sel broadcasting ONE half of a source pair give a result
// that is independent of the other half's content on gfx950?  (Round 3: compiler-formed packed operations on pairs with an
// undefined half made the wildfire blur depend on other kernels' register leftovers, R5 DESIGN 4.6.)
//   build: hipcc --offload-arch=gfx950 -O2 tools/micro/pk_opsel.hip -o gpurun_out/pk_opsel   run: gpurun_out/pk_opsel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void k(const float* x, const unsigned* junk, const float* taps, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  f32x2 src0 = {x[i], __builtin_bit_cast(float, junk[i])};        // {value, whatever}
  const f32x2 t = {taps[2 * i], taps[2 * i + 1]};
  f32x2 prod, acc = {1.0f, 2.0f};
  // prod = {src0.lo * t.lo, src0.lo * t.hi}   (op_sel_hi for src0 = 0: the high result takes src0's LOW half)
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(prod) : "v"(src0), "v"(t));
  // acc = {acc.lo + prod.hi, acc.hi + prod.lo}
  asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(acc) : "v"(acc), "v"(prod));
  out[2 * i] = acc[0];
  out[2 * i + 1] = acc[1];
}

int main() {
  const int n = 1 << 16;
  std::vector<float> x(n), taps(2 * n), ref(2 * n), got(2 * n);
  std::vector<unsigned> junk(n);
  const unsigned patterns[] = {0x00000000u, 0x3f800000u, 0x7fc00000u, 0x7f800001u, 0xffffffffu, 0x7f800000u, 0xff800000u,
                               0x00000001u, 0x807fffffu, 0x7f7fffffu, 0x00012345u, 0xdeadbeefu};
  for (int i = 0; i < n; ++i) {
    x[i] = 0.001f * (float)(i % 977) - 0.3f;
    taps[2 * i] = 0.01f * (float)(i % 31) + 0.1f;
    taps[2 * i + 1] = 0.02f * (float)(i % 17) - 0.05f;
    junk[i] = patterns[i % 12];
    ref[2 * i] = 1.0f + x[i] * taps[2 * i + 1];
    ref[2 * i + 1] = 2.0f + x[i] * taps[2 * i];
  }
  float *dx, *dt, *dout;
  unsigned* dj;
  hipMalloc(&dx, n * 4); hipMalloc(&dt, 2 * n * 4); hipMalloc(&dout, 2 * n * 4); hipMalloc(&dj, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dt, taps.data(), 2 * n * 4, hipMemcpyHostToDevice);
  hipMemcpy(dj, junk.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dj, dt, dout, n);
  hipMemcpy(got.data(), dout, 2 * n * 4, hipMemcpyDeviceToHost);
  int bad_by_pattern[12] = {0};
  for (int i = 0; i < n; ++i)
    for (int h = 0; h < 2; ++h)
      if (memcmp(&got[2 * i + h], &ref[2 * i + h], 4) != 0) ++bad_by_pattern[i % 12];
  for (int p = 0; p < 12; ++p) printf("other half = 0x%08x: %d of %d results differ from the scalar reference\n", patterns[p], bad_by_pattern[p], 2 * (n / 12));
  return 0;
}
