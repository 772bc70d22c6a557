// Microbenchmark: cost of launching many short workgroups as a function of their dynamic LDS allocation and of a
// first-touch global load (gfx950).  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/micro/wg_launch_cost.hip -o /tmp/wgl && /tmp/wgl
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void empty_kernel(float* out, int iters) {
  extern __shared__ float sm[];
  float v = 0.f;
  for (int i = 0; i < iters; ++i) v += __builtin_amdgcn_s_getreg(63492) * 1e-30f;   // keeps the loop alive, no memory
  if (v == 123.f) out[0] = v + sm[threadIdx.x];
}

int main() {
  float* out;
  hipMalloc(&out, 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int lds_list[] = {0, 16 << 10, 32 << 10, 64 << 10, 75 << 10, 120 << 10};
  const int wg_list[] = {512, 2304, 12800};
  for (int lds : lds_list) {
    hipFuncSetAttribute((const void*)empty_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int wgs : wg_list)
      for (int iters : {0, 2000}) {
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), lds, 0, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), lds, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("lds %6d B  wgs %6d  iters %5d : %8.1f us per launch, %6.3f us per workgroup\n", lds, wgs, iters,
               ms * 100.f, ms * 100.f / wgs);
      }
  }
  return 0;
}
