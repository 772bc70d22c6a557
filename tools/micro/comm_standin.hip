// Stand-in for RCCL's reduce kernels on ONE GPU (round 5, review item 9): a persistent streaming reduce a[i] += b[i] that
// occupies a GIVEN number of workgroups for a given number of passes -- what a ring all-reduce's channels do to the chip
// while the backward pass runs (RCCL runs one workgroup per channel; its buffers live in HBM / peer HBM).  Lets
// tools/micro/comm_contention.py measure what the train step loses per occupied workgroup before an 8-GPU node exists.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/comm_standin.hip -o /tmp/libcomm_standin.so
#include <hip/hip_runtime.h>

// ``idle``: s_sleep units (64 clocks each) after every 4 KiB a workgroup moves -- a ring's workgroups wait on the link
// (7 x ~153 GB/s of xGMI per GPU, a fraction of what HBM gives them): idle > 0 throttles the stand-in to a link-like rate.
__global__ __launch_bounds__(256) void standin_reduce_kernel(float4* __restrict__ a, const float4* __restrict__ b, long n4,
                                                             int passes, int idle) {
  for (int p = 0; p < passes; ++p)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
      float4 x = a[i];
      const float4 y = b[i];
      x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      a[i] = x;
      for (int k = 0; k < idle; ++k) __builtin_amdgcn_s_sleep(127);
    }
}

extern "C" int standin_reduce(void* a, const void* b, long n_floats, int workgroups, int passes, int idle, void* stream) {
  hipLaunchKernelGGL(standin_reduce_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (float4*)a, (const float4*)b,
                     n_floats / 4, passes, idle);
  return (int)hipGetLastError();
}
