// Fused multi-tensor ExtraAdam (reference climategan/optim.py:137-291): one launch updates every parameter tensor.
//   update(p):  g' = g + wd * p ; m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2
//               u  = -(lr * sqrt(1 - b2^t) / (1 - b1^t)) * m / (sqrt(v) + eps)
//   extrapolation():  copy = p ; p = p + u          (optim.py:153-172; the copy is taken on the FIRST extrapolation only)
//   step():           p = copy + u                  (optim.py:174-197)
// HBM-bound elementwise: 16 B read + 12 B written per element and mode (fp32 p, g, m, v, copy).
#include "cgan_common.h"

namespace {

__global__ __launch_bounds__(256) void extra_adam_kernel(const CganAdamItem* __restrict__ items, int mode,
                                                         int save_copy, float step_size, float beta1, float beta2,
                                                         float omb1, float omb2, float eps, float weight_decay) {
  const CganAdamItem it = items[blockIdx.y];
  const long n = it.numel;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float p = it.p[i];
    float g = it.g[i];
    if (weight_decay != 0.f) g = g + weight_decay * p;
    float m = it.m[i] * beta1 + omb1 * g;
    float v = it.v[i] * beta2 + omb2 * g * g;
    it.m[i] = m;
    it.v[i] = v;
    const float u = -step_size * m / (sqrtf(v) + eps);
    if (mode == 0) {  // extrapolation
      if (save_copy) it.copy[i] = p;
      it.p[i] = p + u;
    } else {          // step
      it.p[i] = it.copy[i] + u;
    }
  }
}

}  // namespace

extern "C" int cgan_extra_adam_multi_tensor(const CganAdamItem* items_device, int32_t count, int64_t max_numel,
                                            int32_t mode, int32_t save_copy, int32_t step, double lr, double beta1,
                                            double beta2, double eps, double weight_decay, void* stream) {
  CGAN_REQUIRE(items_device && count > 0 && max_numel > 0, "extra_adam: bad arguments");
  CGAN_REQUIRE(mode == 0 || mode == 1, "extra_adam: mode must be 0 (extrapolation) or 1 (step)");
  CGAN_REQUIRE(step >= 1, "extra_adam: step counts from 1");
  CGAN_REQUIRE(lr >= 0. && eps >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1.,
               "extra_adam: Invalid hyper-parameter");
  // bias corrections in double on the host, as the reference does in Python floats (optim.py:287-289)
  // (hyper-parameters arrive as doubles: the reference forms 1 - beta and the step size in Python floats before
  // they meet the fp32 tensors, and 1 - (float)0.999 differs from (float)(1 - 0.999) by 1.3e-5 relative)
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr * sqrt(bc2) / bc1);
  long blocks = (max_numel + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(extra_adam_kernel, dim3((unsigned)blocks, count), dim3(256), 0, (hipStream_t)stream, items_device,
                     mode, save_copy, step_size, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                     (float)weight_decay);
  CGAN_CHECK_LAUNCH("extra_adam_multi_tensor");
  return CGAN_OK;
}
