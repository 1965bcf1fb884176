// torch.optim.Adam over a flat fp32 arena (+ the discriminator's weight clip), one HBM pass.
// Roofline: HBM bandwidth; 16 B read + 12 B written per parameter.
#include "common.h"

namespace srhip {

// state = {step, lr/(1-b1^step), sqrt(1-b2^step), 0}; advanced on the device so that the launch pair
// below can sit inside a captured hipGraph (kernel arguments are frozen at capture time).
__global__ void adam_tick_kernel(float* state, float lr, float b1, float b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double step = (double)state[0] + 1.0;
    state[0] = (float)step;
    state[1] = (float)((double)lr / (1.0 - pow((double)b1, step)));
    state[2] = (float)sqrt(1.0 - pow((double)b2, step));
  }
}

__device__ inline float adam_one(float& p, float g, float& m, float& v, float b1, float b2, float eps, float step_size,
                                 float bc2_sqrt, float gscale, float clip) {
  g *= gscale;
  m = m + (g - m) * (1.f - b1);                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * b2 + (1.f - b2) * g * g;              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  float denom = sqrtf(v) / bc2_sqrt + eps;
  float q = p - step_size * (m / denom);        // param.addcdiv_(exp_avg, denom, value=-step_size)
  if (clip > 0.f) q = fminf(fmaxf(q, -clip), clip);
  p = q;
  return q;
}

__global__ void adam_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                            float4* __restrict__ v, const float* __restrict__ state, long n4, float b1, float b2,
                            float eps, float gscale, float clip) {
  const float step_size = state[1], bc2_sqrt = state[2];
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, b1, b2, eps, step_size, bc2_sqrt, gscale, clip);
    adam_one(pp.y, gg.y, mm.y, vv.y, b1, b2, eps, step_size, bc2_sqrt, gscale, clip);
    adam_one(pp.z, gg.z, mm.z, vv.z, b1, b2, eps, step_size, bc2_sqrt, gscale, clip);
    adam_one(pp.w, gg.w, mm.w, vv.w, b1, b2, eps, step_size, bc2_sqrt, gscale, clip);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
  }
}

}  // namespace srhip

using namespace srhip;

extern "C" int srhip_adam_step(float* p, const float* g, float* m, float* v, float* state, long n, float lr, float b1,
                               float b2, float eps, float grad_scale, float clip, void* stream) {
  SRHIP_REQUIRE(p && g && m && v && state && n >= 0, "adam_step: null tensor");
  SRHIP_REQUIRE(n % 4 == 0, "adam_step: arena length must be a multiple of 4 elements");
  SRHIP_REQUIRE(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0,
                "adam_step: arenas must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, st, state, lr, b1, b2);
  long n4 = n / 4;
  if (n4 > 0) {
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, (float4*)p, (const float4*)g, (float4*)m,
                       (float4*)v, state, n4, b1, b2, eps, grad_scale, clip);
  }
  return check_launch("adam_step");
}
