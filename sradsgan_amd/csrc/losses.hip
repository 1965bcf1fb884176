// Loss reductions of the training step (SRADSGAN/model/sradsgan.py): nn.L1Loss (:686, used :834, :838), the WGAN
// critic means of GANLoss (:35-67, used :847, :876-878) and the gradient-penalty reduction (:630-637: per-pixel L2
// norm over channels, (norm - 1)^2, mean).  Each is one pass over its input (HBM-bound, 16-byte loads where the
// layout allows) into per-block partial sums, and one single-block pass that adds the partials in a fixed order:
// deterministic, no atomics.  The scalar results stay on the device; backward kernels read the incoming scalar
// gradient through a device pointer, so nothing synchronises with the host.
#include "common.h"

namespace srhip {

constexpr int LS_MAXB = 1024;     // partial sums per reduction

__device__ inline float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// partial[block] = sum |a - b| over a grid-stride range
__global__ __launch_bounds__(256) void l1_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ partial, long n) {
  __shared__ float red[4];
  const long n4 = n >> 2;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    s += (fabsf(x.x - y.x) + fabsf(x.y - y.y)) + (fabsf(x.z - y.z) + fabsf(x.w - y.w));
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += fabsf(a[i] - b[i]);
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// partial[block] = sum x
__global__ __launch_bounds__(256) void sum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial, long n) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += x[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// partial[block] = sum over pixels of (||g_pixel||_2 - 1)^2, pixel = c consecutive floats (c <= 4)
__global__ __launch_bounds__(256) void gp_partial_kernel(const float* __restrict__ g, float* __restrict__ partial, long npix, int c) {
  __shared__ float red[4];
  float s = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    float q = 0.f;
    for (int j = 0; j < c; ++j) {
      const float v = g[p * c + j];
      q += v * v;
    }
    const float d = sqrtf(q) - 1.f;
    s += d * d;
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// out[0] = scale * sum(partial[0..nb))
__global__ __launch_bounds__(256) void finish_sum_kernel(const float* __restrict__ partial, int nb, float scale, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[i];
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

// da = sign(a - b) * gout / n  (torch: sign(0) = 0); db = -da when asked for
__global__ __launch_bounds__(256) void l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ gout, float* __restrict__ da,
                                                     float* __restrict__ db, long n, float inv_n) {
  const float gs = gout[0] * inv_n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = a[i] - b[i];
    const float v = d > 0.f ? gs : (d < 0.f ? -gs : 0.f);
    da[i] = v;
    if (db) db[i] = -v;
  }
}

// dx = gout * scale everywhere (backward of a mean)
__global__ __launch_bounds__(256) void fill_scaled_kernel(const float* __restrict__ gout, float* __restrict__ dx, long n, float scale) {
  const float v = gout[0] * scale;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) dx[i] = v;
}

// d penalty / d g = gout * 2 (norm - 1) / npix * g / norm; 0 where norm == 0 (torch's norm backward)
__global__ __launch_bounds__(256) void gp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ gout,
                                                     float* __restrict__ dg, long npix, int c, float inv_npix) {
  const float gs = gout[0] * 2.f * inv_npix;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    float q = 0.f;
    for (int j = 0; j < c; ++j) {
      const float v = g[p * c + j];
      q += v * v;
    }
    const float nrm = sqrtf(q);
    const float f = nrm > 0.f ? gs * (nrm - 1.f) / nrm : 0.f;
    for (int j = 0; j < c; ++j) dg[p * c + j] = f * g[p * c + j];
  }
}

static inline int ls_blocks(long work_items) {
  long b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > LS_MAXB ? LS_MAXB : b));
}

}  // namespace srhip

using namespace srhip;

extern "C" {

size_t srhip_reduce_workspace(void) { return LS_MAXB * sizeof(float); }

static int ws_ok(const char* what, void* ws, size_t bytes) {
  if (!ws || bytes < srhip_reduce_workspace()) {
    set_error("%s: workspace %zu bytes < required %zu", what, bytes, srhip_reduce_workspace());
    return 0;
  }
  return 1;
}

int srhip_l1_mean_fwd(const float* a, const float* b, float* out, void* workspace, size_t workspace_bytes, long count, void* stream) {
  SRHIP_REQUIRE(count > 0, "l1_mean_fwd: empty input");
  SRHIP_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, "l1_mean_fwd: inputs must be 16-byte aligned");
  if (!ws_ok("l1_mean_fwd", workspace, workspace_bytes)) return SRHIP_ERR_WORKSPACE;
  const int nb = ls_blocks(count / 4 + 1);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(l1_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), a, b, part, count);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, as_stream(stream), part, nb, (float)(1.0 / (double)count), out);
  return check_launch("l1_mean_fwd");
}

int srhip_l1_mean_bwd(const float* a, const float* b, const float* gout, float* da, float* db, long count, void* stream) {
  SRHIP_REQUIRE(count > 0 && da != nullptr, "l1_mean_bwd: empty input / missing output");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(ls_blocks(count)), dim3(256), 0, as_stream(stream), a, b, gout, da, db, count,
                     (float)(1.0 / (double)count));
  return check_launch("l1_mean_bwd");
}

int srhip_mean_fwd(const float* x, float* out, void* workspace, size_t workspace_bytes, long count, void* stream) {
  SRHIP_REQUIRE(count > 0, "mean_fwd: empty input");
  if (!ws_ok("mean_fwd", workspace, workspace_bytes)) return SRHIP_ERR_WORKSPACE;
  const int nb = ls_blocks(count);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(sum_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), x, part, count);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, as_stream(stream), part, nb, (float)(1.0 / (double)count), out);
  return check_launch("mean_fwd");
}

int srhip_mean_bwd(const float* gout, float* dx, long count, void* stream) {
  SRHIP_REQUIRE(count > 0, "mean_bwd: empty input");
  hipLaunchKernelGGL(fill_scaled_kernel, dim3(ls_blocks(count)), dim3(256), 0, as_stream(stream), gout, dx, count,
                     (float)(1.0 / (double)count));
  return check_launch("mean_bwd");
}

int srhip_gp_norm_penalty_fwd(const float* grads, float* out, void* workspace, size_t workspace_bytes, long npix, int c, void* stream) {
  SRHIP_REQUIRE(npix > 0 && c >= 1 && c <= 4, "gp_norm_penalty_fwd: needs npix > 0 and 1 <= C <= 4 (image channels), got %ld / %d", npix, c);
  if (!ws_ok("gp_norm_penalty_fwd", workspace, workspace_bytes)) return SRHIP_ERR_WORKSPACE;
  const int nb = ls_blocks(npix);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(gp_partial_kernel, dim3(nb), dim3(256), 0, as_stream(stream), grads, part, npix, c);
  hipLaunchKernelGGL(finish_sum_kernel, dim3(1), dim3(256), 0, as_stream(stream), part, nb, (float)(1.0 / (double)npix), out);
  return check_launch("gp_norm_penalty_fwd");
}

int srhip_gp_norm_penalty_bwd(const float* grads, const float* gout, float* dgrads, long npix, int c, void* stream) {
  SRHIP_REQUIRE(npix > 0 && c >= 1 && c <= 4, "gp_norm_penalty_bwd: needs npix > 0 and 1 <= C <= 4, got %ld / %d", npix, c);
  hipLaunchKernelGGL(gp_bwd_kernel, dim3(ls_blocks(npix)), dim3(256), 0, as_stream(stream), grads, gout, dgrads, npix, c,
                     (float)(1.0 / (double)npix));
  return check_launch("gp_norm_penalty_bwd");
}

}  // extern "C"
