// Channel / spatial attention of the discriminator (ChannelAttention, SRADSGAN/model/base_networks.py:366-403, and
// SpatialAttention, :424-457, used at sradsgan.py:495-496; the generator's stand-alone CLAM / SLAM, sradsgan.py:101-151,
// are the same arithmetic) as a small set of HBM-bound primitives on NHWC [n][hw][c] tensors, c % 4 == 0:
//
//   pool_hw   t[n][2][c] = (mean_hw x, max_hw x) + first arg-max pixel per (n, c)   (fixed != 0: re-pool with GIVEN arg-max)
//   unpool_hw out[n][hw][c] = t[n][0][c] / hw + (p == arg[n][c]) * t[n][1][c]        (the adjoint of pool_hw at fixed arg-max)
//   pool_c / unpool_c   the same along the channel axis: t[n][hw][2], first arg-max channel per pixel
//   scale     out = x * s, s broadcast over pixels (mode 0, s[n][c]) or over channels (mode 1, s[n][hw])
//   dot       out[n][c] = sum_hw a*b (mode 0) or out[n][hw] = sum_c a*b (mode 1)     (the adjoint of scale w.r.t. s)
//   sigmoid_{fwd,bwd,bwd_bwd}   on the small gate tensors; pair != 0: y[n][c] = sigmoid(x[n][0][c] + x[n][1][c])
//
// The set is closed under differentiation (pool <-> unpool, scale <-> dot, sigmoid_bwd -> sigmoid_bwd_bwd), which is what
// the WGAN-GP double backward through the discriminator needs (sradsgan.py:621, :639): every backward pass of a primitive
// is another primitive, so the host composes first- and second-order passes from these launches only.
// Ties keep the first maximum in scan order like ATen (adaptive_max_pool2d / max(dim=1)).
#include <math.h>

#include "common.h"

namespace srhip {

// ---- along hw: one block per (image, 64 channels); thread = (channel, 1 of 4 pixel lanes) ---------------------- //
template <bool FIXED>
__global__ __launch_bounds__(256) void cbam_pool_hw_kernel(const float* __restrict__ x, float* __restrict__ t, int* __restrict__ arg,
                                                           int hw, int c) {
  __shared__ float ssum[256], smax[256];
  __shared__ int sidx[256];
  const int img = blockIdx.y, cq = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + cq;
  float sum = 0.f, mx = -INFINITY;
  int idx = 0;
  if (ch < c) {
    const float* xp = x + (size_t)img * hw * c + ch;
    for (int p = pl; p < hw; p += 4) {
      const float v = xp[(size_t)p * c];
      sum += v;
      if (!FIXED && pool_takes(v, mx)) {   // strict >: the first maximum of this lane's increasing p sequence
        mx = v;
        idx = p;
      }
    }
  }
  ssum[threadIdx.x] = sum;
  smax[threadIdx.x] = mx;
  sidx[threadIdx.x] = idx;
  __syncthreads();
  if (pl == 0 && ch < c) {
    float s = (ssum[cq] + ssum[cq + 64]) + (ssum[cq + 128] + ssum[cq + 192]);
    float* o = t + (size_t)img * 2 * c;
    o[ch] = s / (float)hw;
    if (FIXED) {
      o[c + ch] = x[((size_t)img * hw + arg[(size_t)img * c + ch]) * c + ch];
    } else {
      float m = smax[cq];
      int ix = sidx[cq];
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        const float mj = smax[cq + 64 * j];
        const int ij = sidx[cq + 64 * j];
        if (pool_merge_takes(mj, ij, m, ix)) {
          m = mj;
          ix = ij;
        }
      }
      o[c + ch] = m;
      arg[(size_t)img * c + ch] = ix;
    }
  }
}

__global__ __launch_bounds__(256) void cbam_unpool_hw_kernel(const float* __restrict__ t, const int* __restrict__ arg,
                                                             float* __restrict__ out, int hw, int c, long total4) {
  const int c4 = c >> 2;
  const float inv = 1.f / (float)hw;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const int q = (int)(i % c4);
    const long pix = i / c4;
    const int p = (int)(pix % hw), img = (int)(pix / hw);
    const float4 a = *reinterpret_cast<const float4*>(t + (size_t)img * 2 * c + q * 4);
    const float4 m = *reinterpret_cast<const float4*>(t + (size_t)img * 2 * c + c + q * 4);
    const int4 ix = *reinterpret_cast<const int4*>(arg + (size_t)img * c + q * 4);
    float4 o;
    o.x = a.x * inv + (ix.x == p ? m.x : 0.f);
    o.y = a.y * inv + (ix.y == p ? m.y : 0.f);
    o.z = a.z * inv + (ix.z == p ? m.z : 0.f);
    o.w = a.w * inv + (ix.w == p ? m.w : 0.f);
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

// ---- along c: one wave per pixel ----------------------------------------------------------------------------------- //
template <bool FIXED>
__global__ __launch_bounds__(256) void cbam_pool_c_kernel(const float* __restrict__ x, float* __restrict__ t, int* __restrict__ argc,
                                                          long npix, int c) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  const float* xp = x + pix * c;
  float sum = 0.f, mx = -INFINITY;
  int idx = 0;
  for (int q = lane; q * 4 < c; q += 64) {
    const float4 v = *reinterpret_cast<const float4*>(xp + q * 4);
    sum += (v.x + v.y) + (v.z + v.w);
    if (!FIXED) {
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (pool_takes(vv[e], mx)) {
          mx = vv[e];
          idx = q * 4 + e;
        }
    }
  }
  sum = wave_sum(sum);
  if (!FIXED) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(mx, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (pool_merge_takes(om, oi, mx, idx)) {
        mx = om;
        idx = oi;
      }
    }
  }
  if (lane == 0) {
    t[pix * 2] = sum / (float)c;
    if (FIXED) {
      t[pix * 2 + 1] = xp[argc[pix]];
    } else {
      t[pix * 2 + 1] = mx;
      argc[pix] = idx;
    }
  }
}

__global__ __launch_bounds__(256) void cbam_unpool_c_kernel(const float* __restrict__ t, const int* __restrict__ argc,
                                                            float* __restrict__ out, int c, long total4) {
  const int c4 = c >> 2;
  const float inv = 1.f / (float)c;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const int q = (int)(i % c4);
    const long pix = i / c4;
    const float a = t[pix * 2] * inv, m = t[pix * 2 + 1];
    const int ix = argc[pix] - q * 4;
    float4 o;
    o.x = a + (ix == 0 ? m : 0.f);
    o.y = a + (ix == 1 ? m : 0.f);
    o.z = a + (ix == 2 ? m : 0.f);
    o.w = a + (ix == 3 ? m : 0.f);
    reinterpret_cast<float4*>(out)[i] = o;
  }
}

// ---- scale / dot ---------------------------------------------------------------------------------------------------- //
template <int MODE>
__global__ __launch_bounds__(256) void cbam_scale_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                         float* __restrict__ out, int hw, int c, long total4) {
  const int c4 = c >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    const long pix = i / c4;
    if (MODE == 0) {
      const int q = (int)(i - pix * c4);
      const float4 f = *reinterpret_cast<const float4*>(s + (pix / hw) * c + q * 4);
      v.x *= f.x; v.y *= f.y; v.z *= f.z; v.w *= f.w;
    } else {
      const float f = s[pix];
      v.x *= f; v.y *= f; v.z *= f; v.w *= f;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// out[n][c] = sum_hw a * b: block = (image, 64 channels), 4 pixel lanes per channel
__global__ __launch_bounds__(256) void cbam_dot_hw_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, int hw, int c) {
  __shared__ float ssum[256];
  const int img = blockIdx.y, cq = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int ch = blockIdx.x * 64 + cq;
  float sum = 0.f;
  if (ch < c) {
    const size_t base = (size_t)img * hw * c + ch;
    for (int p = pl; p < hw; p += 4) sum += a[base + (size_t)p * c] * b[base + (size_t)p * c];
  }
  ssum[threadIdx.x] = sum;
  __syncthreads();
  if (pl == 0 && ch < c) out[(size_t)img * c + ch] = (ssum[cq] + ssum[cq + 64]) + (ssum[cq + 128] + ssum[cq + 192]);
}

// out[pixel] = sum_c a * b: one wave per pixel
__global__ __launch_bounds__(256) void cbam_dot_c_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, long npix, int c) {
  const int lane = threadIdx.x & 63;
  const long pix = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= npix) return;
  float sum = 0.f;
  for (int q = lane; q * 4 < c; q += 64) {
    const float4 u = *reinterpret_cast<const float4*>(a + pix * c + q * 4);
    const float4 v = *reinterpret_cast<const float4*>(b + pix * c + q * 4);
    sum += (u.x * v.x + u.y * v.y) + (u.z * v.z + u.w * v.w);
  }
  sum = wave_sum(sum);
  if (lane == 0) out[pix] = sum;
}

// ---- sigmoid family (small tensors) --------------------------------------------------------------------------------- //
__device__ inline float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// pair: x [n][2][c] -> y [n][c] = sigmoid(x0 + x1); count = elements of y
__global__ void cbam_sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long count, int c, int pair) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  if (pair) {
    const long n = i / c, ch = i - n * c;
    y[i] = sigmoidf(x[n * 2 * c + ch] + x[n * 2 * c + c + ch]);
  } else {
    y[i] = sigmoidf(x[i]);
  }
}
// dx = g * y (1 - y), written to both rows when pair
__global__ void cbam_sigmoid_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ dx, long count,
                                        int c, int pair) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float yv = y[i];
  const float d = g[i] * (yv * (1.f - yv));
  if (pair) {
    const long n = i / c, ch = i - n * c;
    dx[n * 2 * c + ch] = d;
    dx[n * 2 * c + c + ch] = d;
  } else {
    dx[i] = d;
  }
}
// cotangent gg on dx -> dg = gg y (1 - y), dy = gg g (1 - 2y); pair: gg = gg0 + gg1
__global__ void cbam_sigmoid_bwd_bwd_kernel(const float* __restrict__ gg, const float* __restrict__ g, const float* __restrict__ y,
                                            float* __restrict__ dg, float* __restrict__ dy, long count, int c, int pair) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  float u;
  if (pair) {
    const long n = i / c, ch = i - n * c;
    u = gg[n * 2 * c + ch] + gg[n * 2 * c + c + ch];
  } else {
    u = gg[i];
  }
  const float yv = y[i];
  if (dg) dg[i] = u * (yv * (1.f - yv));
  if (dy) dy[i] = u * g[i] * (1.f - 2.f * yv);
}

static inline int ew_blocks(long total4) {
  long b = (total4 + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace srhip

using namespace srhip;

extern "C" {

#define CBAM_CHECK(what)                                                                                        \
  SRHIP_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 4 == 0, what ": needs n, hw > 0 and C %% 4 == 0, got %d / %d / %d", n, hw, c)

int srhip_cbam_pool_hw(const float* x, float* t, int* arg, int fixed_arg, int n, int hw, int c, void* stream) {
  CBAM_CHECK("cbam_pool_hw");
  SRHIP_REQUIRE(x && t && arg, "cbam_pool_hw: null tensor");
  const dim3 grid(cdiv(c, 64), n);
  if (fixed_arg)
    hipLaunchKernelGGL(cbam_pool_hw_kernel<true>, grid, dim3(256), 0, as_stream(stream), x, t, arg, hw, c);
  else
    hipLaunchKernelGGL(cbam_pool_hw_kernel<false>, grid, dim3(256), 0, as_stream(stream), x, t, arg, hw, c);
  return check_launch("cbam_pool_hw");
}

int srhip_cbam_unpool_hw(const float* t, const int* arg, float* out, int n, int hw, int c, void* stream) {
  CBAM_CHECK("cbam_unpool_hw");
  SRHIP_REQUIRE(t && arg && out && (((uintptr_t)t | (uintptr_t)arg | (uintptr_t)out) & 15) == 0, "cbam_unpool_hw: null / unaligned tensor");
  const long total4 = (long)n * hw * (c / 4);
  hipLaunchKernelGGL(cbam_unpool_hw_kernel, dim3(ew_blocks(total4)), dim3(256), 0, as_stream(stream), t, arg, out, hw, c, total4);
  return check_launch("cbam_unpool_hw");
}

int srhip_cbam_pool_c(const float* x, float* t, int* argc, int fixed_arg, int n, int hw, int c, void* stream) {
  CBAM_CHECK("cbam_pool_c");
  SRHIP_REQUIRE(x && t && argc && ((uintptr_t)x & 15) == 0, "cbam_pool_c: null / unaligned tensor");
  const long npix = (long)n * hw;
  if (fixed_arg)
    hipLaunchKernelGGL(cbam_pool_c_kernel<true>, dim3(cdiv(npix, 4)), dim3(256), 0, as_stream(stream), x, t, argc, npix, c);
  else
    hipLaunchKernelGGL(cbam_pool_c_kernel<false>, dim3(cdiv(npix, 4)), dim3(256), 0, as_stream(stream), x, t, argc, npix, c);
  return check_launch("cbam_pool_c");
}

int srhip_cbam_unpool_c(const float* t, const int* argc, float* out, int n, int hw, int c, void* stream) {
  CBAM_CHECK("cbam_unpool_c");
  SRHIP_REQUIRE(t && argc && out && ((uintptr_t)out & 15) == 0, "cbam_unpool_c: null / unaligned tensor");
  const long total4 = (long)n * hw * (c / 4);
  hipLaunchKernelGGL(cbam_unpool_c_kernel, dim3(ew_blocks(total4)), dim3(256), 0, as_stream(stream), t, argc, out, c, total4);
  return check_launch("cbam_unpool_c");
}

int srhip_cbam_scale(const float* x, const float* s, float* out, int n, int hw, int c, int mode, void* stream) {
  CBAM_CHECK("cbam_scale");
  SRHIP_REQUIRE(x && s && out && (((uintptr_t)x | (uintptr_t)out | (uintptr_t)s) & 15) == 0 && (mode == 0 || mode == 1),
                "cbam_scale: null / unaligned tensor or bad mode");
  const long total4 = (long)n * hw * (c / 4);
  if (mode == 0)
    hipLaunchKernelGGL(cbam_scale_kernel<0>, dim3(ew_blocks(total4)), dim3(256), 0, as_stream(stream), x, s, out, hw, c, total4);
  else
    hipLaunchKernelGGL(cbam_scale_kernel<1>, dim3(ew_blocks(total4)), dim3(256), 0, as_stream(stream), x, s, out, hw, c, total4);
  return check_launch("cbam_scale");
}

int srhip_cbam_dot(const float* a, const float* b, float* out, int n, int hw, int c, int mode, void* stream) {
  CBAM_CHECK("cbam_dot");
  SRHIP_REQUIRE(a && b && out && (((uintptr_t)a | (uintptr_t)b) & 15) == 0 && (mode == 0 || mode == 1), "cbam_dot: null / unaligned tensor or bad mode");
  if (mode == 0)
    hipLaunchKernelGGL(cbam_dot_hw_kernel, dim3(cdiv(c, 64), n), dim3(256), 0, as_stream(stream), a, b, out, hw, c);
  else
    hipLaunchKernelGGL(cbam_dot_c_kernel, dim3(cdiv((long)n * hw, 4)), dim3(256), 0, as_stream(stream), a, b, out, (long)n * hw, c);
  return check_launch("cbam_dot");
}

int srhip_sigmoid_fwd(const float* x, float* y, long count, int c, int pair, void* stream) {
  SRHIP_REQUIRE(x && y && count > 0 && (!pair || c > 0), "sigmoid_fwd: bad arguments");
  hipLaunchKernelGGL(cbam_sigmoid_fwd_kernel, dim3(cdiv(count, 256)), dim3(256), 0, as_stream(stream), x, y, count, c, pair);
  return check_launch("sigmoid_fwd");
}

int srhip_sigmoid_bwd(const float* g, const float* y, float* dx, long count, int c, int pair, void* stream) {
  SRHIP_REQUIRE(g && y && dx && count > 0 && (!pair || c > 0), "sigmoid_bwd: bad arguments");
  hipLaunchKernelGGL(cbam_sigmoid_bwd_kernel, dim3(cdiv(count, 256)), dim3(256), 0, as_stream(stream), g, y, dx, count, c, pair);
  return check_launch("sigmoid_bwd");
}

int srhip_sigmoid_bwd_bwd(const float* gg, const float* g, const float* y, float* dg, float* dy, long count, int c, int pair,
                          void* stream) {
  SRHIP_REQUIRE(gg && g && y && count > 0 && (!pair || c > 0), "sigmoid_bwd_bwd: bad arguments");
  hipLaunchKernelGGL(cbam_sigmoid_bwd_bwd_kernel, dim3(cdiv(count, 256)), dim3(256), 0, as_stream(stream), gg, g, y, dg, dy, count, c, pair);
  return check_launch("sigmoid_bwd_bwd");
}

}  // extern "C"
