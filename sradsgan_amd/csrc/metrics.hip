// Validation metrics of mfeNew_validate / validate (reference sradsgan.py:1314-1325, 1112-1125) computed on
// the device: both images are quantised exactly like torchvision's ToPILImage on a float tensor --
// mul(255).byte(): truncation toward zero and wrap modulo 256, no clamp -- then
//   * sum of squared uint8 differences and sum of the ground-truth bytes per image (-> MSE, PSNR, ERGAS),
//   * SSIM as scikit-image 0.15 compare_ssim(multichannel=True): 7x7 uniform window, K1=.01, K2=.03,
//     sample covariance, data_range 255, mean over the (H-6)x(W-6) interior and the channels.
// Integer sums are exact (uint64); SSIM runs in fp64.  HBM-bound, one pass each; NHWC input.
#include "common.h"

namespace srhip {

__device__ inline int quant_u8(float x) {
  float v = x * 255.f;
  v = fminf(fmaxf(v, -2.0e9f), 2.0e9f);
  return ((int)v) & 255;                       // (int) truncates toward zero; & 255 == mod 256 in two's complement
}

__global__ __launch_bounds__(256) void quant_sse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        unsigned long long* __restrict__ partial, long per_image) {
  __shared__ unsigned long long r0[256], r1[256];
  const int n = blockIdx.y;
  const float* pa = a + (size_t)n * per_image;
  const float* pb = b + (size_t)n * per_image;
  unsigned long long sse = 0, sb = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_image; i += (long)gridDim.x * 256) {
    const int qa = quant_u8(pa[i]), qb = quant_u8(pb[i]);
    const int d = qa - qb;
    sse += (unsigned long long)(d * d);
    sb += (unsigned long long)qb;
  }
  r0[threadIdx.x] = sse;
  r1[threadIdx.x] = sb;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      r0[threadIdx.x] += r0[threadIdx.x + o];
      r1[threadIdx.x] += r1[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[((size_t)n * gridDim.x + blockIdx.x) * 2 + 0] = r0[0];
    partial[((size_t)n * gridDim.x + blockIdx.x) * 2 + 1] = r1[0];
  }
}

// Tiled form (round 5, C <= 4): a block walks 16 x 16 tiles of interior pixels; the 22 x 22 halo of both images is quantised ONCE into
// LDS bytes, a thread sums its 7 x 7 window from there in int32 -- the byte sums are exact integers either way, so the five moments are
// the same numbers the fp64 accumulation of the reference form below produces -- and only the SSIM formula itself runs in fp64.
// (The reference form reads and quantises 294 floats from global memory per pixel: 140 us for sixteen 216 x 216 images, 2.4 % of the
// inference step.)
__global__ __launch_bounds__(256) void ssim_u8_tiled_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            double* __restrict__ partial, int h, int w, int c) {
  constexpr int T = 16, HT = T + 6;
  __shared__ unsigned char qa[HT * HT * 4], qb[HT * HT * 4];
  __shared__ double red[256];
  const int n = blockIdx.y, tid = threadIdx.x;
  const int ih = h - 6, iw = w - 6;
  const int tiles_x = (iw + T - 1) / T, tiles_y = (ih + T - 1) / T;
  const double c1 = (0.01 * 255.0) * (0.01 * 255.0), c2 = (0.03 * 255.0) * (0.03 * 255.0);
  const double cov_norm = 49.0 / 48.0;
  const int ly = tid >> 4, lx = tid & 15;
  double acc = 0.0;
  for (int t = blockIdx.x; t < tiles_x * tiles_y; t += gridDim.x) {
    const int ty0 = (t / tiles_x) * T, tx0 = (t % tiles_x) * T;
    __syncthreads();
    for (int e = tid; e < HT * HT * c; e += 256) {
      const int pix = e / c, ch = e - pix * c;
      const int gy = ty0 + pix / HT, gx = tx0 + pix % HT;
      int va = 0, vb = 0;
      if (gy < h && gx < w) {
        const size_t o = (((size_t)n * h + gy) * w + gx) * c + ch;
        va = quant_u8(a[o]);
        vb = quant_u8(b[o]);
      }
      qa[pix * 4 + ch] = (unsigned char)va;
      qb[pix * 4 + ch] = (unsigned char)vb;
    }
    __syncthreads();
    if (ty0 + ly < ih && tx0 + lx < iw) {
      for (int ch = 0; ch < c; ++ch) {
        int sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int dy = 0; dy < 7; ++dy)
#pragma unroll
          for (int dx = 0; dx < 7; ++dx) {
            const int o = ((ly + dy) * HT + lx + dx) * 4 + ch;
            const int va = qa[o], vb = qb[o];
            sx += va;
            sy += vb;
            sxx += va * va;
            syy += vb * vb;
            sxy += va * vb;
          }
        const double ux = (double)sx / 49.0, uy = (double)sy / 49.0;
        const double vx = cov_norm * ((double)sxx / 49.0 - ux * ux), vy = cov_norm * ((double)syy / 49.0 - uy * uy);
        const double vxy = cov_norm * ((double)sxy / 49.0 - ux * uy);
        acc += ((2.0 * ux * uy + c1) * (2.0 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
      }
    }
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = red[0];
}

// reference form: one thread per interior pixel (all channels); a = test image, b = ground truth, both [N,H,W,C] floats
__global__ __launch_bounds__(256) void ssim_u8_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      double* __restrict__ partial, int h, int w, int c) {
  __shared__ double red[256];
  const int n = blockIdx.y;
  const int ih = h - 6, iw = w - 6;
  const double c1 = (0.01 * 255.0) * (0.01 * 255.0), c2 = (0.03 * 255.0) * (0.03 * 255.0);
  const double cov_norm = 49.0 / 48.0;
  double acc = 0.0;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < (long)ih * iw; p += (long)gridDim.x * 256) {
    const int y = (int)(p / iw), x = (int)(p - (long)y * iw);
    for (int ch = 0; ch < c; ++ch) {
      double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
      for (int dy = 0; dy < 7; ++dy) {
        const size_t row = ((size_t)n * h + y + dy) * w;
        for (int dx = 0; dx < 7; ++dx) {
          const size_t o = (row + x + dx) * c + ch;
          const double va = (double)quant_u8(a[o]), vb = (double)quant_u8(b[o]);
          sx += va;
          sy += vb;
          sxx += va * va;
          syy += vb * vb;
          sxy += va * vb;
        }
      }
      const double ux = sx / 49.0, uy = sy / 49.0;
      const double vx = cov_norm * (sxx / 49.0 - ux * ux), vy = cov_norm * (syy / 49.0 - uy * uy);
      const double vxy = cov_norm * (sxy / 49.0 - ux * uy);
      acc += ((2.0 * ux * uy + c1) * (2.0 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = red[0];
}

constexpr int METRIC_BLOCKS = 64;

// The scalar tail of the metrics (utils.py:923-962, sradsgan.py:1314-1325) for all images in ONE launch: block partials -> MSE, PSNR,
// SSIM, ERGAS in float64.  Replaces ~22 element-wise ATen launches on [N] tensors per metric call (0.1 ms of the 5.7 ms inference step).
__global__ void metric_finish_kernel(const unsigned long long* __restrict__ sse, const double* __restrict__ ssim, double* __restrict__ out, int n,
                                     double count, double ssim_count, int c, double scale) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n) return;
  unsigned long long s0 = 0, s1 = 0;
  double ss = 0.0;
  for (int j = 0; j < METRIC_BLOCKS; ++j) {
    s0 += sse[((size_t)b * METRIC_BLOCKS + j) * 2];
    s1 += sse[((size_t)b * METRIC_BLOCKS + j) * 2 + 1];
    ss += ssim[(size_t)b * METRIC_BLOCKS + j];
  }
  const double mse = (double)s0 / count, mean_gt = (double)s1 / count;
  out[b] = mse;
  out[n + b] = mse > 0.0 ? 10.0 * log10(255.0 * 255.0 / mse) : __builtin_inf();
  out[2 * n + b] = ss / ssim_count;
  out[3 * n + b] = 100.0 * sqrt(mse / (mean_gt * mean_gt) / (double)c) / scale;
}

}  // namespace srhip

using namespace srhip;

extern "C" {

int srhip_metric_blocks(void) { return METRIC_BLOCKS; }

/* partial: uint64 [n][srhip_metric_blocks()][2] = {sum of squared byte differences, sum of b's bytes} */
int srhip_quant_sse(const float* a, const float* b, unsigned long long* partial, int n, long per_image, void* stream) {
  SRHIP_REQUIRE(a && b && partial && n > 0 && per_image > 0, "quant_sse: bad argument");
  hipLaunchKernelGGL(quant_sse_kernel, dim3(METRIC_BLOCKS, n), dim3(256), 0, as_stream(stream), a, b, partial, per_image);
  return check_launch("quant_sse");
}

/* partial: double [n][srhip_metric_blocks()] = sums of the per-pixel, per-channel SSIM index over the interior */
int srhip_ssim_u8(const float* a, const float* b, double* partial, int n, int h, int w, int c, void* stream) {
  SRHIP_REQUIRE(a && b && partial && n > 0 && h >= 7 && w >= 7 && c > 0, "ssim_u8: bad argument (needs H, W >= 7)");
  if (c <= 4)
    hipLaunchKernelGGL(ssim_u8_tiled_kernel, dim3(METRIC_BLOCKS, n), dim3(256), 0, as_stream(stream), a, b, partial, h, w, c);
  else
    hipLaunchKernelGGL(ssim_u8_kernel, dim3(METRIC_BLOCKS, n), dim3(256), 0, as_stream(stream), a, b, partial, h, w, c);
  return check_launch("ssim_u8");
}

/* out: double [4][n] = rows mse, psnr, ssim, ergas per image from the two partial arrays above (per_image = c * h * w values) */
int srhip_metric_finish(const unsigned long long* sse_partial, const double* ssim_partial, double* out, int n, int h, int w, int c,
                        double scale, void* stream) {
  SRHIP_REQUIRE(sse_partial && ssim_partial && out && n > 0 && h >= 7 && w >= 7 && c > 0 && scale > 0.0, "metric_finish: bad argument");
  hipLaunchKernelGGL(metric_finish_kernel, dim3(cdiv(n, 64)), dim3(64), 0, as_stream(stream), sse_partial, ssim_partial, out, n,
                     (double)c * h * w, (double)(h - 6) * (w - 6) * c, c, scale);
  return check_launch("metric_finish");
}

}  // extern "C"
