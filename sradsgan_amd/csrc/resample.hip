// Input-pipeline kernels: Pillow-compatible image resampling of uint8 tiles on the device and the
// uint8 -> float tensor conversion, i.e. what RGB_TrainDatasetFromFolder.__getitem__ (SRADSGAN/data/dataset.py:403-438)
// does per sample through PIL on CPU workers: lr = resize(img, BICUBIC), bc = resize(lr, BICUBIC), to_tensor().
// Integer arithmetic restated from Pillow's src/libImaging/Resample.c (8 bits per channel): double-precision filter
// weights normalised per output pixel, 22-bit fixed point, accumulation with rounding, clip to uint8 after each pass,
// horizontal pass first.  Bit-exact against Pillow (tests/golden/pil_resample.npz).
#include <cmath>
#include <vector>

#include "common.h"

namespace srhip {

constexpr int RS_PRECISION_BITS = 32 - 8 - 2;

static double rs_bicubic(double x) {
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}
static double rs_bilinear(double x) {
  if (x < 0.0) x = -x;
  if (x < 1.0) return 1.0 - x;
  return 0.0;
}
static int rs_ksize(int in_size, int out_size, int filter) {
  double filterscale = (double)in_size / out_size;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = (filter == SRHIP_FILTER_BICUBIC ? 2.0 : 1.0) * filterscale;
  return (int)std::ceil(support) * 2 + 1;
}

// one pass along x (horizontal: line = image row, stride 1 pixel) or y (vertical: line = column)
//   src [N][in_lines...]: generic strides in elements: pixel step `sstep`, line step `sline`; channels innermost
__global__ void resample_pass_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                     const int* __restrict__ bounds, const int* __restrict__ coeffs, int ksize,
                                     int out_size, long total, int c, long s_img, long s_line, long s_step, int lines,
                                     long d_img, long d_line, long d_step) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // (img, line, xx, ch), ch fastest
  if (idx >= total) return;
  const int ch = (int)(idx % c);
  long t = idx / c;
  const int xx = (int)(t % out_size);
  t /= out_size;
  const int line = (int)(t % lines);
  const long img = t / lines;
  const int xmin = bounds[2 * xx], xmax = bounds[2 * xx + 1];
  const int* k = coeffs + (long)xx * ksize;
  const unsigned char* p = src + img * s_img + (long)line * s_line + (long)xmin * s_step + ch;
  int ss = 1 << (RS_PRECISION_BITS - 1);
  for (int x = 0; x < xmax; ++x) ss += (int)p[(long)x * s_step] * k[x];
  ss >>= RS_PRECISION_BITS;
  dst[img * d_img + (long)line * d_line + (long)xx * d_step + ch] = (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
}

// [N][H][W][C] uint8 -> float32 value / 255 (torchvision to_tensor), same NHWC element order
__global__ void u8_to_float_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i] / 255.0f;
}

}  // namespace srhip

using namespace srhip;

extern "C" {

int srhip_resample_ksize(int in_size, int out_size, int filter) {
  if (in_size <= 0 || out_size <= 0 || (filter != SRHIP_FILTER_BICUBIC && filter != SRHIP_FILTER_BILINEAR)) return 0;
  return rs_ksize(in_size, out_size, filter);
}

/* host only: fills bounds[2*out_size] and coeffs[out_size * ksize] */
int srhip_resample_coeffs(int in_size, int out_size, int filter, int* bounds, int* coeffs) {
  SRHIP_REQUIRE(in_size > 0 && out_size > 0 && bounds && coeffs, "resample_coeffs: bad argument");
  SRHIP_REQUIRE(filter == SRHIP_FILTER_BICUBIC || filter == SRHIP_FILTER_BILINEAR, "resample_coeffs: unknown filter");
  double (*filt)(double) = filter == SRHIP_FILTER_BICUBIC ? rs_bicubic : rs_bilinear;
  const double scale = (double)in_size / out_size;
  double filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = (filter == SRHIP_FILTER_BICUBIC ? 2.0 : 1.0) * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < ksize; ++x) k[x] = 0.0;
    for (int x = 0; x < xmax; ++x) {
      const double w = filt((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = 0; x < ksize; ++x)
      coeffs[(long)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << RS_PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << RS_PRECISION_BITS));
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
  return SRHIP_OK;
}

/* one pass over device images [n][h][w][c] uint8 (dense): axis 1 = horizontal (w -> out_size), 0 = vertical (h -> out_size);
 * bounds / coeffs are DEVICE copies of srhip_resample_coeffs' output */
int srhip_resample_pass_u8(const unsigned char* src, unsigned char* dst, const int* bounds, const int* coeffs, int ksize,
                           int n, int h, int w, int c, int axis, int out_size, void* stream) {
  SRHIP_REQUIRE(src && dst && bounds && coeffs && ksize > 0 && n > 0 && h > 0 && w > 0 && c > 0 && out_size > 0 &&
                    (axis == 0 || axis == 1),
                "resample_pass_u8: bad argument");
  long total, s_img = (long)h * w * c, s_line, s_step, d_img, d_line, d_step;
  int lines;
  if (axis == 1) {
    lines = h; s_line = (long)w * c; s_step = c;
    d_img = (long)h * out_size * c; d_line = (long)out_size * c; d_step = c;
  } else {
    lines = w; s_line = c; s_step = (long)w * c;
    d_img = (long)out_size * w * c; d_line = c; d_step = (long)w * c;
  }
  total = (long)n * lines * out_size * c;
  hipLaunchKernelGGL(resample_pass_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, as_stream(stream), src, dst,
                     bounds, coeffs, ksize, out_size, total, c, s_img, s_line, s_step, lines, d_img, d_line, d_step);
  return check_launch("resample_pass_u8");
}

int srhip_u8_to_float(const unsigned char* src, float* dst, long count, void* stream) {
  SRHIP_REQUIRE(src && dst && count >= 0, "u8_to_float: bad argument");
  if (count == 0) return SRHIP_OK;
  hipLaunchKernelGGL(u8_to_float_kernel, dim3((unsigned)cdiv(count, 256)), dim3(256), 0, as_stream(stream), src, dst, count);
  return check_launch("u8_to_float");
}

}  // extern "C"
