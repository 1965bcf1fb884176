// HBM-bound pieces of the SRADSGAN step: activation backward, pixel shuffle, column sums.
// All are one pass over the tensor with 16-byte accesses; roofline = HBM bandwidth.
#include <atomic>
#include <mutex>
#include "conv_internal.h"

#include <string.h>

namespace srhip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__global__ void lrelu_bwd_kernel(const float4* __restrict__ dy, const float4* __restrict__ y, float4* __restrict__ dx,
                                 long n4, float slope) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 g = dy[i], v = y[i];
    g.x = v.x > 0.f ? g.x : g.x * slope;
    g.y = v.y > 0.f ? g.y : g.y * slope;
    g.z = v.z > 0.f ? g.z : g.z * slope;
    g.w = v.w > 0.f ? g.w : g.w * slope;
    dx[i] = g;
  }
}
// The same with the SIGN of y as a bit mask (round 5): MODE 1 reads y and also WRITES one bit per element -- per wave chunk of 64 float4s
// the four component ballots, 8 dwords --, MODE 2 reads that mask INSTEAD of y (32 bytes per 1024 bytes of y).  The gradient penalty
// applies the backward of the discriminator's first LeakyReLU twice to the same 382 MB activation (first order, then its double
// backward): the second application reads 12 MB.
template <int MODE>
__global__ __launch_bounds__(256) void lrelu_bwd_bits_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                             unsigned long long* __restrict__ bits, float4* __restrict__ dx, long n4, float slope) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;   // a multiple of 64: a wave always holds one aligned chunk of 64 float4s
  const int lane = threadIdx.x & 63;
  for (; (i & ~63L) < n4; i += stride) {               // whole waves iterate together (the ballots need every lane of the chunk)
    const bool live = i < n4;
    float4 g = live ? dy[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    bool p0, p1, p2, p3;
    unsigned long long* w = bits + (i >> 6) * 4;
    if (MODE == 1) {
      const float4 v = live ? y[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      p0 = v.x > 0.f; p1 = v.y > 0.f; p2 = v.z > 0.f; p3 = v.w > 0.f;
      const unsigned long long b0 = __ballot(p0), b1 = __ballot(p1), b2 = __ballot(p2), b3 = __ballot(p3);
      if (lane == 0) { w[0] = b0; w[1] = b1; w[2] = b2; w[3] = b3; }
    } else {
      p0 = (w[0] >> lane) & 1ull; p1 = (w[1] >> lane) & 1ull; p2 = (w[2] >> lane) & 1ull; p3 = (w[3] >> lane) & 1ull;
    }
    if (live) {
      g.x = p0 ? g.x : g.x * slope;
      g.y = p1 ? g.y : g.y * slope;
      g.z = p2 ? g.z : g.z * slope;
      g.w = p3 ? g.w : g.w * slope;
      dx[i] = g;
    }
  }
}
// out = ((src[0] + src[1]) + src[2]) + ... in that order: the generator's stratified dense-sampling bus (sradsgan.py:455-460) as ONE
// pass over its n terms instead of n - 1 chained adds that each re-read the running sum (12 x 72 MB -> 14 x 24 MB at B = 32)
struct SumSrcs {
  const float4* p[16];
};
__global__ void sum_n_kernel(SumSrcs a, int n, float4* __restrict__ out, long n4) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    float4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < n) v[k] = a.p[k][i];                      // all loads in flight before the first add
    float4 acc = v[0];
#pragma unroll
    for (int k = 1; k < 16; ++k)
      if (k < n) {
        acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w;
      }
    out[i] = acc;
  }
}

// torch.cat(dim = 1) of NHWC tensors and its backward (round 6): row p of the wide tensor = the rows p of the narrow ones side by side
// (sradsgan.py:340-344: the multi-scale block's three branches; ATen's channels-last cat took 169 us for 72 MB, 5x its bytes' worth).
// SPLIT: the wide tensor is the source, the narrow ones the destinations.  One float4 per thread and step, rows are whole 16-byte runs.
struct CatSrcs {
  float4* p[8];
  int q0[9];                                              // first quad of tensor k inside a wide row; q0[n] = quads per wide row
};
template <bool SPLIT>
__global__ void cat_channels_kernel(CatSrcs a, int n, float4* __restrict__ wide, long total) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  const int qt = a.q0[n];
  for (; i < total; i += stride) {
    const long row = i / qt;
    const int q = (int)(i - row * qt);
    int k = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j)
      if (j < n && q >= a.q0[j]) k = j;
    float4* narrow = a.p[k] + row * (a.q0[k + 1] - a.q0[k]) + (q - a.q0[k]);
    if (SPLIT) *narrow = wide[i];
    else wide[i] = *narrow;
  }
}

__global__ void lrelu_bwd_tail_kernel(const float* dy, const float* y, float* dx, long begin, long n, float slope) {
  long i = begin + (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = y[i] > 0.f ? dy[i] : dy[i] * slope;
}

// NHWC pixel shuffle. One thread per 4 output channels: out[n, h*r+i, w*r+j, c..c+3] gathers
// in[n,h,w,(c+e)*r*r + i*r + j], e=0..3 (stride r*r apart) -- reads of a wave cover the whole
// r*r*C contiguous source row, writes are fully coalesced.
__global__ void pixel_shuffle_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int h, int w,
                                         int c, int r, float slope, int act) {
  long total = (long)n * h * r * w * r * (c / 4);
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int c4 = (int)(idx % (c / 4));
  long pix = idx / (c / 4);
  int ow = (int)(pix % (w * r));
  long t = pix / (w * r);
  int oh = (int)(t % (h * r));
  int b = (int)(t / (h * r));
  int hh = oh / r, i = oh - hh * r, ww = ow / r, j = ow - ww * r;
  const float* s = in + ((size_t)(b * h + hh) * w + ww) * ((size_t)c * r * r) + (size_t)(c4 * 4) * r * r + i * r + j;
  float4 v;
  v.x = s[0];
  v.y = s[r * r];
  v.z = s[2 * r * r];
  v.w = s[3 * r * r];
  if (act) {
    v.x = v.x > 0.f ? v.x : v.x * slope;
    v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope;
    v.w = v.w > 0.f ? v.w : v.w * slope;
  }
  reinterpret_cast<float4*>(out)[idx] = v;
}

// Block form (round 5, r = 2 or 3): a thread owns 4 consecutive OUTPUT channels c .. c+3 of one INPUT pixel, i.e. the 4 r^2
// consecutive input channels c r^2 .. (c + 4) r^2 - 1 (r^2 float4, contiguous) and, for each of the r^2 sub-pixels (i, j), the float4
// out[n, h r + i, w r + j, c .. c+3]: every load and every store is a 16-byte access and consecutive lanes touch consecutive pieces.
// FWD: in -> out (+ LeakyReLU); !FWD: dout (x LeakyReLU'(outv)) -> din.  (The one-thread-per-output-quad forms above / below read or
// wrote 4-byte scalars r^2 floats apart: 118 / 145 us for the up-sampler's two stages at B = 32.)
template <int R, bool FWD>
__global__ __launch_bounds__(256) void pixel_shuffle_block_kernel(const float* __restrict__ src, const float* __restrict__ outv,
                                                                  float* __restrict__ dst, int n, int h, int w, int c, float slope, int act) {
  constexpr int RR = R * R;
  const int c4 = c / 4;
  const long total = (long)n * h * w * c4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % c4);
  long pix = idx / c4;
  const int ww = (int)(pix % w);
  pix /= w;
  const int hh = (int)(pix % h);
  const int b = (int)(pix / h);
  const size_t in_off = (((size_t)b * h + hh) * w + ww) * ((size_t)c * RR) + (size_t)q * 4 * RR;      // first of the 4 RR input channels
  float v[4 * RR];                                                                                       // [k][ij]: input channel (4 q + k) RR + ij
  if (FWD) {
#pragma unroll
    for (int e = 0; e < RR; ++e) {
      const float4 t = reinterpret_cast<const float4*>(src + in_off)[e];
      v[4 * e] = t.x; v[4 * e + 1] = t.y; v[4 * e + 2] = t.z; v[4 * e + 3] = t.w;
    }
  }
#pragma unroll
  for (int ij = 0; ij < RR; ++ij) {
    const int i = ij / R, j = ij - i * R;
    const size_t o = ((((size_t)b * (h * R) + (hh * R + i)) * (size_t)(w * R) + (ww * R + j)) * (size_t)c) + (size_t)q * 4;
    if (FWD) {
      float4 t = make_float4(v[0 * RR + ij], v[1 * RR + ij], v[2 * RR + ij], v[3 * RR + ij]);
      if (act) {
        t.x = t.x > 0.f ? t.x : t.x * slope;
        t.y = t.y > 0.f ? t.y : t.y * slope;
        t.z = t.z > 0.f ? t.z : t.z * slope;
        t.w = t.w > 0.f ? t.w : t.w * slope;
      }
      *reinterpret_cast<float4*>(dst + o) = t;
    } else {
      float4 t = *reinterpret_cast<const float4*>(src + o);
      if (act) {
        const float4 y = *reinterpret_cast<const float4*>(outv + o);
        t.x = y.x > 0.f ? t.x : t.x * slope;
        t.y = y.y > 0.f ? t.y : t.y * slope;
        t.z = y.z > 0.f ? t.z : t.z * slope;
        t.w = y.w > 0.f ? t.w : t.w * slope;
      }
      v[0 * RR + ij] = t.x; v[1 * RR + ij] = t.y; v[2 * RR + ij] = t.z; v[3 * RR + ij] = t.w;
    }
  }
  if (!FWD) {
#pragma unroll
    for (int e = 0; e < RR; ++e)
      reinterpret_cast<float4*>(dst + in_off)[e] = make_float4(v[4 * e], v[4 * e + 1], v[4 * e + 2], v[4 * e + 3]);
  }
}

// Gather form: one thread per 4 consecutive INPUT channels of one input pixel -- a 16-byte store per lane, whole rows per store
// instruction; the four sources are scalar loads that consecutive lanes take from consecutive channels (r = 2: the same output pixel).
// (The scatter form of rounds 1-4 -- one thread per output quad, four 4-byte stores 16 bytes apart -- made the memory side write
// 478 MB for a 239 MB tensor: profiles/r05_step_traffic.txt.)
__global__ void pixel_shuffle_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ outv,
                                         float* __restrict__ din, int n, int h, int w, int c, int r, float slope,
                                         int act) {
  const int rr = r * r, cin4 = c * rr / 4;
  const long total = (long)n * h * w * cin4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % cin4);
  long pix = idx / cin4;
  const int ww = (int)(pix % w);
  pix /= w;
  const int hh = (int)(pix % h);
  const int b = (int)(pix / h);
  float g[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int cc = q * 4 + e;
    const int co = cc / rr, ij = cc - co * rr;
    const int i = ij / r, j = ij - i * r;
    const size_t o = (((size_t)b * (h * r) + (hh * r + i)) * (size_t)(w * r) + (ww * r + j)) * (size_t)c + co;
    float v = dout[o];
    if (act) v = outv[o] > 0.f ? v : v * slope;
    g[e] = v;
  }
  reinterpret_cast<float4*>(din)[idx] = make_float4(g[0], g[1], g[2], g[3]);
}

// nn.MaxPool2d(2, 2) on NHWC (vgg19.features[4], [9]); one thread per 4 channels of one output pixel.
// Ties keep the first element in window scan order (0,0),(0,1),(1,0),(1,1), like ATen.
__global__ void maxpool2x2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w, int c) {
  const int ho = h / 2, wo = w / 2, c4 = c / 4;
  const long total = (long)n * ho * wo * c4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % c4);
  long pix = idx / c4;
  const int ow = (int)(pix % wo);
  pix /= wo;
  const int oh = (int)(pix % ho);
  const int b = (int)(pix / ho);
  const float* p = x + (((size_t)b * h + oh * 2) * w + ow * 2) * c + q * 4;
  const float4 v00 = *reinterpret_cast<const float4*>(p), v01 = *reinterpret_cast<const float4*>(p + c);
  const float4 v10 = *reinterpret_cast<const float4*>(p + (size_t)w * c), v11 = *reinterpret_cast<const float4*>(p + (size_t)w * c + c);
  float4 o;
  o.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
  o.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
  o.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
  o.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
  reinterpret_cast<float4*>(y)[idx] = o;
}

// The same forward that also leaves, per thread (4 channels of one output pixel), a 16-bit record of what the backward needs from
// x: per channel the arg-max position of pool_bwd1 (2 bits: first maximum in window scan order) and whether that maximum is > 0
// (1 bit) -- 1 / 32 of x's bytes.  maxpool2x2_bwd_idx_kernel reads the record instead of the four x values: the backward of VGG's
// first pool reads 12 MB instead of the 382 MB activation (on the main stream, in the generator's backward).
__global__ void maxpool2x2_fwd_idx_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned short* __restrict__ rec, int n, int h,
                                          int w, int c) {
  const int ho = h / 2, wo = w / 2, c4 = c / 4;
  const long total = (long)n * ho * wo * c4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % c4);
  long pix = idx / c4;
  const int ow = (int)(pix % wo);
  pix /= wo;
  const int oh = (int)(pix % ho);
  const int b = (int)(pix / ho);
  const float* p = x + (((size_t)b * h + oh * 2) * w + ow * 2) * c + q * 4;
  const float4 v00 = *reinterpret_cast<const float4*>(p), v01 = *reinterpret_cast<const float4*>(p + c);
  const float4 v10 = *reinterpret_cast<const float4*>(p + (size_t)w * c), v11 = *reinterpret_cast<const float4*>(p + (size_t)w * c + c);
  float4 o;
  o.x = fmaxf(fmaxf(v00.x, v01.x), fmaxf(v10.x, v11.x));
  o.y = fmaxf(fmaxf(v00.y, v01.y), fmaxf(v10.y, v11.y));
  o.z = fmaxf(fmaxf(v00.z, v01.z), fmaxf(v10.z, v11.z));
  o.w = fmaxf(fmaxf(v00.w, v01.w), fmaxf(v10.w, v11.w));
  reinterpret_cast<float4*>(y)[idx] = o;
  auto one = [](float a_, float b_, float c_, float d_) {      // pool_bwd1's comparisons
    int am = 0;
    float m = a_;
    if (b_ > m) { m = b_; am = 1; }
    if (c_ > m) { m = c_; am = 2; }
    if (d_ > m) { m = d_; am = 3; }
    return (unsigned)(am | ((m > 0.f) ? 4 : 0));
  };
  rec[idx] = (unsigned short)(one(v00.x, v01.x, v10.x, v11.x) | (one(v00.y, v01.y, v10.y, v11.y) << 3) | (one(v00.z, v01.z, v10.z, v11.z) << 6) |
                              (one(v00.w, v01.w, v10.w, v11.w) << 9));
}
__global__ void maxpool2x2_bwd_idx_kernel(const float* __restrict__ dy, const unsigned short* __restrict__ rec, float* __restrict__ dx,
                                          int n, int h, int w, int c, int relu) {
  const int ho = h / 2, wo = w / 2, c4 = c / 4;
  const long total = (long)n * ho * wo * c4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % c4);
  long pix = idx / c4;
  const int ow = (int)(pix % wo);
  pix /= wo;
  const int oh = (int)(pix % ho);
  const int b = (int)(pix / ho);
  const size_t o00 = (((size_t)b * h + oh * 2) * w + ow * 2) * c + q * 4;
  const size_t o01 = o00 + c, o10 = o00 + (size_t)w * c, o11 = o10 + c;
  const float4 g = reinterpret_cast<const float4*>(dy)[idx];
  const unsigned r = rec[idx];
  const float gv[4] = {g.x, g.y, g.z, g.w};
  float o[4][4];                                        // [window position][channel]
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const unsigned f = (r >> (3 * k)) & 7u;
    const float gg = (!relu || (f & 4u)) ? gv[k] : 0.f;
#pragma unroll
    for (int pos = 0; pos < 4; ++pos) o[pos][k] = (f & 3u) == (unsigned)pos ? gg : 0.f;
  }
  *reinterpret_cast<float4*>(dx + o00) = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
  *reinterpret_cast<float4*>(dx + o01) = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
  *reinterpret_cast<float4*>(dx + o10) = make_float4(o[2][0], o[2][1], o[2][2], o[2][3]);
  *reinterpret_cast<float4*>(dx + o11) = make_float4(o[3][0], o[3][1], o[3][2], o[3][3]);
}

// backward of MaxPool2d(2,2) fused with the backward of the ReLU that produced its input x (= relu
// output, so x >= 0): dx = dy at the first maximum of each window if that maximum is > 0, else 0.
__device__ inline void pool_bwd1(float a, float b, float c, float d, float g, int relu, float& oa, float& ob,
                                 float& oc, float& od) {
  int am = 0;
  float m = a;
  if (b > m) { m = b; am = 1; }
  if (c > m) { m = c; am = 2; }
  if (d > m) { m = d; am = 3; }
  const float gg = (!relu || m > 0.f) ? g : 0.f;
  oa = am == 0 ? gg : 0.f;
  ob = am == 1 ? gg : 0.f;
  oc = am == 2 ? gg : 0.f;
  od = am == 3 ? gg : 0.f;
}
__global__ void maxpool2x2_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx,
                                      int n, int h, int w, int c, int relu) {
  const int ho = h / 2, wo = w / 2, c4 = c / 4;
  const long total = (long)n * ho * wo * c4;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int q = (int)(idx % c4);
  long pix = idx / c4;
  const int ow = (int)(pix % wo);
  pix /= wo;
  const int oh = (int)(pix % ho);
  const int b = (int)(pix / ho);
  const size_t o00 = (((size_t)b * h + oh * 2) * w + ow * 2) * c + q * 4;
  const size_t o01 = o00 + c, o10 = o00 + (size_t)w * c, o11 = o10 + c;
  const float4 v00 = *reinterpret_cast<const float4*>(x + o00), v01 = *reinterpret_cast<const float4*>(x + o01);
  const float4 v10 = *reinterpret_cast<const float4*>(x + o10), v11 = *reinterpret_cast<const float4*>(x + o11);
  const float4 g = reinterpret_cast<const float4*>(dy)[idx];
  float4 a, bb, cc, d;
  pool_bwd1(v00.x, v01.x, v10.x, v11.x, g.x, relu, a.x, bb.x, cc.x, d.x);
  pool_bwd1(v00.y, v01.y, v10.y, v11.y, g.y, relu, a.y, bb.y, cc.y, d.y);
  pool_bwd1(v00.z, v01.z, v10.z, v11.z, g.z, relu, a.z, bb.z, cc.z, d.z);
  pool_bwd1(v00.w, v01.w, v10.w, v11.w, g.w, relu, a.w, bb.w, cc.w, d.w);
  *reinterpret_cast<float4*>(dx + o00) = a;
  *reinterpret_cast<float4*>(dx + o01) = bb;
  *reinterpret_cast<float4*>(dx + o10) = cc;
  *reinterpret_cast<float4*>(dx + o11) = d;
}

// column sums of a [rows][ld] matrix, first C columns (bias gradient on the generic conv path).
// stage 1: <= 1024 blocks, each reduces a contiguous slab of rows with 16-byte loads (T = float4)
// or scalar loads (T = float); stage 2: 4 row-lanes per column over the block partials.
template <typename T>
__global__ void colsum_stage1(const float* __restrict__ dy, float* __restrict__ partial, long rows, int c, int ld,
                              long rows_per_block) {
  constexpr int V = sizeof(T) / 4;
  __shared__ T red[256];
  const int tid = threadIdx.x;
  const int q = c / V;                       // column groups (q <= 256)
  const int nrl = 256 / q;                   // row lanes
  const int cq = tid % q, rl = tid / q;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  T s;
  for (int e = 0; e < V; ++e) reinterpret_cast<float*>(&s)[e] = 0.f;
  if (rl < nrl)
    for (long r = r0 + rl; r < r1; r += nrl) {
      T v = *reinterpret_cast<const T*>(dy + (size_t)r * ld + cq * V);
      for (int e = 0; e < V; ++e) reinterpret_cast<float*>(&s)[e] += reinterpret_cast<float*>(&v)[e];
    }
  red[tid] = s;
  __syncthreads();
  if (tid < q) {
    T t = red[tid];
    for (int k = 1; k < nrl; ++k) {
      T v = red[k * q + tid];
      for (int e = 0; e < V; ++e) reinterpret_cast<float*>(&t)[e] += reinterpret_cast<float*>(&v)[e];
    }
    *reinterpret_cast<T*>(partial + (size_t)blockIdx.x * c + tid * V) = t;
  }
}
__global__ void colsum_stage2(const float* __restrict__ partial, float* __restrict__ out, int nblk, int c) {
  __shared__ float red[256];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), sub = threadIdx.x >> 4;       // 16 columns x 16 slab lanes
  float s = 0.f;
  if (col < c)
    for (int b = sub; b < nblk; b += 16) s += partial[(size_t)b * c + col];
  red[threadIdx.x] = s;
  __syncthreads();
  if (sub == 0 && col < c) {
    float t = 0.f;
    for (int k = 0; k < 16; ++k) t += red[threadIdx.x + 16 * k];
    out[col] = t;
  }
}

static long colsum_nblk(long rows) {
  long nblk = (rows + 63) / 64;
  return nblk > 256 ? 256 : (nblk < 1 ? 1 : nblk);
}
size_t colsum_workspace_bytes(long rows, int c) { return (size_t)colsum_nblk(rows) * c * sizeof(float); }

int colsum_launch(const float* dy, float* db, void* workspace, long rows, int c, int ld, hipStream_t st) {
  const long nblk = colsum_nblk(rows);
  const long rpb = (rows + nblk - 1) / nblk;
  const bool vec = (c % 4 == 0) && (ld % 4 == 0) && ((uintptr_t)dy % 16 == 0) && c <= 1024;
  if (vec)
    hipLaunchKernelGGL(colsum_stage1<float4>, dim3((int)nblk), dim3(256), 0, st, dy, (float*)workspace, rows, c, ld, rpb);
  else
    hipLaunchKernelGGL(colsum_stage1<float>, dim3((int)nblk), dim3(256), 0, st, dy, (float*)workspace, rows, c, ld, rpb);
  hipLaunchKernelGGL(colsum_stage2, dim3(cdiv(c, 16)), dim3(256), 0, st, (const float*)workspace, db, (int)nblk, c);
  return check_launch("colsum");
}

}  // namespace srhip

using namespace srhip;

extern "C" {

const char* srhip_last_error(void) { return g_err; }
int srhip_abi_version(void) { return 11; }

// `to` waits for everything enqueued on `from` so far: one event record + one stream wait through a small ring of
// timing-less events (an event can be re-recorded once the wait that used it has been ENQUEUED: hipStreamWaitEvent
// captures the record that is current at the call).  The weight-gradient kernels are forked to their stream ~150 times per
// training step; from Python that is torch.cuda.Event() + record + wait + a stream context per launch.
// Contract: ONE device per process (the events are created on the device that is current at the first call) and the
// ring is shared by every caller -- the slot counter is atomic and creation is guarded, so two driving threads (the main thread
// and autograd's device thread both fork) cannot hand out the same slot twice.
int srhip_stream_fork(void* from_stream, void* to_stream) {
  constexpr int RING = 64;
  static hipEvent_t ring[RING];
  static std::atomic<unsigned> next{0};
  static std::once_flag once;
  static bool created = false;
  if (from_stream == to_stream) return SRHIP_OK;
  std::call_once(once, [&]() {
    created = true;
    for (int i = 0; i < RING; ++i)
      if (hipEventCreateWithFlags(&ring[i], hipEventDisableTiming) != hipSuccess) created = false;
  });
  if (!created) {
    set_error("stream_fork: hipEventCreateWithFlags failed");
    return SRHIP_ERR_LAUNCH;
  }
  hipEvent_t ev = ring[next.fetch_add(1u, std::memory_order_relaxed) % RING];
  if (hipEventRecord(ev, as_stream(from_stream)) != hipSuccess || hipStreamWaitEvent(as_stream(to_stream), ev, 0) != hipSuccess) {
    set_error("stream_fork: %s", hipGetErrorString(hipGetLastError()));
    return SRHIP_ERR_LAUNCH;
  }
  return SRHIP_OK;
}

int srhip_lrelu_bwd(const float* dy, const float* y, float* dx, long count, float slope, void* stream) {
  SRHIP_REQUIRE(dy && y && dx && count >= 0, "lrelu_bwd: bad argument");
  if (count == 0) return SRHIP_OK;
  hipStream_t st = as_stream(stream);
  bool aligned = ((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)dx)) & 15) == 0;
  long n4 = aligned ? count / 4 : 0;
  if (n4 > 0) {
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(lrelu_bwd_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)dy, (const float4*)y,
                       (float4*)dx, n4, slope);
  }
  long done = n4 * 4;
  if (done < count)
    hipLaunchKernelGGL(lrelu_bwd_tail_kernel, dim3(cdiv(count - done, 256)), dim3(256), 0, st, dy, y, dx, done, count,
                       slope);
  return check_launch("lrelu_bwd");
}

/* ABI 9: dx = dy * LeakyReLU'(y) with the sign of y as a bit mask: y != NULL: read y and WRITE the mask; y == NULL: READ the mask instead of
 * y.  count % 4 == 0, 16-byte aligned tensors; mask = srhip_lrelu_mask_bytes(count) bytes (one bit per element, chunks of 256 elements). */
size_t srhip_lrelu_mask_bytes(long count) { return count <= 0 ? 0 : (size_t)((count / 4 + 63) / 64) * 32; }
int srhip_lrelu_bwd_bits(const float* dy, const float* y, void* mask, float* dx, long count, float slope, void* stream) {
  SRHIP_REQUIRE(dy && mask && dx && count >= 0 && count % 4 == 0, "lrelu_bwd_bits: bad argument (count % 4 == 0)");
  SRHIP_REQUIRE(((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)dx) | ((uintptr_t)mask)) & 15) == 0, "lrelu_bwd_bits: 16-byte aligned tensors");
  if (count == 0) return SRHIP_OK;
  const long n4 = count / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (y != nullptr)
    hipLaunchKernelGGL(lrelu_bwd_bits_kernel<1>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float4*)dy, (const float4*)y,
                       static_cast<unsigned long long*>(mask), (float4*)dx, n4, slope);
  else
    hipLaunchKernelGGL(lrelu_bwd_bits_kernel<2>, dim3(blocks), dim3(256), 0, as_stream(stream), (const float4*)dy, (const float4*)nullptr,
                       static_cast<unsigned long long*>(mask), (float4*)dx, n4, slope);
  return check_launch("lrelu_bwd_bits");
}

int srhip_sum_n(const float* const* srcs, int n, float* out, long count, void* stream) {
  SRHIP_REQUIRE(srcs && out && n >= 2 && n <= 16 && count >= 0 && count % 4 == 0, "sum_n: 2..16 sources, element count % 4 == 0");
  if (count == 0) return SRHIP_OK;
  SumSrcs a;
  uintptr_t bits = (uintptr_t)out;
  for (int k = 0; k < 16; ++k) {
    const float* p = srcs[k < n ? k : 0];
    SRHIP_REQUIRE(p != nullptr, "sum_n: null source");
    a.p[k] = reinterpret_cast<const float4*>(p);
    bits |= (uintptr_t)p;
  }
  SRHIP_REQUIRE((bits & 15) == 0, "sum_n: tensors must be 16-byte aligned");
  const long n4 = count / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(sum_n_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), a, n, reinterpret_cast<float4*>(out), n4);
  return check_launch("sum_n");
}

static int cat_channels_impl(float* const* narrow, const int* chans, int n, float* wide, long rows, bool split, void* stream, const char* what) {
  SRHIP_REQUIRE(narrow && chans && wide && n >= 2 && n <= 8 && rows >= 0, "%s: 2..8 tensors", what);
  CatSrcs a;
  uintptr_t bits = (uintptr_t)wide;
  int q = 0;
  for (int k = 0; k < 8; ++k) {
    const int kk = k < n ? k : 0;
    SRHIP_REQUIRE(narrow[kk] != nullptr && chans[kk] > 0 && chans[kk] % 4 == 0, "%s: null tensor / channel count not a multiple of 4", what);
    a.p[k] = reinterpret_cast<float4*>(narrow[kk]);
    bits |= (uintptr_t)narrow[kk];
    a.q0[k] = q;
    if (k < n) q += chans[k] / 4;
  }
  for (int k = n; k <= 8; ++k) a.q0[k] = q;
  SRHIP_REQUIRE((bits & 15) == 0, "%s: tensors must be 16-byte aligned", what);
  const long total = rows * q;
  if (total == 0) return SRHIP_OK;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (split) hipLaunchKernelGGL((cat_channels_kernel<true>), dim3(blocks), dim3(256), 0, as_stream(stream), a, n, reinterpret_cast<float4*>(wide), total);
  else hipLaunchKernelGGL((cat_channels_kernel<false>), dim3(blocks), dim3(256), 0, as_stream(stream), a, n, reinterpret_cast<float4*>(wide), total);
  return check_launch(what);
}
int srhip_cat_channels(const float* const* srcs, const int* chans, int n, float* out, long rows, void* stream) {
  return cat_channels_impl(const_cast<float* const*>(srcs), chans, n, out, rows, false, stream, "cat_channels");
}
int srhip_split_channels(const float* in, const int* chans, int n, float* const* dsts, long rows, void* stream) {
  return cat_channels_impl(dsts, chans, n, const_cast<float*>(in), rows, true, stream, "split_channels");
}

int srhip_pixel_shuffle_fwd(const float* in, float* out, int n, int h, int w, int cout, int r, float slope,
                            int apply_act, void* stream) {
  SRHIP_REQUIRE(in && out && n >= 0 && h > 0 && w > 0 && cout > 0 && r > 0, "pixel_shuffle_fwd: bad argument");
  SRHIP_REQUIRE(cout % 4 == 0, "pixel_shuffle_fwd: output channels must be a multiple of 4");
  long total = (long)n * h * r * w * r * (cout / 4);
  if (total == 0) return SRHIP_OK;
  if ((r == 2 || r == 3) && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0) {      // 16-byte accesses on both sides (block form)
    const long tb = (long)n * h * w * (cout / 4);
    if (r == 2)
      hipLaunchKernelGGL((pixel_shuffle_block_kernel<2, true>), dim3(cdiv(tb, 256)), dim3(256), 0, as_stream(stream), in, nullptr, out, n, h, w,
                         cout, slope, apply_act);
    else
      hipLaunchKernelGGL((pixel_shuffle_block_kernel<3, true>), dim3(cdiv(tb, 256)), dim3(256), 0, as_stream(stream), in, nullptr, out, n, h, w,
                         cout, slope, apply_act);
    return check_launch("pixel_shuffle_fwd");
  }
  hipLaunchKernelGGL(pixel_shuffle_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), in, out, n, h,
                     w, cout, r, slope, apply_act);
  return check_launch("pixel_shuffle_fwd");
}

int srhip_pixel_shuffle_bwd(const float* dout, const float* out, float* din, int n, int h, int w, int cout, int r,
                            float slope, int apply_act, void* stream) {
  SRHIP_REQUIRE(dout && din && (out || !apply_act) && n >= 0 && h > 0 && w > 0 && cout > 0 && r > 0,
                "pixel_shuffle_bwd: bad argument");
  SRHIP_REQUIRE(cout % 4 == 0, "pixel_shuffle_bwd: output channels must be a multiple of 4");
  SRHIP_REQUIRE((cout * r * r) % 4 == 0, "pixel_shuffle_bwd: input channels must be a multiple of 4");
  long total = (long)n * h * w * (cout * r * r / 4);
  if (total == 0) return SRHIP_OK;
  if ((r == 2 || r == 3) && ((((uintptr_t)dout) | ((uintptr_t)out) | ((uintptr_t)din)) & 15) == 0) {
    const long tb = (long)n * h * w * (cout / 4);
    if (r == 2)
      hipLaunchKernelGGL((pixel_shuffle_block_kernel<2, false>), dim3(cdiv(tb, 256)), dim3(256), 0, as_stream(stream), dout, out, din, n, h, w,
                         cout, slope, apply_act);
    else
      hipLaunchKernelGGL((pixel_shuffle_block_kernel<3, false>), dim3(cdiv(tb, 256)), dim3(256), 0, as_stream(stream), dout, out, din, n, h, w,
                         cout, slope, apply_act);
    return check_launch("pixel_shuffle_bwd");
  }
  hipLaunchKernelGGL(pixel_shuffle_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), dout, out, din,
                     n, h, w, cout, r, slope, apply_act);
  return check_launch("pixel_shuffle_bwd");
}

size_t srhip_colsum_workspace(long rows, int c) { return colsum_workspace_bytes(rows, c); }

int srhip_colsum(const float* dy, float* db, void* workspace, size_t workspace_bytes, long rows, int c, int ld,
                 void* stream) {
  SRHIP_REQUIRE(dy && db && rows > 0 && c > 0 && ld >= c, "colsum: bad argument");
  SRHIP_REQUIRE(c % 4 == 0 ? c <= 1024 : c <= 256, "colsum: too many columns");
  size_t need = colsum_workspace_bytes(rows, c);
  if (!workspace || workspace_bytes < need) {
    set_error("colsum: workspace %zu bytes < required %zu", workspace_bytes, need);
    return SRHIP_ERR_WORKSPACE;
  }
  return colsum_launch(dy, db, workspace, rows, c, ld, as_stream(stream));
}

int srhip_maxpool2x2_fwd(const float* x, float* y, int n, int h, int w, int c, void* stream) {
  SRHIP_REQUIRE(x && y && n > 0 && h >= 2 && w >= 2 && c >= 4, "maxpool2x2_fwd: bad argument");
  SRHIP_REQUIRE(c % 4 == 0 && h % 2 == 0 && w % 2 == 0, "maxpool2x2_fwd: C % 4 == 0 and even H, W only");
  const long total = (long)n * (h / 2) * (w / 2) * (c / 4);
  hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), x, y, n, h, w, c);
  return check_launch("maxpool2x2_fwd");
}

/* ABI 9: the pool forward that also leaves a 2-byte record per 4 output elements (arg-max position + "maximum > 0" per channel), and the
 * backward that reads the record instead of x: n * (h/2) * (w/2) * (c/4) records */
int srhip_maxpool2x2_fwd_idx(const float* x, float* y, void* rec, int n, int h, int w, int c, void* stream) {
  SRHIP_REQUIRE(x && y && rec && n > 0 && h >= 2 && w >= 2 && c >= 4, "maxpool2x2_fwd_idx: bad argument");
  SRHIP_REQUIRE(c % 4 == 0 && h % 2 == 0 && w % 2 == 0, "maxpool2x2_fwd_idx: C % 4 == 0 and even H, W only");
  const long total = (long)n * (h / 2) * (w / 2) * (c / 4);
  hipLaunchKernelGGL(maxpool2x2_fwd_idx_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), x, y, static_cast<unsigned short*>(rec),
                     n, h, w, c);
  return check_launch("maxpool2x2_fwd_idx");
}
int srhip_maxpool2x2_bwd_idx(const float* dy, const void* rec, float* dx, int n, int h, int w, int c, int relu_input, void* stream) {
  SRHIP_REQUIRE(dy && rec && dx && n > 0 && h >= 2 && w >= 2 && c >= 4, "maxpool2x2_bwd_idx: bad argument");
  SRHIP_REQUIRE(c % 4 == 0 && h % 2 == 0 && w % 2 == 0, "maxpool2x2_bwd_idx: C % 4 == 0 and even H, W only");
  const long total = (long)n * (h / 2) * (w / 2) * (c / 4);
  hipLaunchKernelGGL(maxpool2x2_bwd_idx_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), dy, static_cast<const unsigned short*>(rec),
                     dx, n, h, w, c, relu_input);
  return check_launch("maxpool2x2_bwd_idx");
}

int srhip_maxpool2x2_bwd(const float* dy, const float* x, float* dx, int n, int h, int w, int c, int relu_input,
                         void* stream) {
  SRHIP_REQUIRE(dy && x && dx && n > 0 && h >= 2 && w >= 2 && c >= 4, "maxpool2x2_bwd: bad argument");
  SRHIP_REQUIRE(c % 4 == 0 && h % 2 == 0 && w % 2 == 0, "maxpool2x2_bwd: C % 4 == 0 and even H, W only");
  const long total = (long)n * (h / 2) * (w / 2) * (c / 4);
  hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), dy, x, dx, n, h, w,
                     c, relu_input);
  return check_launch("maxpool2x2_bwd");
}

}  // extern "C"
