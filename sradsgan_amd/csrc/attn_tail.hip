// Fused local-attention tail of RAB / ResGroup (reference sradsgan.py:254-274, 303-323):
//     y = CLAM(u) = s[b,c] * u            s = sigmoid(MLP(avgpool u) + MLP(maxpool u))   (:117-127)
//     z = SLAM(y) = m[b,h,w] * y          m = sigmoid(conv7x7([mean_c y, max_c y]))       (:141-151)
//     out = conv1x1(z) + bias + skip                                                      (:262,:274)
// The reference runs ~15 elementwise/reduction launches and 5 extra HBM round trips per tail (x48 per
// generator forward).  Here y and z are never materialised: the 1x1 conv consumes u with the two
// scales folded into its A operand (per-image channel scale s) and its epilogue (per-pixel row scale
// m); what is left are HBM-bound passes over u, each one 16-byte coalesced NHWC reads with 16 lanes
// per pixel (4 channels per lane) and shuffle reductions inside the 16-lane group.
// All kernels: C == 64 channels (the generator's width), ld == C, roofline = HBM bandwidth.
#include "common.h"

namespace srhip {

constexpr int TC = 64;          // channels
constexpr int SEG = 32;         // pooling segments per image (32 x batch blocks: 8 left most CUs idle at B = 32)

__device__ inline float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
  return v;
}

// ---- F1a: per (image, segment) partial avg-sum / max / first-argmax over pixels ------------------ //
__global__ void clam_pool_partial_kernel(const float* __restrict__ u, float* __restrict__ psum,
                                         float* __restrict__ pmax, int* __restrict__ parg, int hw) {
  __shared__ float ssum[4][TC], smax[4][TC];
  __shared__ int sarg[4][TC];
  const int b = blockIdx.x / SEG, seg = blockIdx.x % SEG;
  const int per = (hw + SEG - 1) / SEG;
  const int p0 = seg * per, p1 = min(p0 + per, hw);
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;          // 4 row lanes x 64 channels
  const float* base = u + (size_t)b * hw * TC + c;
  float s = 0.f, mx = -INFINITY;
  int am = 0x7fffffff;
  for (int p = p0 + rl; p < p1; p += 4) {
    const float v = base[(size_t)p * TC];
    s += v;
    if (pool_takes(v, mx)) {                           // NaN propagates (common.h)
      mx = v;
      am = p;
    }
  }
  ssum[rl][c] = s;
  smax[rl][c] = mx;
  sarg[rl][c] = am;
  __syncthreads();
  if (rl == 0) {
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      s += ssum[k][c];
      const float v = smax[k][c];
      const int a = sarg[k][c];
      if (pool_merge_takes(v, a, mx, am)) {
        mx = v;
        am = a;
      }
    }
    const int o = (b * SEG + seg) * TC + c;
    psum[o] = s;
    pmax[o] = mx;
    parg[o] = am;
  }
}

// ---- F1b: combine segments, shared MLP 64 -> 4 -> 64 (no bias), sigmoid -> s[b,c] ---------------- //
// 256 threads per image: 4 lanes per channel walk the segments, 16 lanes per hidden unit do the 64-long dot products
// (one thread per channel with serial loops was latency-bound: 16 us for 64 x 4 numbers)
__global__ __launch_bounds__(256) void clam_mlp_kernel(const float* __restrict__ psum, const float* __restrict__ pmax,
                                                        const int* __restrict__ parg, const float* __restrict__ fc1,
                                                        const float* __restrict__ fc2, float* __restrict__ avg,
                                                        float* __restrict__ mx, int* __restrict__ arg,
                                                        float* __restrict__ s, int hw, int hidden) {
  __shared__ float sa[TC], sm[TC], ha[16], hm[16], qs[4][TC], qm[4][TC];
  __shared__ int qa[4][TC];
  const int b = blockIdx.x, c = threadIdx.x & 63, q = threadIdx.x >> 6;
  float sum = 0.f, m = -INFINITY;
  int am = 0x7fffffff;
  for (int k = q; k < SEG; k += 4) {
    const int o = (b * SEG + k) * TC + c;
    sum += psum[o];
    const float v = pmax[o];
    const int a = parg[o];
    if (pool_merge_takes(v, a, m, am)) {
      m = v;
      am = a;
    }
  }
  qs[q][c] = sum;
  qm[q][c] = m;
  qa[q][c] = am;
  __syncthreads();
  if (q == 0) {
    sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) sum += qs[k][c];      // segment order: (0,4,8,..) + (1,5,..) + ...
    m = qm[0][c];
    am = qa[0][c];
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      const float v = qm[k][c];
      const int a = qa[k][c];
      if (pool_merge_takes(v, a, m, am)) {
        m = v;
        am = a;
      }
    }
    const float a_ = sum / (float)hw;
    avg[b * TC + c] = a_;
    mx[b * TC + c] = m;
    arg[b * TC + c] = am;
    sa[c] = a_;
    sm[c] = m;
  }
  __syncthreads();
  {
    const int j = threadIdx.x >> 4, part = threadIdx.x & 15;   // hidden unit j (<= 16), 4 inputs per lane
    float x0 = 0.f, x1 = 0.f;
    if (j < hidden) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float w = fc1[j * TC + part * 4 + k];
        x0 += w * sa[part * 4 + k];
        x1 += w * sm[part * 4 + k];
      }
    }
    x0 = group16_sum(x0);
    x1 = group16_sum(x1);
    if (j < hidden && part == 0) {
      ha[j] = x0 < 0.f ? 0.f : x0;                     // ReLU that lets a NaN through like ATen's (fmaxf would return 0)
      hm[j] = x1 < 0.f ? 0.f : x1;
    }
  }
  __syncthreads();
  if (q == 0) {
    float l0 = 0.f, l1 = 0.f;
    for (int j = 0; j < hidden; ++j) {
      const float w = fc2[c * hidden + j];
      l0 += w * ha[j];
      l1 += w * hm[j];
    }
    const float l = l0 + l1;
    s[b * TC + c] = 1.f / (1.f + expf(-l));
  }
}

// ---- F2: pooled[pix] = (mean_c, max_c) of y = s*u, argc[pix] = first argmax channel -------------- //
__global__ void slam_pool_kernel(const float* __restrict__ u, const float* __restrict__ s,
                                 float2* __restrict__ pooled, int* __restrict__ argc, int hw, long npix) {
  const long pix = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int cq = threadIdx.x & 15;
  if (pix >= npix) return;
  const int b = (int)(pix / hw);
  const float4 v = *reinterpret_cast<const float4*>(u + pix * TC + cq * 4);
  const float4 sc = *reinterpret_cast<const float4*>(s + b * TC + cq * 4);
  const float y0 = v.x * sc.x, y1 = v.y * sc.y, y2 = v.z * sc.z, y3 = v.w * sc.w;
  float sum = (y0 + y1) + (y2 + y3);
  float mx = y0;
  int am = cq * 4;
  if (pool_takes(y1, mx)) { mx = y1; am = cq * 4 + 1; }
  if (pool_takes(y2, mx)) { mx = y2; am = cq * 4 + 2; }
  if (pool_takes(y3, mx)) { mx = y3; am = cq * 4 + 3; }
  sum = group16_sum(sum);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o, 16);
    const int oa = __shfl_xor(am, o, 16);
    if (pool_merge_takes(ov, oa, mx, am)) {
      mx = ov;
      am = oa;
    }
  }
  if (cq == 0) {
    pooled[pix] = make_float2(sum / (float)TC, mx);
    argc[pix] = am;
  }
}

// ---- F3: m = sigmoid(conv7x7 pad 3 (2 -> 1, no bias)(pooled)) ------------------------------------- //
__global__ void slam_conv7_kernel(const float2* __restrict__ pooled, const float* __restrict__ w7,
                                  float* __restrict__ m, int h, int w, long npix) {
  __shared__ float sw[98];
  if (threadIdx.x < 98) sw[threadIdx.x] = w7[threadIdx.x];        // [ch][kh][kw]
  __syncthreads();
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int hw = h * w;
  const long b = pix / hw;
  const int rem = (int)(pix - b * hw);
  const int y = rem / w, x = rem - y * w;
  const float2* img = pooled + b * hw;
  float acc = 0.f;
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    const int yy = y + kh - 3;
    if (yy < 0 || yy >= h) continue;
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const int xx = x + kw - 3;
      if (xx < 0 || xx >= w) continue;
      const float2 p = img[yy * w + xx];
      acc += sw[kh * 7 + kw] * p.x;
      acc += sw[49 + kh * 7 + kw] * p.y;
    }
  }
  m[pix] = 1.f / (1.f + expf(-acc));
}

// ---- B1: da[pix] = (sum_c dz*s*u) * m*(1-m)  (gradient at the 7x7 conv's output) ------------------ //
__global__ void tail_bwd_da_kernel(const float* __restrict__ dz, const float* __restrict__ u,
                                   const float* __restrict__ s, const float* __restrict__ m,
                                   float* __restrict__ da, int hw, long npix) {
  const long pix = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int cq = threadIdx.x & 15;
  if (pix >= npix) return;
  const int b = (int)(pix / hw);
  const float4 g = *reinterpret_cast<const float4*>(dz + pix * TC + cq * 4);
  const float4 v = *reinterpret_cast<const float4*>(u + pix * TC + cq * 4);
  const float4 sc = *reinterpret_cast<const float4*>(s + b * TC + cq * 4);
  float d = (g.x * (v.x * sc.x) + g.y * (v.y * sc.y)) + (g.z * (v.z * sc.z) + g.w * (v.w * sc.w));
  d = group16_sum(d);
  if (cq == 0) {
    const float mm = m[pix];
    da[pix] = d * mm * (1.f - mm);
  }
}

// ---- B2a: dpooled[pix][ch] = sum_{kh,kw} w7[ch][kh][kw] * da[(y-kh+3, x-kw+3)] --------------------- //
__global__ void slam_conv7_dgrad_kernel(const float* __restrict__ da, const float* __restrict__ w7,
                                        float2* __restrict__ dpooled, int h, int w, long npix) {
  __shared__ float sw[98];
  if (threadIdx.x < 98) sw[threadIdx.x] = w7[threadIdx.x];
  __syncthreads();
  const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int hw = h * w;
  const long b = pix / hw;
  const int rem = (int)(pix - b * hw);
  const int y = rem / w, x = rem - y * w;
  const float* img = da + b * hw;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    const int yy = y - kh + 3;
    if (yy < 0 || yy >= h) continue;
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const int xx = x - kw + 3;
      if (xx < 0 || xx >= w) continue;
      const float g = img[yy * w + xx];
      a0 += sw[kh * 7 + kw] * g;
      a1 += sw[49 + kh * 7 + kw] * g;
    }
  }
  dpooled[pix] = make_float2(a0, a1);
}

// ---- B2b: dw7[ch][kh][kw] = sum_pix da[pix] * pooled[(y+kh-3, x+kw-3)][ch] ---------------------------- //
// Every thread takes one pixel and all 98 taps (register accumulators), waves reduce with shuffles,
// the block writes one 98-vector of partials; slam_conv7_wgrad_reduce_kernel sums the block partials.
// A block stages a strip of W7_ROWS image rows (gradient + the 3-pixel halo of the pooled map) in LDS; thread
// (tap, half) then walks half of the strip's pixels with one multiply-add per pixel -- no per-thread tap array,
// no cross-lane reduction (the first version kept 98 accumulators per pixel-thread and spent its time in 98
// wave reductions: 53 us for 9 MFLOP).  part[tap][block] feeds slam_conv7_wgrad_reduce_kernel.
constexpr int W7_ROWS = 8;
__global__ __launch_bounds__(256) void slam_conv7_wgrad_kernel(const float* __restrict__ da,
                                                                const float2* __restrict__ pooled,
                                                                float* __restrict__ part, int h, int w, int strips) {
  extern __shared__ __attribute__((aligned(16))) float w7lds[];
  const int b = blockIdx.x / strips, sidx = blockIdx.x - b * strips;
  const int y0 = sidx * W7_ROWS;
  const int rows = min(W7_ROWS, h - y0);
  const int wp = w + 6;
  float* g = w7lds;                                   // [W7_ROWS][w]
  float2* pl = reinterpret_cast<float2*>(w7lds + W7_ROWS * w);   // [W7_ROWS + 6][w + 6], zero outside the image
  const size_t img = (size_t)b * h * w;
  for (int i = threadIdx.x; i < rows * w; i += 256) g[i] = da[img + (size_t)y0 * w + i];
  for (int i = threadIdx.x; i < (rows + 6) * wp; i += 256) {
    const int r = i / wp, c = i - r * wp;
    const int yy = y0 + r - 3, xx = c - 3;
    pl[i] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? pooled[img + (size_t)yy * w + xx] : make_float2(0.f, 0.f);
  }
  __syncthreads();
  const int t = threadIdx.x >> 1, half = threadIdx.x & 1;
  if (t < 98) {
    const int ch = t >= 49, tt = t - 49 * ch;
    const int kh = tt / 7, kw = tt - kh * 7;
    const float* src = reinterpret_cast<const float*>(pl) + ch;
    float acc = 0.f;
    for (int r = half; r < rows; r += 2) {
      const float* grow = g + r * w;
      const float* prow = src + 2 * ((r + kh) * wp + kw);
      for (int x = 0; x < w; ++x) acc += grow[x] * prow[2 * x];
    }
    acc += __shfl_xor(acc, 1, 64);
    if (half == 0) part[(size_t)t * gridDim.x + blockIdx.x] = acc;   // tap-major partials: the reduce reads them contiguously
  }
}
__global__ void slam_conv7_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw7, int nblk,
                                               int accumulate) {
  const int t = blockIdx.x;                          // one wave per tap
  float s = 0.f;
  for (int k = threadIdx.x; k < nblk; k += 64) s += part[(size_t)t * nblk + k];
  s = wave_sum(s);
  if (threadIdx.x == 0) dw7[t] = accumulate ? dw7[t] + s : s;
}

// ---- B3: dy = m*dz + dpooled.x/C + [c == argc]*dpooled.y ; du = s*dy ; ds partial = sum_pix dy*u --- //
// grid = (blocks_per_image, B); each block walks pixels of ONE image; 256 threads = 16 pixel lanes x 16
// channel quads; dsp[b][blk][c] holds the block's partial of ds.
// FOLD: the grid carries one extra block column that does slam_conv7_wgrad_reduce_kernel's work (same summation order per
// tap, the 98 taps dealt over the column's blocks and waves) -- one launch less in the serial chain of every tail.
template <bool FOLD>
__global__ void tail_bwd_main_kernel(const float* __restrict__ dz, const float* __restrict__ u,
                                     const float* __restrict__ s, const float* __restrict__ m,
                                     const float2* __restrict__ dpooled, const int* __restrict__ argc,
                                     float* __restrict__ du, float* __restrict__ dsp, int hw,
                                     const float* __restrict__ w7part, float* __restrict__ dw7, int w7blk, int acc7) {
  __shared__ float4 red[256];
  const int b = blockIdx.y, nblk = FOLD ? gridDim.x - 1 : gridDim.x;
  if (FOLD && blockIdx.x == nblk) {                 // the extra column: image b's block sums taps 4 b + wave, + 4 n, ...
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int t = b * 4 + wv; t < 98; t += 4 * (int)gridDim.y) {
      float a = 0.f;
      for (int k = lane; k < w7blk; k += 64) a += w7part[(size_t)t * w7blk + k];
      a = wave_sum(a);
      if (lane == 0) dw7[t] = acc7 ? dw7[t] + a : a;
    }
    return;
  }
  const int pl = threadIdx.x >> 4, cq = threadIdx.x & 15;
  const float4 sc = *reinterpret_cast<const float4*>(s + b * TC + cq * 4);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = blockIdx.x * 16 + pl; p < hw; p += nblk * 16) {
    const long pix = (long)b * hw + p;
    const float4 g = *reinterpret_cast<const float4*>(dz + pix * TC + cq * 4);
    const float4 v = *reinterpret_cast<const float4*>(u + pix * TC + cq * 4);
    const float mm = m[pix];
    const float2 dp = dpooled[pix];
    const int a = argc[pix] - cq * 4;
    const float dmean = dp.x * (1.f / (float)TC);
    float4 d;
    d.x = mm * g.x + dmean + (a == 0 ? dp.y : 0.f);
    d.y = mm * g.y + dmean + (a == 1 ? dp.y : 0.f);
    d.z = mm * g.z + dmean + (a == 2 ? dp.y : 0.f);
    d.w = mm * g.w + dmean + (a == 3 ? dp.y : 0.f);
    acc.x += d.x * v.x;
    acc.y += d.y * v.y;
    acc.z += d.z * v.z;
    acc.w += d.w * v.w;
    *reinterpret_cast<float4*>(du + pix * TC + cq * 4) = make_float4(sc.x * d.x, sc.y * d.y, sc.z * d.z, sc.w * d.w);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 o = red[k * 16 + cq];
      acc.x += o.x;
      acc.y += o.y;
      acc.z += o.z;
      acc.w += o.w;
    }
    *reinterpret_cast<float4*>(dsp + ((size_t)b * nblk + blockIdx.x) * TC + cq * 4) = acc;
  }
}

// ---- B4: ds[b,c] = sum over block partials --------------------------------------------------------- //
__global__ void tail_bwd_ds_kernel(const float* __restrict__ dsp, float* __restrict__ ds, int nblk) {
  const int b = blockIdx.x, c = threadIdx.x;
  float a = 0.f;
  for (int k = 0; k < nblk; ++k) a += dsp[((size_t)b * nblk + k) * TC + c];
  ds[b * TC + c] = a;
}

// ---- B5: du += davg[b,c]/HW + [pix == argmax_hw[b,c]] * dmax[b,c]  (in place) ----------------------- //
// pw1 != nullptr: blocks past the pixel range do clam_mlp_bwd_reduce_kernel's work (same order).
__global__ void tail_bwd_fix_kernel(float* __restrict__ du, const float* __restrict__ davg,
                                    const float* __restrict__ dmax, const int* __restrict__ arg, int hw, long npix,
                                    int pix_blocks, const float* __restrict__ pw1, const float* __restrict__ pw2,
                                    float* __restrict__ dfc1, float* __restrict__ dfc2, int n, int hidden, int accfc) {
  if ((int)blockIdx.x >= pix_blocks) {
    const int i = ((int)blockIdx.x - pix_blocks) * blockDim.x + threadIdx.x;
    const int per = hidden * TC;
    if (i >= 2 * per) return;
    const float* src = i < per ? pw1 + i : pw2 + (i - per);
    float acc = 0.f;
    for (int b = 0; b < n; ++b) acc += src[(size_t)b * per];
    float* o = i < per ? dfc1 + i : dfc2 + (i - per);
    *o = accfc ? *o + acc : acc;
    return;
  }
  const long pix = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int cq = threadIdx.x & 15;
  if (pix >= npix) return;
  const int b = (int)(pix / hw);
  const int p = (int)(pix - (long)b * hw);
  const float inv = 1.f / (float)hw;
  const float4 ga = *reinterpret_cast<const float4*>(davg + b * TC + cq * 4);
  const float4 gm = *reinterpret_cast<const float4*>(dmax + b * TC + cq * 4);
  const int4 am = *reinterpret_cast<const int4*>(arg + b * TC + cq * 4);
  float4 d = *reinterpret_cast<float4*>(du + pix * TC + cq * 4);
  d.x += ga.x * inv + (am.x == p ? gm.x : 0.f);
  d.y += ga.y * inv + (am.y == p ? gm.y : 0.f);
  d.z += ga.z * inv + (am.z == p ? gm.z : 0.f);
  d.w += ga.w * inv + (am.w == p ? gm.w : 0.f);
  *reinterpret_cast<float4*>(du + pix * TC + cq * 4) = d;
}

// ---- B4b: backward of s = sigmoid(W2 relu(W1 avg) + W2 relu(W1 max)) for one image per block ------- //
// ds [B][64] -> davg, dmax [B][64] and per-image partials of dW1 [H][64], dW2 [64][H] (summed by the
// reduce kernel below).  64 threads = 64 channels; H <= 16.
// nblk > 0: `ds` holds tail_bwd_main_kernel's block partials [b][nblk][c]; they are summed here in tail_bwd_ds_kernel's order.
__global__ void clam_mlp_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ avg,
                                    const float* __restrict__ mx, const float* __restrict__ s,
                                    const float* __restrict__ fc1, const float* __restrict__ fc2,
                                    float* __restrict__ davg, float* __restrict__ dmax, float* __restrict__ pw1,
                                    float* __restrict__ pw2, int hidden, int nblk) {
  __shared__ float sa[TC], sm[TC], sdl[TC], pa[16], pm[16], dpa[16], dpm[16];
  const int b = blockIdx.x, c = threadIdx.x;
  const float sv = s[b * TC + c];
  float dsv;
  if (nblk > 0) {
    dsv = 0.f;
#pragma unroll 16
    for (int k = 0; k < nblk; ++k) dsv += ds[((size_t)b * nblk + k) * TC + c];      // (loads in flight together, adds in order)
  } else {
    dsv = ds[b * TC + c];
  }
  const float dl = dsv * sv * (1.f - sv);
  const float a_ = avg[b * TC + c], m_ = mx[b * TC + c];
  sa[c] = a_;
  sm[c] = m_;
  sdl[c] = dl;
  __syncthreads();
  if (c < hidden) {
    float x0 = 0.f, x1 = 0.f, dh = 0.f;
    for (int k = 0; k < TC; ++k) {
      const float w = fc1[c * TC + k];
      x0 += w * sa[k];
      x1 += w * sm[k];
      dh += fc2[k * hidden + c] * sdl[k];
    }
    pa[c] = x0 < 0.f ? 0.f : x0;
    pm[c] = x1 < 0.f ? 0.f : x1;
    dpa[c] = x0 > 0.f ? dh : 0.f;
    dpm[c] = x1 > 0.f ? dh : 0.f;
  }
  __syncthreads();
  float ga = 0.f, gm = 0.f;
  for (int j = 0; j < hidden; ++j) {
    const float w = fc1[j * TC + c];
    ga += w * dpa[j];
    gm += w * dpm[j];
    pw1[((size_t)b * hidden + j) * TC + c] = dpa[j] * a_ + dpm[j] * m_;          // dW1[j][c]
    pw2[((size_t)b * TC + c) * hidden + j] = dl * (pa[j] + pm[j]);               // dW2[c][j]
  }
  davg[b * TC + c] = ga;
  dmax[b * TC + c] = gm;
}
__global__ void clam_mlp_bwd_reduce_kernel(const float* __restrict__ pw1, const float* __restrict__ pw2,
                                           float* __restrict__ dfc1, float* __restrict__ dfc2, int n, int hidden,
                                           int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int per = hidden * TC;
  if (i >= 2 * per) return;
  const float* src = i < per ? pw1 + i : pw2 + (i - per);
  float acc = 0.f;
  for (int b = 0; b < n; ++b) acc += src[(size_t)b * per];
  float* o = i < per ? dfc1 + i : dfc2 + (i - per);
  *o = accumulate ? *o + acc : acc;
}

constexpr int TAIL_BLK = 48;     // blocks per image in tail_bwd_main

}  // namespace srhip

using namespace srhip;

extern "C" {

size_t srhip_attn_tail_workspace(int n) { return (size_t)n * (SEG > TAIL_BLK ? SEG : TAIL_BLK) * TC * 3 * sizeof(float); }

int srhip_attn_tail_fwd(const float* u, const float* fc1, const float* fc2, const float* w7, float* avg, float* mx,
                        int* argmax_hw, float* s, float* pooled, int* argc, float* m, void* workspace,
                        size_t workspace_bytes, int n, int h, int w, int c, int hidden, void* stream) {
  SRHIP_REQUIRE(u && fc1 && fc2 && w7 && avg && mx && argmax_hw && s && pooled && argc && m, "attn_tail_fwd: null tensor");
  SRHIP_REQUIRE(c == TC && hidden >= 1 && hidden <= 16 && n > 0 && h > 0 && w > 0, "attn_tail_fwd: C must be 64, hidden <= 16");
  SRHIP_REQUIRE(workspace && workspace_bytes >= srhip_attn_tail_workspace(n), "attn_tail_fwd: workspace too small");
  hipStream_t st = as_stream(stream);
  const int hw = h * w;
  const long npix = (long)n * hw;
  float* psum = static_cast<float*>(workspace);
  float* pmax = psum + (size_t)n * SEG * TC;
  int* parg = reinterpret_cast<int*>(pmax + (size_t)n * SEG * TC);
  hipLaunchKernelGGL(clam_pool_partial_kernel, dim3(n * SEG), dim3(256), 0, st, u, psum, pmax, parg, hw);
  hipLaunchKernelGGL(clam_mlp_kernel, dim3(n), dim3(256), 0, st, psum, pmax, parg, fc1, fc2, avg, mx, argmax_hw, s, hw, hidden);
  hipLaunchKernelGGL(slam_pool_kernel, dim3(cdiv(npix, 16)), dim3(256), 0, st, u, s, reinterpret_cast<float2*>(pooled), argc, hw, npix);
  hipLaunchKernelGGL(slam_conv7_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, st, reinterpret_cast<const float2*>(pooled), w7, m, h, w, npix);
  return check_launch("attn_tail_fwd");
}

size_t srhip_attn_tail_bwd_workspace(int n, int h, int w) {
  return ((size_t)n * h * w * 3 + (size_t)n * TAIL_BLK * TC + (size_t)n * cdiv(h, W7_ROWS) * 98) * sizeof(float);
}

int srhip_attn_tail_bwd_spatial(const float* dz, const float* u, const float* s, const float* m, const float* pooled,
                                const int* argc, const float* w7, float* du, float* ds, float* dw7, int accumulate_dw7,
                                void* workspace, size_t workspace_bytes, int n, int h, int w, int c, void* stream) {
  SRHIP_REQUIRE(dz && u && s && m && pooled && argc && w7 && du && ds && dw7, "attn_tail_bwd_spatial: null tensor");
  SRHIP_REQUIRE(c == TC && n > 0 && h > 0 && w > 0, "attn_tail_bwd_spatial: C must be 64");
  const int hw = h * w;
  const long npix = (long)n * hw;
  const size_t need = srhip_attn_tail_bwd_workspace(n, h, w);
  SRHIP_REQUIRE(workspace && workspace_bytes >= need, "attn_tail_bwd_spatial: workspace too small");
  hipStream_t st = as_stream(stream);
  float* da = static_cast<float*>(workspace);
  float2* dpooled = reinterpret_cast<float2*>(da + npix);
  float* dsp = da + 3 * npix;
  float* w7part = dsp + (size_t)n * TAIL_BLK * TC;
  hipLaunchKernelGGL(tail_bwd_da_kernel, dim3(cdiv(npix, 16)), dim3(256), 0, st, dz, u, s, m, da, hw, npix);
  hipLaunchKernelGGL(slam_conv7_dgrad_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, st, da, w7, dpooled, h, w, npix);
  const int strips = (int)cdiv(h, W7_ROWS), w7blk = n * strips;
  const size_t w7lds = ((size_t)W7_ROWS * w + 2 * (size_t)(W7_ROWS + 6) * (w + 6)) * sizeof(float);
  SRHIP_REQUIRE(w7lds <= 64 * 1024, "attn_tail_bwd_spatial: image too wide for the 7x7 weight-gradient strip");
  hipLaunchKernelGGL(slam_conv7_wgrad_kernel, dim3(w7blk), dim3(256), w7lds, st, da, reinterpret_cast<const float2*>(pooled), w7part, h, w, strips);
  hipLaunchKernelGGL(slam_conv7_wgrad_reduce_kernel, dim3(98), dim3(64), 0, st, w7part, dw7, w7blk, accumulate_dw7);
  hipLaunchKernelGGL(tail_bwd_main_kernel<false>, dim3(TAIL_BLK, n), dim3(256), 0, st, dz, u, s, m, dpooled, argc, du, dsp, hw, nullptr, nullptr, 0, 0);
  hipLaunchKernelGGL(tail_bwd_ds_kernel, dim3(n), dim3(TC), 0, st, dsp, ds, TAIL_BLK);
  return check_launch("attn_tail_bwd_spatial");
}



int srhip_attn_tail_bwd_channel(float* du, const float* davg, const float* dmax, const int* argmax_hw, int n, int h,
                                int w, int c, void* stream) {
  SRHIP_REQUIRE(du && davg && dmax && argmax_hw && c == TC && n > 0 && h > 0 && w > 0, "attn_tail_bwd_channel: bad argument");
  const int hw = h * w;
  const long npix = (long)n * hw;
  hipLaunchKernelGGL(tail_bwd_fix_kernel, dim3(cdiv(npix, 16)), dim3(256), 0, as_stream(stream), du, davg, dmax, argmax_hw, hw, npix,
                     (int)cdiv(npix, 16), nullptr, nullptr, nullptr, nullptr, 0, 0, 0);
  return check_launch("attn_tail_bwd_channel");
}

// The whole backward of the tail behind the 1x1 conv's data gradient in ONE call and 7 launches (the three entry points
// above: 10): the 7x7 weight-gradient reduce rides in an extra block column of the main pass, the sum of the ds partials
// in the MLP-backward kernel, the MLP weight-gradient reduce in extra blocks of the final fix-up pass.  Same arithmetic
// and summation orders as the separate kernels (bit-identical results).
size_t srhip_attn_tail_bwd_fused_workspace(int n, int h, int w, int hidden) {
  return srhip_attn_tail_bwd_workspace(n, h, w) + ((size_t)n * 2 * hidden * TC + 2 * (size_t)n * TC) * sizeof(float);
}

int srhip_attn_tail_bwd(const float* dz, const float* u, const float* s, const float* m, const float* pooled, const int* argc,
                        const float* avg, const float* mx, const int* argmax_hw, const float* w7, const float* fc1,
                        const float* fc2, float* du, float* dw7, int accumulate_dw7, float* dfc1, float* dfc2,
                        int accumulate_dfc, void* workspace, size_t workspace_bytes, int n, int h, int w, int c, int hidden,
                        void* stream) {
  SRHIP_REQUIRE(dz && u && s && m && pooled && argc && avg && mx && argmax_hw && w7 && fc1 && fc2 && du && dw7 && dfc1 && dfc2,
                "attn_tail_bwd: null tensor");
  SRHIP_REQUIRE(c == TC && hidden >= 1 && hidden <= 16 && n > 0 && h > 0 && w > 0, "attn_tail_bwd: C must be 64, hidden <= 16");
  SRHIP_REQUIRE(workspace && workspace_bytes >= srhip_attn_tail_bwd_fused_workspace(n, h, w, hidden), "attn_tail_bwd: workspace too small");
  const int hw = h * w;
  const long npix = (long)n * hw;
  hipStream_t st = as_stream(stream);
  float* da = static_cast<float*>(workspace);
  float2* dpooled = reinterpret_cast<float2*>(da + npix);
  float* dsp = da + 3 * npix;
  float* w7part = dsp + (size_t)n * TAIL_BLK * TC;
  const int strips = (int)cdiv(h, W7_ROWS), w7blk = n * strips;
  float* pw1 = w7part + (size_t)w7blk * 98;
  float* pw2 = pw1 + (size_t)n * hidden * TC;
  float* davg = pw2 + (size_t)n * hidden * TC;
  float* dmax = davg + (size_t)n * TC;
  const size_t w7lds = ((size_t)W7_ROWS * w + 2 * (size_t)(W7_ROWS + 6) * (w + 6)) * sizeof(float);
  SRHIP_REQUIRE(w7lds <= 64 * 1024, "attn_tail_bwd: image too wide for the 7x7 weight-gradient strip");
  hipLaunchKernelGGL(tail_bwd_da_kernel, dim3(cdiv(npix, 16)), dim3(256), 0, st, dz, u, s, m, da, hw, npix);
  hipLaunchKernelGGL(slam_conv7_dgrad_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, st, da, w7, dpooled, h, w, npix);
  hipLaunchKernelGGL(slam_conv7_wgrad_kernel, dim3(w7blk), dim3(256), w7lds, st, da, reinterpret_cast<const float2*>(pooled), w7part, h, w, strips);
  hipLaunchKernelGGL(tail_bwd_main_kernel<true>, dim3(TAIL_BLK + 1, n), dim3(256), 0, st, dz, u, s, m, dpooled, argc, du, dsp, hw,
                     w7part, dw7, w7blk, accumulate_dw7);
  hipLaunchKernelGGL(clam_mlp_bwd_kernel, dim3(n), dim3(TC), 0, st, dsp, avg, mx, s, fc1, fc2, davg, dmax, pw1, pw2, hidden, TAIL_BLK);
  const int pix_blocks = (int)cdiv(npix, 16), red_blocks = (int)cdiv(2 * hidden * TC, 256);
  hipLaunchKernelGGL(tail_bwd_fix_kernel, dim3(pix_blocks + red_blocks), dim3(256), 0, st, du, davg, dmax, argmax_hw, hw, npix, pix_blocks,
                     pw1, pw2, dfc1, dfc2, n, hidden, accumulate_dfc);
  return check_launch("attn_tail_bwd");
}

size_t srhip_attn_tail_mlp_workspace(int n, int hidden) { return (size_t)n * 2 * hidden * TC * sizeof(float); }

int srhip_attn_tail_bwd_mlp(const float* ds, const float* avg, const float* mx, const float* s, const float* fc1,
                            const float* fc2, float* davg, float* dmax, float* dfc1, float* dfc2, int accumulate_dfc,
                            void* workspace, size_t workspace_bytes, int n, int c, int hidden, void* stream) {
  SRHIP_REQUIRE(ds && avg && mx && s && fc1 && fc2 && davg && dmax && dfc1 && dfc2, "attn_tail_bwd_mlp: null tensor");
  SRHIP_REQUIRE(c == TC && hidden >= 1 && hidden <= 16 && n > 0, "attn_tail_bwd_mlp: C must be 64, hidden <= 16");
  SRHIP_REQUIRE(workspace && workspace_bytes >= srhip_attn_tail_mlp_workspace(n, hidden), "attn_tail_bwd_mlp: workspace too small");
  hipStream_t st = as_stream(stream);
  float* pw1 = static_cast<float*>(workspace);
  float* pw2 = pw1 + (size_t)n * hidden * TC;
  hipLaunchKernelGGL(clam_mlp_bwd_kernel, dim3(n), dim3(TC), 0, st, ds, avg, mx, s, fc1, fc2, davg, dmax, pw1, pw2, hidden, 0);
  hipLaunchKernelGGL(clam_mlp_bwd_reduce_kernel, dim3(cdiv(2 * hidden * TC, 256)), dim3(256), 0, st, pw1, pw2, dfc1, dfc2, n, hidden, accumulate_dfc);
  return check_launch("attn_tail_bwd_mlp");
}

}  // extern "C"
