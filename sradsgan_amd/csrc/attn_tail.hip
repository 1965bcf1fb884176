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

// value of lane (l + N) mod 16 of this lane's 16-lane row (DPP row_ror: one VALU instruction, no LDS round trip)
template <int N>
__device__ __forceinline__ float row_ror16(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
}
// Sum over the 16 lanes of a row, every lane gets the total.  Round 4: rotations instead of the xor butterfly of __shfl_xor
// (ds_bpermute: an LDS round trip per step; the fused inference tail does 160 of them per thread).  Bit-identical to the
// butterfly: after step k all lanes that differ only in the bits already folded hold the same value (a + b == b + a), and
// (l + o) mod 16 lies in the same class as l ^ o, so every lane adds the same two numbers in the same order as before.
__device__ inline float group16_sum(float v) {
  v += row_ror16<8>(v);
  v += row_ror16<4>(v);
  v += row_ror16<2>(v);
  v += row_ror16<1>(v);
  return v;
}
// NaN-propagating maximum over the 16 lanes of a row (value only), same argument
__device__ inline float group16_max(float v) {
  float o;
  o = row_ror16<8>(v); if (pool_takes(o, v)) v = o;
  o = row_ror16<4>(v); if (pool_takes(o, v)) v = o;
  o = row_ror16<2>(v); if (pool_takes(o, v)) v = o;
  o = row_ror16<1>(v); if (pool_takes(o, v)) v = o;
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
  int p = p0 + rl;
  for (; p + 28 < p1; p += 32) {                     // eight loads in flight per lane (the loop was one dependent-latency load per pass:
    float v[8];                                      // 9 us for 12 MB at B = 16); same visiting order, same sums
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = base[(size_t)(p + 4 * j) * TC];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s += v[j];
      if (pool_takes(v[j], mx)) {                      // NaN propagates (common.h)
        mx = v[j];
        am = p + 4 * j;
      }
    }
  }
  for (; p < p1; p += 4) {
    const float v = base[(size_t)p * TC];
    s += v;
    if (pool_takes(v, mx)) {
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
                                                        float* __restrict__ s, int hw, int hidden, int nseg) {
  __shared__ float sa[TC], sm[TC], ha[16], hm[16], qs[4][TC], qm[4][TC];
  __shared__ int qa[4][TC];
  const int b = blockIdx.x, c = threadIdx.x & 63, q = threadIdx.x >> 6;
  float sum = 0.f, m = -INFINITY;
  int am = 0x7fffffff;
  for (int k = q; k < nseg; k += 4) {                // nseg: SEG from clam_pool_partial_kernel, 2 x tiles per image from the conv epilogue
    const int o = (b * nseg + k) * TC + c;
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
  // Branch-free taps (round 4): a tap outside the image reads a clamped address and contributes w * 0 -- the same sum as
  // skipping it (acc starts at +0, weights are finite) -- so the 49 loads of a thread are issued together instead of one
  // dependent L1 round trip per iteration behind a divergent branch (10.4 -> ~3 us at B = 32)
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {
    const int yy = y + kh - 3;
    const bool oky = yy >= 0 && yy < h;
    const int yc = min(max(yy, 0), h - 1);
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const int xx = x + kw - 3;
      const bool ok = oky && xx >= 0 && xx < w;
      const int xc = min(max(xx, 0), w - 1);
      float2 p = img[yc * w + xc];
      if (!ok) p = make_float2(0.f, 0.f);
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
__device__ __forceinline__ void slam_conv7_dgrad_block(const float* __restrict__ da, const float* __restrict__ w7,
                                                       float2* __restrict__ dpooled, int h, int w, long npix, int bid) {
  __shared__ float sw[98];
  if (threadIdx.x < 98) sw[threadIdx.x] = w7[threadIdx.x];
  __syncthreads();
  const long pix = (long)bid * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int hw = h * w;
  const long b = pix / hw;
  const int rem = (int)(pix - b * hw);
  const int y = rem / w, x = rem - y * w;
  const float* img = da + b * hw;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int kh = 0; kh < 7; ++kh) {                   // branch-free like slam_conv7_kernel: 49 independent loads per thread
    const int yy = y - kh + 3;
    const bool oky = yy >= 0 && yy < h;
    const int yc = min(max(yy, 0), h - 1);
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
      const int xx = x - kw + 3;
      const bool ok = oky && xx >= 0 && xx < w;
      const int xc = min(max(xx, 0), w - 1);
      float g = img[yc * w + xc];
      if (!ok) g = 0.f;
      a0 += sw[kh * 7 + kw] * g;
      a1 += sw[49 + kh * 7 + kw] * g;
    }
  }
  dpooled[pix] = make_float2(a0, a1);
}
__global__ void slam_conv7_dgrad_kernel(const float* __restrict__ da, const float* __restrict__ w7,
                                        float2* __restrict__ dpooled, int h, int w, long npix) {
  slam_conv7_dgrad_block(da, w7, dpooled, h, w, npix, blockIdx.x);
}

// ---- B2b: dw7[ch][kh][kw] = sum_pix da[pix] * pooled[(y+kh-3, x+kw-3)][ch] ---------------------------- //
// Every thread takes one pixel and all 98 taps (register accumulators), waves reduce with shuffles,
// the block writes one 98-vector of partials; slam_conv7_wgrad_reduce_kernel sums the block partials.
// A block stages a strip of W7_ROWS image rows (gradient + the 3-pixel halo of the pooled map) in LDS; thread
// (tap, half) then walks half of the strip's pixels with one multiply-add per pixel -- no per-thread tap array,
// no cross-lane reduction (the first version kept 98 accumulators per pixel-thread and spent its time in 98
// wave reductions: 53 us for 9 MFLOP).  part[tap][block] feeds slam_conv7_wgrad_reduce_kernel.
constexpr int W7_ROWS = 8;
__device__ __forceinline__ void slam_conv7_wgrad_block(const float* __restrict__ da, const float2* __restrict__ pooled,
                                                       float* __restrict__ part, int h, int w, int strips, int bid, int nblk) {
  extern __shared__ __attribute__((aligned(16))) float w7lds[];
  const int b = bid / strips, sidx = bid - b * strips;
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
    if (half == 0) part[(size_t)t * nblk + bid] = acc;   // tap-major partials: the reduce reads them contiguously
  }
}
__global__ __launch_bounds__(256) void slam_conv7_wgrad_kernel(const float* __restrict__ da,
                                                                const float2* __restrict__ pooled,
                                                                float* __restrict__ part, int h, int w, int strips) {
  slam_conv7_wgrad_block(da, pooled, part, h, w, strips, blockIdx.x, gridDim.x);
}
// Round 4: both halves of the 7x7 conv's backward behind ONE launch (they only share their input `da`): blocks [0, nd) do the data
// gradient, blocks [nd, nd + nw) the weight-gradient strips.  Inside the step every launch of the tail's serial chain waits for
// block slots the weight-gradient stream holds (13-29 us per launch in the step for kernels that take 4-10 us alone).
__global__ __launch_bounds__(256) void slam_conv7_bwd_kernel(const float* __restrict__ da, const float* __restrict__ w7,
                                                              const float2* __restrict__ pooled, float2* __restrict__ dpooled,
                                                              float* __restrict__ part, int h, int w, long npix, int strips,
                                                              int nd, int nw) {
  if ((int)blockIdx.x < nd) slam_conv7_dgrad_block(da, w7, dpooled, h, w, npix, blockIdx.x);
  else slam_conv7_wgrad_block(da, pooled, part, h, w, strips, (int)blockIdx.x - nd, nw);
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

// ---- B3' (round 6): the main pass WITH the 7x7 conv's data gradient inside -------------------------------------------------------- //
// slam_conv7_bwd_kernel sat in every tail's serial chain between the da pass and this one (8 us alone, 18 us inside the training step:
// its launch waits for block slots like every other) only to turn da into dpooled = conv_transpose7x7(da, w7) -- 98 multiply-adds per
// pixel on a 4-byte-per-pixel map.  Here a block owns a CONTIGUOUS range of 16-pixel groups of one image, stages the da rows of that
// range plus the 3-row / 3-column halo in LDS (zero outside the image) and the 16 lanes that share a pixel split the 49 taps between
// them (lane q: taps q, q + 16, q + 32, q + 48; one LDS read serves both channels of the pooled map), group16_sum gives every lane
// dpooled of its pixel.  The rest is tail_bwd_main_kernel.  The grid's extra block columns [nblk, nblk + strips) run the 7x7 conv's
// weight-gradient strips (slam_conv7_wgrad_block, unchanged); their partials are summed by extra blocks of tail_bwd_fix_kernel
// (w7part != nullptr there), in slam_conv7_wgrad_reduce_kernel's order.  dpooled differs from slam_conv7_dgrad_block's by summation
// order only (16 lane partials of <= 4 taps instead of one chain of 49).
__global__ __launch_bounds__(256) void tail_bwd_main2_kernel(const float* __restrict__ dz, const float* __restrict__ u,
                                                              const float* __restrict__ s, const float* __restrict__ m,
                                                              const float* __restrict__ da, const float* __restrict__ w7,
                                                              const float2* __restrict__ pooled, const int* __restrict__ argc,
                                                              float* __restrict__ du, float* __restrict__ dsp, float* __restrict__ w7part,
                                                              int h, int w, int nblk, int strips, int gpb, int tile_rows) {
  extern __shared__ __attribute__((aligned(16))) float w7lds[];
  __shared__ float4 red[256];
  const int b = blockIdx.y, hw = h * w;
  if ((int)blockIdx.x >= nblk) {                      // weight-gradient strip (int)blockIdx.x - nblk of image b
    slam_conv7_wgrad_block(da, pooled, w7part, h, w, strips, b * strips + ((int)blockIdx.x - nblk), (int)gridDim.y * strips);
    return;
  }
  const int pl = threadIdx.x >> 4, cq = threadIdx.x & 15;
  const int ngroups = (hw + 15) >> 4;
  const int g0 = blockIdx.x * gpb, g1 = min(g0 + gpb, ngroups);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g0 < g1) {                                      // (block-uniform)
    // ---- da rows [ya - 3, yb + 3] x columns [-3, w + 3) of image b -> LDS, zero outside the image ----
    const int p0 = g0 * 16, p1 = min(g1 * 16, hw);
    const int ya = p0 / w, yb = (p1 - 1) / w;
    const int wp = w + 6, rows = yb - ya + 7;          // <= tile_rows
    const float* img = da + (size_t)b * hw;
    for (int i = threadIdx.x; i < rows * wp; i += 256) {
      const int r = i / wp, c = i - r * wp;
      const int yy = ya - 3 + r, xx = c - 3;
      w7lds[i] = (yy >= 0 && yy < h && xx >= 0 && xx < w) ? img[yy * w + xx] : 0.f;
    }
    // this lane's taps: t = cq + 16 k (k = 0..3, t < 49): weight of both pooled channels and the tile offset relative to the pixel
    float w0[4], w1[4];
    int toff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int t = cq + 16 * k;
      const bool ok = t < 49;
      const int tt = ok ? t : 0, kh = tt / 7, kw = tt - kh * 7;
      w0[k] = ok ? w7[tt] : 0.f;
      w1[k] = ok ? w7[49 + tt] : 0.f;
      toff[k] = (3 - kh) * wp + (3 - kw);              // da[(y - kh + 3, x - kw + 3)] relative to tile position of (y, x)
    }
    const float4 sc = *reinterpret_cast<const float4*>(s + b * TC + cq * 4);
    __syncthreads();
    for (int grp = g0; grp < g1; ++grp) {
      const int p = grp * 16 + pl;
      const bool ok = p < hw;                          // (uniform over the 16 lanes of a pixel)
      const int pc = ok ? p : p0;
      const int y = pc / w, x = pc - y * w;
      const float* ctr = w7lds + (y - ya + 3) * wp + (x + 3);
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float g = ctr[toff[k]];
        a0 += w0[k] * g;
        a1 += w1[k] * g;
      }
      a0 = group16_sum(a0);
      a1 = group16_sum(a1);
      if (!ok) continue;
      const long pix = (long)b * hw + p;
      const float4 g = *reinterpret_cast<const float4*>(dz + pix * TC + cq * 4);
      const float4 v = *reinterpret_cast<const float4*>(u + pix * TC + cq * 4);
      const float mm = m[pix];
      const int a = argc[pix] - cq * 4;
      const float dmean = a0 * (1.f / (float)TC);
      float4 d;
      d.x = mm * g.x + dmean + (a == 0 ? a1 : 0.f);
      d.y = mm * g.y + dmean + (a == 1 ? a1 : 0.f);
      d.z = mm * g.z + dmean + (a == 2 ? a1 : 0.f);
      d.w = mm * g.w + dmean + (a == 3 ? a1 : 0.f);
      acc.x += d.x * v.x;
      acc.y += d.y * v.y;
      acc.z += d.z * v.z;
      acc.w += d.w * v.w;
      *reinterpret_cast<float4*>(du + pix * TC + cq * 4) = make_float4(sc.x * d.x, sc.y * d.y, sc.z * d.z, sc.w * d.w);
    }
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
                                    float* __restrict__ dfc1, float* __restrict__ dfc2, int n, int hidden, int accfc,
                                    unsigned* __restrict__ du_pp = nullptr, int wd = 0, int guard = 0, int red_blocks = 1 << 30,
                                    const float* __restrict__ w7part = nullptr, float* __restrict__ dw7 = nullptr, int w7blk = 0,
                                    int acc7 = 0) {
  // du_pp (round 5): the final du leaves as padded split-bf16 planes instead of fp32 (csrc/conv_wgrad_flat.hip: pixel row of 64 * 4 bytes,
  // per 8 channels 8 hi | 8 lo halves): the RAB's conv2 data gradient and weight gradient read it without a conversion pass
  if ((int)blockIdx.x >= pix_blocks + red_blocks) {      // (round 6, behind tail_bwd_main2_kernel) slam_conv7_wgrad_reduce_kernel's sums: one wave per tap
    const int t = ((int)blockIdx.x - pix_blocks - red_blocks) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= 98) return;
    float a = 0.f;
    for (int k = lane; k < w7blk; k += 64) a += w7part[(size_t)t * w7blk + k];
    a = wave_sum(a);
    if (lane == 0) dw7[t] = acc7 ? dw7[t] + a : a;
    return;
  }
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
  if (du_pp == nullptr) {                               // (with planes wanted the fp32 tensor stays the main pass's partial result: 24 MB less written per tail)
    *reinterpret_cast<float4*>(du + pix * TC + cq * 4) = d;
  } else {
    const int y = p / wd, x = p - y * wd;
    const long row = guard + ((long)b * (hw / wd + 1) + y) * (wd + 1) + x;
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t h01 = {(__bf16)d.x, (__bf16)d.y}, h23 = {(__bf16)d.z, (__bf16)d.w};
    const unsigned uh01 = __builtin_bit_cast(unsigned, h01), uh23 = __builtin_bit_cast(unsigned, h23);
    const bf16x2_t l01 = {(__bf16)(d.x - __uint_as_float(uh01 << 16)), (__bf16)(d.y - __uint_as_float(uh01 & 0xffff0000u))};
    const bf16x2_t l23 = {(__bf16)(d.z - __uint_as_float(uh23 << 16)), (__bf16)(d.w - __uint_as_float(uh23 & 0xffff0000u))};
    // the two lanes of an octet trade halves: the even lane stores the octet's 8 hi halves, the odd lane its 8 lo halves, 16 bytes each
    // (whole rows per store instruction; see epi_finish in conv_fast.hip)
    const bool odd = (cq & 1) != 0;
    const unsigned ul01 = __builtin_bit_cast(unsigned, l01), ul23 = __builtin_bit_cast(unsigned, l23);
    const unsigned r0 = pair_swap(odd ? uh01 : ul01), r1 = pair_swap(odd ? uh23 : ul23);
    unsigned* o = du_pp + row * TC + (cq >> 1) * 8 + (odd ? 4 : 0);
    *reinterpret_cast<uint4*>(o) = odd ? make_uint4(r0, r1, ul01, ul23) : make_uint4(uh01, uh23, r0, r1);
  }
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

// ================================================================================================ //
// Inference form of the whole tail (round 4): ONE kernel after the pooling partials.
//     s = sigmoid(MLP(avg) + MLP(max))  ->  pooled = (mean_c, max_c)(s*u)  ->  m = sigmoid(conv7x7(pooled))
//     out = m * (Wc (s*u)) + bc + skip
// With grad mode off nothing has to be saved, so the four launches of srhip_attn_tail_fwd's second half and the 1x1 conv collapse
// into one: a block owns a PH x PW tile of pixels (<= 128), recomputes s for its image from the pooling partials (64 x hidden
// multiply-adds), pools the tile plus the 3-pixel halo the 7x7 conv needs (u is read 2.5x: 12 MB at B = 16, L2 / Infinity-Cache
// resident right behind conv2), convolves, and runs the 64 x 64 1x1 conv of its 128 pixels on the MFMA in split-bf16 with the
// weights as the row operand (a lane ends up with 4 consecutive channels of one pixel).  Every step keeps the arithmetic and the
// operation order of the training-mode kernels (clam_mlp_kernel, slam_pool_kernel, slam_conv7_kernel,
// fast_conv_dma_kernel<128, 64, bias|residual|rowscale|chanscale, split-bf16>): the output is bit-identical to the training forward.
// ================================================================================================ //
int g_tail_dbg = 0;     // srhip_debug_set(7, bits): timing-only ablations of attn_tail_eval_kernel (1 pooling, 2 7x7 conv, 4 MFMA, 8 epilogue)
typedef __bf16 tl_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tl_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int tl_u32x2 __attribute__((ext_vector_type(2)));
constexpr int EV_PITCH = 272;                       // bytes per 64-channel row of the split images (256 + 16: conflict-free b128 reads)
constexpr int EV_MAXREG = 320;                      // pixels of the pooled region (tile + halo): 20 register slots of 16 bytes per thread

__global__ __launch_bounds__(256, 2) void attn_tail_eval_kernel(const float* __restrict__ u, const float* __restrict__ skip,
                                                             const float* __restrict__ psum, const float* __restrict__ pmax,
                                                             const float* __restrict__ fc1, const float* __restrict__ fc2,
                                                             const float* __restrict__ w7, const float* __restrict__ wsplit,
                                                             const float* __restrict__ bc, float* __restrict__ out, int h, int w,
                                                             int hidden, int PH, int PW, int tiles_h, int tiles_w, int dbg, int nseg) {
  __shared__ __attribute__((aligned(16))) char a_img[128 * EV_PITCH];      // s*u of the tile, split hi|lo per 8 channels
  __shared__ __attribute__((aligned(16))) char w_img[TC * EV_PITCH];       // Wc, split when it was packed
  __shared__ float2 pooled_s[EV_MAXREG];
  __shared__ __attribute__((aligned(16))) float s_s[TC];
  __shared__ float sa[TC], sm[TC], ha[16], hm[16], qs[4][TC], qm[4][TC], m_s[128], sw[98];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tpi = tiles_h * tiles_w;
  // XCD-aware tile order (block i runs on XCD i % 8): each XCD walks a contiguous range of tiles, so the halo pixels two
  // neighbouring tiles both read come out of that XCD's L2
  int vb;
  {
    const int nblk = (int)gridDim.x, qd = nblk >> 3, rd = nblk & 7, xc = (int)blockIdx.x & 7;
    vb = (xc < rd ? xc * (qd + 1) : rd * (qd + 1) + (xc - rd) * qd) + ((int)blockIdx.x >> 3);
  }
  const int b = vb / tpi, trem = vb - b * tpi;
  const int ty = trem / tiles_w, tx = trem - ty * tiles_w;
  const int oh0 = ty * PH, ow0 = tx * PW;
  const int hw = h * w;

  // ---- everything that does not depend on s is fetched first: the 1x1 conv's weights (64 rows x 256 B of the packed split section
  // -> LDS, pitch 272), the 7x7 taps, and this thread's share of u over the tile + halo (up to 32 pixel slots x 16 bytes in
  // registers): the MLP below runs under their latency ----
  const int RH = PH + 6, RW = PW + 6, nreg = RH * RW;
  const int cq = tid & 15;
  constexpr int NSLOT = EV_MAXREG / 16;              // 16 pixels per pass of the 256 threads
  // q / RW for q < 512 by a float multiply (exact: (q + 0.5) / RW is never within 0.5 / RW of an integer): an integer division
  // costs ~40 VALU instructions and there are two per pixel slot
  const float inv_rw = 1.0f / (float)RW, inv_pw = 1.0f / (float)PW;
  float4 uv[NSLOT];
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int qq = i * 16 + (tid >> 4);
    const int ry = (int)(((float)qq + 0.5f) * inv_rw), rx = qq - ry * RW;
    const int y = oh0 - 3 + ry, x = ow0 - 3 + rx;
    const bool in = qq < nreg && y >= 0 && y < h && x >= 0 && x < w;
    uv[i] = in ? *reinterpret_cast<const float4*>(u + ((size_t)b * hw + (size_t)y * w + x) * TC + cq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = i * 256 + tid;                 // 16-byte piece: row = piece / 16, quad = piece % 16
    const float4 v = *reinterpret_cast<const float4*>(wsplit + piece * 4);
    *reinterpret_cast<float4*>(w_img + (piece >> 4) * EV_PITCH + (piece & 15) * 16) = v;
  }
  if (tid < 98) sw[tid] = w7[tid];

  // ---- s: clam_mlp_kernel's arithmetic on this image's pooling partials; every global operand is fetched before the first
  // dependent instruction (the partials, this thread's fc1 row quad, this channel's fc2 row) ----
  {
    const int c = tid & 63, q = tid >> 6;
    constexpr int PSEG = 64 / 4;                      // up to 64 partial segments per image (POOL_MAXSEG), nseg of them live
    float pv[PSEG], pm[PSEG];
#pragma unroll
    for (int i = 0; i < PSEG; ++i) {
      const bool live = q + 4 * i < nseg;
      const int o = (b * nseg + (live ? q + 4 * i : 0)) * TC + c;
      pv[i] = live ? psum[o] : 0.f;
      pm[i] = live ? pmax[o] : -INFINITY;
    }
    const int j = tid >> 4, part = tid & 15;
    float w1[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < hidden) {
#pragma unroll
      for (int k = 0; k < 4; ++k) w1[k] = fc1[j * TC + part * 4 + k];
    }
    float w2[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) w2[jj] = (q == 0 && jj < hidden) ? fc2[c * hidden + jj] : 0.f;
    float sum = 0.f, m = -INFINITY;
#pragma unroll
    for (int i = 0; i < PSEG; ++i) {
      if (q + 4 * i < nseg) sum += pv[i];            // (the same additions in the same order as clam_mlp_kernel's loop)
      if (pool_takes(pm[i], m)) m = pm[i];           // values only matter here: a NaN wins, else the maximum
    }
    qs[q][c] = sum;
    qm[q][c] = m;
    __syncthreads();
    if (q == 0) {
      sum = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) sum += qs[k][c];
      m = qm[0][c];
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        const float v = qm[k][c];
        if (pool_takes(v, m)) m = v;
      }
      sa[c] = sum / (float)hw;
      sm[c] = m;
    }
    __syncthreads();
    {
      float x0 = 0.f, x1 = 0.f;
      if (j < hidden) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          x0 += w1[k] * sa[part * 4 + k];
          x1 += w1[k] * sm[part * 4 + k];
        }
      }
      x0 = group16_sum(x0);
      x1 = group16_sum(x1);
      if (j < hidden && part == 0) {
        ha[j] = x0 < 0.f ? 0.f : x0;
        hm[j] = x1 < 0.f ? 0.f : x1;
      }
    }
    __syncthreads();
    if (q == 0) {
      float l0 = 0.f, l1 = 0.f;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj)
        if (jj < hidden) {
          l0 += w2[jj] * ha[jj];
          l1 += w2[jj] * hm[jj];
        }
      const float l = l0 + l1;
      s_s[c] = 1.f / (1.f + expf(-l));
    }
    __syncthreads();
  }

  // ---- pooled over the tile + halo (slam_pool_kernel's arithmetic); the tile's own pixels also leave s*u, split, in a_img ----
  const float4 sc = *reinterpret_cast<const float4*>(s_s + cq * 4);
  if (!(dbg & 1))
#pragma unroll
  for (int i = 0; i < NSLOT; ++i) {
    const int qq = i * 16 + (tid >> 4);
    const int ry = (int)(((float)qq + 0.5f) * inv_rw), rx = qq - ry * RW;
    const int y = oh0 - 3 + ry, x = ow0 - 3 + rx;
    const bool in = qq < nreg && y >= 0 && y < h && x >= 0 && x < w;
    const float y0 = uv[i].x * sc.x, y1 = uv[i].y * sc.y, y2 = uv[i].z * sc.z, y3 = uv[i].w * sc.w;
    float sum = (y0 + y1) + (y2 + y3);
    float mx = y0;
    if (pool_takes(y1, mx)) mx = y1;
    if (pool_takes(y2, mx)) mx = y2;
    if (pool_takes(y3, mx)) mx = y3;
    sum = group16_sum(sum);
    mx = group16_max(mx);                              // value-only form of the arg-max merge: the larger one, a NaN on either side stays
    if (qq < nreg) {
      if (cq == 0) pooled_s[qq] = in ? make_float2(sum / (float)TC, mx) : make_float2(0.f, 0.f);
      if (in && ry >= 3 && ry < 3 + PH && rx >= 3 && rx < 3 + PW) {
        // z = s*u of 4 channels: 8-group g = cq / 2, half hh = cq & 1: hi pair at g*32 + hh*8, lo pair 16 bytes further
        const int r = (ry - 3) * PW + (rx - 3);
        const tl_bf16x2 h01 = {(__bf16)y0, (__bf16)y1}, h23 = {(__bf16)y2, (__bf16)y3};
        const unsigned uh01 = __builtin_bit_cast(unsigned, h01), uh23 = __builtin_bit_cast(unsigned, h23);
        const tl_bf16x2 l01 = {(__bf16)(y0 - __uint_as_float(uh01 << 16)), (__bf16)(y1 - __uint_as_float(uh01 & 0xffff0000u))};
        const tl_bf16x2 l23 = {(__bf16)(y2 - __uint_as_float(uh23 << 16)), (__bf16)(y3 - __uint_as_float(uh23 & 0xffff0000u))};
        char* dstp = a_img + r * EV_PITCH + (cq >> 1) * 32 + (cq & 1) * 8;
        const tl_u32x2 hv = {uh01, uh23};
        const tl_u32x2 lv = {__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
        *reinterpret_cast<tl_u32x2*>(dstp) = hv;
        *reinterpret_cast<tl_u32x2*>(dstp + 16) = lv;
      }
    }
  }
  __syncthreads();

  // ---- m = sigmoid(conv7x7(pooled)) for the tile's pixels (slam_conv7_kernel's order: taps outside the IMAGE are skipped) ----
  if (tid < PH * PW && !(dbg & 2)) {
    const int py = (int)(((float)tid + 0.5f) * inv_pw), px = tid - py * PW;
    const int y = oh0 + py, x = ow0 + px;
    float acc = 0.f;
    if (y < h && x < w) {
#pragma unroll
      for (int kh = 0; kh < 7; ++kh) {
        const int yy = y + kh - 3;
        if (yy < 0 || yy >= h) continue;
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
          const int xx = x + kw - 3;
          if (xx < 0 || xx >= w) continue;
          const float2 p = pooled_s[(py + kh) * RW + (px + kw)];
          acc += sw[kh * 7 + kw] * p.x;
          acc += sw[49 + kh * 7 + kw] * p.y;
        }
      }
    }
    m_s[tid] = 1.f / (1.f + expf(-acc));
  }
  __syncthreads();

  // ---- 1x1 conv on the MFMA: wave = 32 pixels x 64 output channels, K = 64 in four chunks, products al*bh, ah*bl, ah*bh ----
  const int l31 = lane & 31, khalf = lane >> 5;
  const int r = wave * 32 + l31;
  const int rpy = (int)(((float)r + 0.5f) * inv_pw), rpx = r - rpy * PW;
  const bool rok = r < PH * PW && oh0 + rpy < h && ow0 + rpx < w;
  const size_t pix = (size_t)b * hw + (size_t)(oh0 + rpy) * w + (ow0 + rpx);
  float4 sk[8];                                      // the skip values of this lane's 32 outputs: in flight under the MFMAs
#pragma unroll
  for (int i = 0; i < 8; ++i)
    sk[i] = rok ? *reinterpret_cast<const float4*>(skip + pix * TC + (i >> 2) * 32 + 8 * (i & 3) + 4 * khalf) : make_float4(0.f, 0.f, 0.f, 0.f);
  f32x16 acc2[2];
#pragma unroll
  for (int uu = 0; uu < 2; ++uu)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[uu][r] = 0.f;
  const char* prow = a_img + (wave * 32 + l31) * EV_PITCH + khalf * 32;
  if (!(dbg & 4))
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const tl_bf16x8 ph = *reinterpret_cast<const tl_bf16x8*>(prow + cc * 64);
    const tl_bf16x8 pl = *reinterpret_cast<const tl_bf16x8*>(prow + cc * 64 + 16);
    tl_bf16x8 wh[2], wl[2];
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      const char* wrow = w_img + (uu * 32 + l31) * EV_PITCH + khalf * 32 + cc * 64;
      wh[uu] = *reinterpret_cast<const tl_bf16x8*>(wrow);
      wl[uu] = *reinterpret_cast<const tl_bf16x8*>(wrow + 16);
    }
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) acc2[uu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[uu], pl, acc2[uu], 0, 0, 0);   // al * bh
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) acc2[uu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[uu], ph, acc2[uu], 0, 0, 0);   // ah * bl
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) acc2[uu] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[uu], ph, acc2[uu], 0, 0, 0);   // ah * bh
  }

  // ---- epilogue: out = m * acc + bias + skip (epi_apply_store's order: row scale, bias, residual) ----
  if (rok && !(dbg & 8)) {
    const float mm = m_s[r];
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = uu * 32 + 8 * q + 4 * khalf;
        float4 v = make_float4(acc2[uu][4 * q], acc2[uu][4 * q + 1], acc2[uu][4 * q + 2], acc2[uu][4 * q + 3]);
        v.x *= mm; v.y *= mm; v.z *= mm; v.w *= mm;
        if (bc != nullptr) {
          const float4 bb = *reinterpret_cast<const float4*>(bc + n);
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        const float4 r4 = sk[uu * 4 + q];
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
        *reinterpret_cast<float4*>(out + pix * TC + n) = v;
      }
  }
}


extern "C" {

size_t srhip_attn_tail_workspace(int n) { return (size_t)n * (SEG > TAIL_BLK ? SEG : TAIL_BLK) * TC * 3 * sizeof(float); }

static int tail_fwd_impl(const float* u, const float* psum, const float* pmax, const int* parg, int nseg, const float* fc1, const float* fc2,
                         const float* w7, float* avg, float* mx, int* argmax_hw, float* s, float* pooled, int* argc, float* m, int n, int h,
                         int w, int hidden, hipStream_t st) {
  const int hw = h * w;
  const long npix = (long)n * hw;
  hipLaunchKernelGGL(clam_mlp_kernel, dim3(n), dim3(256), 0, st, psum, pmax, parg, fc1, fc2, avg, mx, argmax_hw, s, hw, hidden, nseg);
  hipLaunchKernelGGL(slam_pool_kernel, dim3(cdiv(npix, 16)), dim3(256), 0, st, u, s, reinterpret_cast<float2*>(pooled), argc, hw, npix);
  hipLaunchKernelGGL(slam_conv7_kernel, dim3(cdiv(npix, 256)), dim3(256), 0, st, reinterpret_cast<const float2*>(pooled), w7, m, h, w, npix);
  return check_launch("attn_tail_fwd");
}

int srhip_attn_tail_fwd(const float* u, const float* fc1, const float* fc2, const float* w7, float* avg, float* mx,
                        int* argmax_hw, float* s, float* pooled, int* argc, float* m, void* workspace,
                        size_t workspace_bytes, int n, int h, int w, int c, int hidden, void* stream) {
  SRHIP_REQUIRE(u && fc1 && fc2 && w7 && avg && mx && argmax_hw && s && pooled && argc && m, "attn_tail_fwd: null tensor");
  SRHIP_REQUIRE(c == TC && hidden >= 1 && hidden <= 16 && n > 0 && h > 0 && w > 0, "attn_tail_fwd: C must be 64, hidden <= 16");
  SRHIP_REQUIRE(workspace && workspace_bytes >= srhip_attn_tail_workspace(n), "attn_tail_fwd: workspace too small");
  hipStream_t st = as_stream(stream);
  float* psum = static_cast<float*>(workspace);
  float* pmax = psum + (size_t)n * SEG * TC;
  int* parg = reinterpret_cast<int*>(pmax + (size_t)n * SEG * TC);
  hipLaunchKernelGGL(clam_pool_partial_kernel, dim3(n * SEG), dim3(256), 0, st, u, psum, pmax, parg, h * w);
  return tail_fwd_impl(u, psum, pmax, parg, SEG, fc1, fc2, w7, avg, mx, argmax_hw, s, pooled, argc, m, n, h, w, hidden, st);
}

/* ABI 8: the CLAM pooling partials as an object of their own.  `pool` = three sections [sum | max | first arg-max pixel] of
 * pool_sec_bytes each, [image][segment][64] inside a section; srhip_clam_pool_partial fills it with srhip_clam_pool_segments()
 * segments per image, srhip_conv2d_fwd_pool (conv_api.hip) lets the producing conv's epilogue fill it; the *_pooled tails consume it. */
int srhip_clam_pool_segments(void) { return SEG; }
int srhip_clam_pool_max_segments(void) { return POOL_MAXSEG; }

static bool pool_sections(const float* pool, size_t sec_bytes, int n, int nseg, const float** psum, const float** pmax, const int** parg) {
  if (!pool || nseg < 1 || nseg > POOL_MAXSEG || sec_bytes % 16 != 0 || (size_t)n * nseg * TC * sizeof(float) > sec_bytes) return false;
  *psum = pool;
  *pmax = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pool) + sec_bytes);
  *parg = reinterpret_cast<const int*>(reinterpret_cast<const char*>(pool) + 2 * sec_bytes);
  return true;
}

int srhip_clam_pool_partial(const float* u, float* pool, size_t pool_sec_bytes, int n, int h, int w, int c, void* stream) {
  const float *psum, *pmax;
  const int* parg;
  SRHIP_REQUIRE(u && c == TC && n > 0 && h > 0 && w > 0 && pool_sections(pool, pool_sec_bytes, n, SEG, &psum, &pmax, &parg),
                "clam_pool_partial: C must be 64, three 16-byte aligned sections of n * segments * 64 floats");
  hipLaunchKernelGGL(clam_pool_partial_kernel, dim3(n * SEG), dim3(256), 0, as_stream(stream), u, const_cast<float*>(psum),
                     const_cast<float*>(pmax), const_cast<int*>(parg), h * w);
  return check_launch("clam_pool_partial");
}

int srhip_attn_tail_fwd_pooled(const float* u, const float* pool, size_t pool_sec_bytes, int nseg, const float* fc1, const float* fc2,
                               const float* w7, float* avg, float* mx, int* argmax_hw, float* s, float* pooled, int* argc, float* m,
                               int n, int h, int w, int c, int hidden, void* stream) {
  const float *psum, *pmax;
  const int* parg;
  SRHIP_REQUIRE(u && fc1 && fc2 && w7 && avg && mx && argmax_hw && s && pooled && argc && m, "attn_tail_fwd_pooled: null tensor");
  SRHIP_REQUIRE(c == TC && hidden >= 1 && hidden <= 16 && n > 0 && h > 0 && w > 0, "attn_tail_fwd_pooled: C must be 64, hidden <= 16");
  SRHIP_REQUIRE(pool_sections(pool, pool_sec_bytes, n, nseg, &psum, &pmax, &parg), "attn_tail_fwd_pooled: bad pooling partials");
  return tail_fwd_impl(u, psum, pmax, parg, nseg, fc1, fc2, w7, avg, mx, argmax_hw, s, pooled, argc, m, n, h, w, hidden, as_stream(stream));
}

/* Inference form of the tail (round 4, ABI 6): out = conv1x1(SLAM(CLAM(u))) + bc + skip in two launches (pooling partials, then one
 * fused kernel) instead of five + the 1x1 conv; nothing is saved for a backward.  `wc_packed` is the fprop packed weight of the 1x1
 * conv (srhip_pack_weight, mode 0: the split-bf16 section is read), bc may be NULL.  Split-bf16 arithmetic only (the caller takes
 * srhip_attn_tail_fwd + srhip_conv2d_fwd in the other modes); bit-identical to that pair in split-bf16.
 * Replaces: the eval-mode forward of sradsgan.py:254-274 / 303-323.                                                          */
static int tail_eval_impl(const float* u, const float* skip, const float* psum, const float* pmax, int nseg, const float* fc1,
                          const float* fc2, const float* w7, const float* wc_packed, const float* bc, float* out, int n, int h, int w,
                          int hidden, hipStream_t st) {
  // tile: PH x PW <= 128 pixels, pooled region (PH + 6) x (PW + 6) <= EV_MAXREG, fewest dead rows over the image
  int PH = 1, PW = 1;
  double best = -1.0;
  for (int pw = 4; pw <= 64 && pw <= w + 3; ++pw) {
    int ph = 128 / pw;
    if (ph > h) ph = h;
    if (ph < 1 || (ph + 6) * (pw + 6) > EV_MAXREG) continue;
    const double eff = (double)h * w / ((double)cdiv(h, ph) * cdiv(w, pw) * 128.0) - 1e-4 * (double)((ph + 6) * (pw + 6)) / (ph * pw);
    if (eff > best) {
      best = eff;
      PH = ph;
      PW = pw;
    }
  }
  const int tiles_h = cdiv(h, PH), tiles_w = cdiv(w, PW);
  hipLaunchKernelGGL(attn_tail_eval_kernel, dim3(n * tiles_h * tiles_w), dim3(256), 0, st, u, skip, psum, pmax, fc1, fc2, w7,
                     wc_packed + (size_t)TC * TC, bc, out, h, w, hidden, PH, PW, tiles_h, tiles_w, g_tail_dbg, nseg);
  return check_launch("attn_tail_eval");
}

int srhip_attn_tail_eval(const float* u, const float* skip, const float* fc1, const float* fc2, const float* w7,
                         const float* wc_packed, const float* bc, float* out, void* workspace, size_t workspace_bytes, int n, int h,
                         int w, int c, int hidden, void* stream) {
  SRHIP_REQUIRE(u && skip && fc1 && fc2 && w7 && wc_packed && out, "attn_tail_eval: null tensor");
  SRHIP_REQUIRE(c == TC && hidden >= 1 && hidden <= 16 && n > 0 && h > 0 && w > 0, "attn_tail_eval: C must be 64, hidden <= 16");
  SRHIP_REQUIRE(workspace && workspace_bytes >= srhip_attn_tail_workspace(n), "attn_tail_eval: workspace too small");
  SRHIP_REQUIRE(srhip_get_conv_math() == SRHIP_MATH_BF16X3, "attn_tail_eval: split-bf16 arithmetic only");
  SRHIP_REQUIRE(((((uintptr_t)u) | ((uintptr_t)skip) | ((uintptr_t)out) | ((uintptr_t)wc_packed) | ((uintptr_t)bc)) & 15) == 0, "attn_tail_eval: 16-byte aligned tensors");
  hipStream_t st = as_stream(stream);
  float* psum = static_cast<float*>(workspace);
  float* pmax = psum + (size_t)n * SEG * TC;
  int* parg = reinterpret_cast<int*>(pmax + (size_t)n * SEG * TC);
  hipLaunchKernelGGL(clam_pool_partial_kernel, dim3(n * SEG), dim3(256), 0, st, u, psum, pmax, parg, h * w);
  return tail_eval_impl(u, skip, psum, pmax, SEG, fc1, fc2, w7, wc_packed, bc, out, n, h, w, hidden, st);
}

int srhip_attn_tail_eval_pooled(const float* u, const float* skip, const float* pool, size_t pool_sec_bytes, int nseg, const float* fc1,
                                const float* fc2, const float* w7, const float* wc_packed, const float* bc, float* out, int n, int h,
                                int w, int c, int hidden, void* stream) {
  const float *psum, *pmax;
  const int* parg;
  SRHIP_REQUIRE(u && skip && fc1 && fc2 && w7 && wc_packed && out, "attn_tail_eval_pooled: null tensor");
  SRHIP_REQUIRE(c == TC && hidden >= 1 && hidden <= 16 && n > 0 && h > 0 && w > 0, "attn_tail_eval_pooled: C must be 64, hidden <= 16");
  SRHIP_REQUIRE(pool_sections(pool, pool_sec_bytes, n, nseg, &psum, &pmax, &parg), "attn_tail_eval_pooled: bad pooling partials");
  SRHIP_REQUIRE(srhip_get_conv_math() == SRHIP_MATH_BF16X3, "attn_tail_eval_pooled: split-bf16 arithmetic only");
  SRHIP_REQUIRE(((((uintptr_t)u) | ((uintptr_t)skip) | ((uintptr_t)out) | ((uintptr_t)wc_packed) | ((uintptr_t)bc)) & 15) == 0, "attn_tail_eval_pooled: 16-byte aligned tensors");
  return tail_eval_impl(u, skip, psum, pmax, nseg, fc1, fc2, w7, wc_packed, bc, out, n, h, w, hidden, as_stream(stream));
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
  return srhip_attn_tail_bwd_pp(dz, u, s, m, pooled, argc, avg, mx, argmax_hw, w7, fc1, fc2, du, nullptr, dw7, accumulate_dw7, dfc1, dfc2,
                                accumulate_dfc, workspace, workspace_bytes, n, h, w, c, hidden, stream);
}
/* ABI 9: the same with the final du as padded split-bf16 planes in du_pp INSTEAD of the fp32 tensor (du then only holds the main pass's
 * partial result: scratch for the caller); du_pp NULL: srhip_attn_tail_bwd */
int srhip_attn_tail_bwd_pp(const float* dz, const float* u, const float* s, const float* m, const float* pooled, const int* argc,
                           const float* avg, const float* mx, const int* argmax_hw, const float* w7, const float* fc1,
                           const float* fc2, float* du, void* du_pp, float* dw7, int accumulate_dw7, float* dfc1, float* dfc2,
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
  const int pix_blocks = (int)cdiv(npix, 16), red_blocks = (int)cdiv(2 * hidden * TC, 256);
  SRHIP_REQUIRE(!du_pp || (((uintptr_t)du_pp) & 15) == 0, "attn_tail_bwd: du_pp must be 16-byte aligned");
  // round 6: the 7x7 conv's data gradient inside the main pass (tail_bwd_main2_kernel) -- one launch less in the tail's serial chain;
  // srhip_debug_set(7, 32): the round-5 sequence (da, slam_conv7_bwd, main, mlp, fix), kept for A/B runs
  const int ngroups = (hw + 15) / 16, gpb = (int)cdiv(ngroups, TAIL_BLK);
  const int tile_rows = (gpb * 16 + w - 1) / w + 1 + 6;
  const size_t tile_lds = (size_t)tile_rows * (w + 6) * sizeof(float);
  const size_t dyn = tile_lds > w7lds ? tile_lds : w7lds;
  if (!(g_tail_dbg & 32) && dyn <= 60 * 1024) {
    hipLaunchKernelGGL(tail_bwd_main2_kernel, dim3(TAIL_BLK + strips, n), dim3(256), dyn, st, dz, u, s, m, da, w7,
                       reinterpret_cast<const float2*>(pooled), argc, du, dsp, w7part, h, w, TAIL_BLK, strips, gpb, tile_rows);
    hipLaunchKernelGGL(clam_mlp_bwd_kernel, dim3(n), dim3(TC), 0, st, dsp, avg, mx, s, fc1, fc2, davg, dmax, pw1, pw2, hidden, TAIL_BLK);
    hipLaunchKernelGGL(tail_bwd_fix_kernel, dim3(pix_blocks + red_blocks + 25), dim3(256), 0, st, du, davg, dmax, argmax_hw, hw, npix, pix_blocks,
                       pw1, pw2, dfc1, dfc2, n, hidden, accumulate_dfc, static_cast<unsigned*>(du_pp), w, srhip_pp_guard(w), red_blocks,
                       w7part, dw7, w7blk, accumulate_dw7);
    return check_launch("attn_tail_bwd");
  }
  const int nd7 = (int)cdiv(npix, 256);
  if (g_tail_dbg & 16) {                              // srhip_debug_set(7, 16): the two launches of rounds 2-3 (A/B)
    hipLaunchKernelGGL(slam_conv7_dgrad_kernel, dim3(nd7), dim3(256), 0, st, da, w7, dpooled, h, w, npix);
    hipLaunchKernelGGL(slam_conv7_wgrad_kernel, dim3(w7blk), dim3(256), w7lds, st, da, reinterpret_cast<const float2*>(pooled), w7part, h, w, strips);
  } else {
    hipLaunchKernelGGL(slam_conv7_bwd_kernel, dim3(nd7 + w7blk), dim3(256), w7lds, st, da, w7, reinterpret_cast<const float2*>(pooled), dpooled,
                       w7part, h, w, npix, strips, nd7, w7blk);
  }
  hipLaunchKernelGGL(tail_bwd_main_kernel<true>, dim3(TAIL_BLK + 1, n), dim3(256), 0, st, dz, u, s, m, dpooled, argc, du, dsp, hw,
                     w7part, dw7, w7blk, accumulate_dw7);
  hipLaunchKernelGGL(clam_mlp_bwd_kernel, dim3(n), dim3(TC), 0, st, dsp, avg, mx, s, fc1, fc2, davg, dmax, pw1, pw2, hidden, TAIL_BLK);
  hipLaunchKernelGGL(tail_bwd_fix_kernel, dim3(pix_blocks + red_blocks), dim3(256), 0, st, du, davg, dmax, argmax_hw, hw, npix, pix_blocks,
                     pw1, pw2, dfc1, dfc2, n, hidden, accumulate_dfc, static_cast<unsigned*>(du_pp), w, srhip_pp_guard(w));
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
