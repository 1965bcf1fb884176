// Train-mode BatchNorm2d (+ the LeakyReLU that follows it) of the discriminator, NHWC
// (reference sradsgan.py:478-479; nn.BatchNorm2d: biased variance for normalisation, unbiased for
// running_var, momentum 0.1, eps 1e-5).  HBM-bound: forward = 2 reads + 1 write of the tensor,
// backward = 5 reads + 1 write (dy, x, y twice; x-hat is recomputed, never stored).
// Statistics are reduced in two deterministic stages (<= 256 row slabs, then per-column); sums are
// taken about a per-channel shift (the first row) so E[d^2] - E[d]^2 does not cancel.
#include "common.h"

namespace srhip {

__device__ inline float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ---- stage 1 of every per-channel reduction: MODE 0: (sum d, sum d^2), d = x - shift
//                                              MODE 1: (sum dz, sum dz*xhat), dz = dy * lrelu'(y)
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_stage1(const float* __restrict__ a, const float* __restrict__ x,
                                                        const float* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ partial,
                                                        long rows, int c, long rows_per_block, float slope, int act,
                                                        const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr) {
  // beta != nullptr (MODE 1, round 5): the LeakyReLU mask is the sign of the RECOMPUTED pre-activation (x - mean) * invstd * gamma + beta
  // -- the forward's own expression, contraction off in both kernels: the same bits, hence the same sign as y's -- instead of a read
  // of y: 2 reads instead of 3 in this pass, 2 + 1 write instead of 3 + 1 in bn_bwd_apply_kernel (both run at the HBM roofline)
  __shared__ float4 red0[256], red1[256];
  const int tid = threadIdx.x;
  const int q = c / 4, nrl = 256 / q;
  const int cq = tid % q, rl = tid / q;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  float4 p0, p1;
  if (MODE == 0) {
    p0 = *reinterpret_cast<const float4*>(x + cq * 4);                     // shift = first row
  } else {
    p0 = *reinterpret_cast<const float4*>(mean + cq * 4);
    p1 = *reinterpret_cast<const float4*>(invstd + cq * 4);
  }
  float4 pg = make_float4(0.f, 0.f, 0.f, 0.f), pb = pg;
  if (MODE == 1 && beta != nullptr) {
    pg = *reinterpret_cast<const float4*>(gamma + cq * 4);
    pb = *reinterpret_cast<const float4*>(beta + cq * 4);
  }
  if (rl < nrl)
    for (long r = r0 + rl; r < r1; r += nrl) {
      const size_t o = (size_t)r * c + cq * 4;
      if (MODE == 0) {
        const float4 v = *reinterpret_cast<const float4*>(x + o);
        const float4 d = make_float4(v.x - p0.x, v.y - p0.y, v.z - p0.z, v.w - p0.w);
        s0 = f4add(s0, d);
        s1 = f4add(s1, make_float4(d.x * d.x, d.y * d.y, d.z * d.z, d.w * d.w));
      } else {
        float4 g = *reinterpret_cast<const float4*>(a + o);
        const float4 v = *reinterpret_cast<const float4*>(x + o);
        if (act) {
          float4 yy;
          if (beta != nullptr) {
            yy.x = (v.x - p0.x) * p1.x * pg.x + pb.x;
            yy.y = (v.y - p0.y) * p1.y * pg.y + pb.y;
            yy.z = (v.z - p0.z) * p1.z * pg.z + pb.z;
            yy.w = (v.w - p0.w) * p1.w * pg.w + pb.w;
          } else {
            yy = *reinterpret_cast<const float4*>(y + o);
          }
          g.x = yy.x > 0.f ? g.x : g.x * slope;
          g.y = yy.y > 0.f ? g.y : g.y * slope;
          g.z = yy.z > 0.f ? g.z : g.z * slope;
          g.w = yy.w > 0.f ? g.w : g.w * slope;
        }
        s0 = f4add(s0, g);
        s1 = f4add(s1, make_float4(g.x * ((v.x - p0.x) * p1.x), g.y * ((v.y - p0.y) * p1.y),
                                   g.z * ((v.z - p0.z) * p1.z), g.w * ((v.w - p0.w) * p1.w)));
      }
    }
  red0[tid] = s0;
  red1[tid] = s1;
  __syncthreads();
  if (tid < q) {
    for (int k = 1; k < nrl; ++k) {
      s0 = f4add(s0, red0[k * q + tid]);
      s1 = f4add(s1, red1[k * q + tid]);
    }
    float* o = partial + (size_t)blockIdx.x * 2 * c;
    *reinterpret_cast<float4*>(o + tid * 4) = s0;
    *reinterpret_cast<float4*>(o + c + tid * 4) = s1;
  }
}

// Stage-2 kernels: 16 channels x BN_SUBS lanes per block walk the <= 1024 stage-1 partials.  With 16 lanes a thread
// made 64 dependent trips (22 - 54 us per launch for a few KB of work, ~40 launches per step in the discriminator's serial chain).
constexpr int BN_SUBS = 64;

// stage 2 (forward): per channel mean / invstd, running statistics
__global__ void bn_stats_stage2(const float* __restrict__ partial, const float* __restrict__ x, float* __restrict__ mean,
                                float* __restrict__ invstd, float* __restrict__ running_mean,
                                float* __restrict__ running_var, int nblk, int c, long rows, float eps,
                                float momentum) {
  __shared__ float r0[16 * BN_SUBS], r1[16 * BN_SUBS];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), sub = threadIdx.x >> 4;      // 16 columns x BN_SUBS slab lanes
  float a = 0.f, b = 0.f;
  if (col < c)
    for (int k = sub; k < nblk; k += BN_SUBS) {
      a += partial[(size_t)k * 2 * c + col];
      b += partial[(size_t)k * 2 * c + c + col];
    }
  r0[threadIdx.x] = a;
  r1[threadIdx.x] = b;
  __syncthreads();
  if (sub == 0 && col < c) {
    const int t = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < BN_SUBS; ++k) {
      s1 += r0[t + 16 * k];
      s2 += r1[t + 16 * k];
    }
    const float n = (float)rows;
    const float md = s1 / n;
    float var = s2 / n - md * md;
    var = var > 0.f ? var : 0.f;
    const float mu = x[col] + md;
    mean[col] = mu;
    invstd[col] = rsqrtf(var + eps);
    if (running_mean) {
      running_mean[col] = (1.f - momentum) * running_mean[col] + momentum * mu;
      const float unb = rows > 1 ? var * (n / (n - 1.f)) : var;
      running_var[col] = (1.f - momentum) * running_var[col] + momentum * unb;
    }
  }
}

// stage 2 (backward): dbeta = sum dz, dgamma = sum dz*xhat
// acc_gamma / acc_beta (optional): the parameters' gradient buffers; this thread is their only writer for its channel, so
// `+=` here replaces the host's two 3-microsecond add launches per BatchNorm backward (72 per training step)
__global__ void bn_bwd_stage2(const float* __restrict__ partial, float* __restrict__ dgamma, float* __restrict__ dbeta,
                              int nblk, int c, float* __restrict__ acc_gamma, float* __restrict__ acc_beta) {
  __shared__ float r0[16 * BN_SUBS], r1[16 * BN_SUBS];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), sub = threadIdx.x >> 4;
  float a = 0.f, b = 0.f;
  if (col < c)
    for (int k = sub; k < nblk; k += BN_SUBS) {
      a += partial[(size_t)k * 2 * c + col];
      b += partial[(size_t)k * 2 * c + c + col];
    }
  r0[threadIdx.x] = a;
  r1[threadIdx.x] = b;
  __syncthreads();
  if (sub == 0 && col < c) {
    const int t = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < BN_SUBS; ++k) {
      s1 += r0[t + 16 * k];
      s2 += r1[t + 16 * k];
    }
    dbeta[col] = s1;
    dgamma[col] = s2;
    if (acc_gamma) acc_gamma[col] += s2;
    if (acc_beta) acc_beta[col] += s1;
  }
}

// y = act((x - mean) * (invstd*gamma) + beta)
__global__ void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                const float* __restrict__ beta, float* __restrict__ y, long n4, int c, float slope,
                                int act) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    const int ch = (int)((i * 4) % c);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 mu = *reinterpret_cast<const float4*>(mean + ch);
    const float4 is = *reinterpret_cast<const float4*>(invstd + ch);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + ch);
    const float4 be = *reinterpret_cast<const float4*>(beta + ch);
    float4 o;
    o.x = (v.x - mu.x) * is.x * ga.x + be.x;
    o.y = (v.y - mu.y) * is.y * ga.y + be.y;
    o.z = (v.z - mu.z) * is.z * ga.z + be.z;
    o.w = (v.w - mu.w) * is.w * ga.w + be.w;
    if (act) {
      o.x = o.x > 0.f ? o.x : o.x * slope;
      o.y = o.y > 0.f ? o.y : o.y * slope;
      o.z = o.z > 0.f ? o.z : o.z * slope;
      o.w = o.w > 0.f ? o.w : o.w * slope;
    }
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

// eval(): y = act((x - running_mean) / sqrt(running_var + eps) * gamma + beta) -- a per-channel affine
__global__ void bn_eval_kernel(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ var,
                               const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                               long n4, int c, float eps, float slope, int act) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    const int ch = (int)((i * 4) % c);
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float vv[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float is = 1.0f / sqrtf(var[ch + e] + eps);
      o[e] = (vv[e] - mean[ch + e]) * is * gamma[ch + e] + beta[ch + e];
      if (act) o[e] = o[e] > 0.f ? o[e] : o[e] * slope;
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// dx = gamma*invstd * (dz - dbeta/N - xhat * dgamma/N)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                    const float* __restrict__ y, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                    float* __restrict__ dx, long n4, int c, float inv_n, float slope, int act,
                                    const float* __restrict__ beta = nullptr, const float* addend = nullptr) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    const int ch = (int)((i * 4) % c);
    float4 g = reinterpret_cast<const float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float4 mu = *reinterpret_cast<const float4*>(mean + ch);
    const float4 is = *reinterpret_cast<const float4*>(invstd + ch);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + ch);
    if (act) {
      float4 yy;
      if (beta != nullptr) {                           // the forward's pre-activation, recomputed (see bn_reduce_stage1)
        const float4 be = *reinterpret_cast<const float4*>(beta + ch);
        yy.x = (v.x - mu.x) * is.x * ga.x + be.x;
        yy.y = (v.y - mu.y) * is.y * ga.y + be.y;
        yy.z = (v.z - mu.z) * is.z * ga.z + be.z;
        yy.w = (v.w - mu.w) * is.w * ga.w + be.w;
      } else {
        yy = reinterpret_cast<const float4*>(y)[i];
      }
      g.x = yy.x > 0.f ? g.x : g.x * slope;
      g.y = yy.y > 0.f ? g.y : g.y * slope;
      g.z = yy.z > 0.f ? g.z : g.z * slope;
      g.w = yy.w > 0.f ? g.w : g.w * slope;
    }
    const float4 dg = *reinterpret_cast<const float4*>(dgamma + ch);
    const float4 db = *reinterpret_cast<const float4*>(dbeta + ch);
    float4 o;
    o.x = ga.x * is.x * (g.x - db.x * inv_n - (v.x - mu.x) * is.x * (dg.x * inv_n));
    o.y = ga.y * is.y * (g.y - db.y * inv_n - (v.y - mu.y) * is.y * (dg.y * inv_n));
    o.z = ga.z * is.z * (g.z - db.z * inv_n - (v.z - mu.z) * is.z * (dg.z * inv_n));
    o.w = ga.w * is.w * (g.w - db.w * inv_n - (v.w - mu.w) * is.w * (dg.w * inv_n));
    if (addend != nullptr) {                           // (srhip_bn_train_bwd_acc_xa: x's other gradient, may be dx itself)
      const float4 e = reinterpret_cast<const float4*>(addend)[i];
      o.x += e.x; o.y += e.y; o.z += e.z; o.w += e.w;
    }
    reinterpret_cast<float4*>(dx)[i] = o;
  }
}

// ---- second-order pass (gradient penalty, sradsgan.py:621,639): backward of
//      dx = a (dz - E[dz] - xhat E[dz xhat]),  a = gamma*invstd,  dz = dy * lrelu'(y)
// for a cotangent u on dx.  With ubar = E[u], w = E[u xhat], p = E[dz], q = E[dz xhat], T = E[u dz] - ubar p - w q:
//      d/d(dy)    = a (u - ubar - xhat w) * lrelu'(y)
//      d/dx       = -gamma invstd^2 [ q (u - ubar) + w (dz - p) + xhat (T - 2 w q) ]
//      d/dgamma   = invstd * N * T
// stage 1 reduces the five per-channel sums, stage 2 turns them into coefficients, apply writes both tensors.
__global__ __launch_bounds__(256) void bn_bwd2_stage1(const float* __restrict__ u, const float* __restrict__ dy,
                                                      const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      float* __restrict__ partial, long rows, int c,
                                                      long rows_per_block, float slope, int act,
                                                      const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr) {
  __shared__ float4 red[5][256];
  const int tid = threadIdx.x;
  const int q = c / 4, nrl = 256 / q;
  const int cq = tid % q, rl = tid / q;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  const float4 mu = *reinterpret_cast<const float4*>(mean + cq * 4);
  const float4 is = *reinterpret_cast<const float4*>(invstd + cq * 4);
  float4 pg = make_float4(0.f, 0.f, 0.f, 0.f), pb = pg;
  if (beta != nullptr) {                               // mask from the recomputed pre-activation (see bn_reduce_stage1)
    pg = *reinterpret_cast<const float4*>(gamma + cq * 4);
    pb = *reinterpret_cast<const float4*>(beta + cq * 4);
  }
  float4 s[5];
  for (int k = 0; k < 5; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < nrl)
    for (long r = r0 + rl; r < r1; r += nrl) {
      const size_t o = (size_t)r * c + cq * 4;
      const float4 uu = *reinterpret_cast<const float4*>(u + o);
      float4 g = *reinterpret_cast<const float4*>(dy + o);
      const float4 v = *reinterpret_cast<const float4*>(x + o);
      if (act) {
        float4 yy;
        if (beta != nullptr) {
          yy.x = (v.x - mu.x) * is.x * pg.x + pb.x;
          yy.y = (v.y - mu.y) * is.y * pg.y + pb.y;
          yy.z = (v.z - mu.z) * is.z * pg.z + pb.z;
          yy.w = (v.w - mu.w) * is.w * pg.w + pb.w;
        } else {
          yy = *reinterpret_cast<const float4*>(y + o);
        }
        g.x = yy.x > 0.f ? g.x : g.x * slope;
        g.y = yy.y > 0.f ? g.y : g.y * slope;
        g.z = yy.z > 0.f ? g.z : g.z * slope;
        g.w = yy.w > 0.f ? g.w : g.w * slope;
      }
      const float4 xh = make_float4((v.x - mu.x) * is.x, (v.y - mu.y) * is.y, (v.z - mu.z) * is.z, (v.w - mu.w) * is.w);
      s[0] = f4add(s[0], uu);
      s[1] = f4add(s[1], make_float4(uu.x * xh.x, uu.y * xh.y, uu.z * xh.z, uu.w * xh.w));
      s[2] = f4add(s[2], g);
      s[3] = f4add(s[3], make_float4(g.x * xh.x, g.y * xh.y, g.z * xh.z, g.w * xh.w));
      s[4] = f4add(s[4], make_float4(uu.x * g.x, uu.y * g.y, uu.z * g.z, uu.w * g.w));
    }
  for (int k = 0; k < 5; ++k) red[k][tid] = s[k];
  __syncthreads();
  if (tid < q) {
    for (int k = 0; k < 5; ++k) {
      float4 a = s[k];
      for (int j = 1; j < nrl; ++j) a = f4add(a, red[k][j * q + tid]);
      *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * 5 + k) * c + tid * 4) = a;
    }
  }
}

// coef[0..4][c] = ubar, w, p, q, T ; dgamma2[c] = invstd * N * T
__global__ void bn_bwd2_stage2(const float* __restrict__ partial, const float* __restrict__ invstd,
                               float* __restrict__ coef, float* __restrict__ dgamma2, int nblk, int c, long rows,
                               float* __restrict__ acc_gamma) {
  __shared__ float r[5][16 * BN_SUBS];
  const int col = blockIdx.x * 16 + (threadIdx.x & 15), sub = threadIdx.x >> 4;
  float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < c)
    for (int k = sub; k < nblk; k += BN_SUBS)
      for (int j = 0; j < 5; ++j) a[j] += partial[((size_t)k * 5 + j) * c + col];
  for (int j = 0; j < 5; ++j) r[j][threadIdx.x] = a[j];
  __syncthreads();
  if (sub == 0 && col < c) {
    float t[5];
    for (int j = 0; j < 5; ++j) {
      float v = 0.f;
      for (int k = 0; k < BN_SUBS; ++k) v += r[j][threadIdx.x + 16 * k];
      t[j] = v;
    }
    const float n = (float)rows;
    const float ubar = t[0] / n, w = t[1] / n, p = t[2] / n, q = t[3] / n;
    const float T = t[4] / n - ubar * p - w * q;
    coef[0 * c + col] = ubar;
    coef[1 * c + col] = w;
    coef[2 * c + col] = p;
    coef[3 * c + col] = q;
    coef[4 * c + col] = T;
    const float dg2 = invstd[col] * n * T;
    dgamma2[col] = dg2;
    if (acc_gamma) acc_gamma[col] += dg2;
  }
}

__global__ void bn_bwd2_apply_kernel(const float* __restrict__ u, const float* __restrict__ dy,
                                     const float* __restrict__ x, const float* __restrict__ y,
                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                     const float* __restrict__ gamma, const float* __restrict__ coef,
                                     float* __restrict__ g_dy, float* __restrict__ g_x, long n4, int c, float slope,
                                     int act, const float* __restrict__ beta = nullptr) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n4; i += stride) {
    const int ch = (int)((i * 4) % c);
    const float4 uu = reinterpret_cast<const float4*>(u)[i];
    const float4 g0 = reinterpret_cast<const float4*>(dy)[i];
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 mk = make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 mu = *reinterpret_cast<const float4*>(mean + ch);
    const float4 is = *reinterpret_cast<const float4*>(invstd + ch);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + ch);
    if (act) {
      float4 yy;
      if (beta != nullptr) {
        const float4 be = *reinterpret_cast<const float4*>(beta + ch);
        yy.x = (v.x - mu.x) * is.x * ga.x + be.x;
        yy.y = (v.y - mu.y) * is.y * ga.y + be.y;
        yy.z = (v.z - mu.z) * is.z * ga.z + be.z;
        yy.w = (v.w - mu.w) * is.w * ga.w + be.w;
      } else {
        yy = reinterpret_cast<const float4*>(y)[i];
      }
      mk = make_float4(yy.x > 0.f ? 1.f : slope, yy.y > 0.f ? 1.f : slope, yy.z > 0.f ? 1.f : slope,
                       yy.w > 0.f ? 1.f : slope);
    }
    const float4 ub = *reinterpret_cast<const float4*>(coef + ch);
    const float4 ww = *reinterpret_cast<const float4*>(coef + c + ch);
    const float4 pp = *reinterpret_cast<const float4*>(coef + 2 * c + ch);
    const float4 qq = *reinterpret_cast<const float4*>(coef + 3 * c + ch);
    const float4 TT = *reinterpret_cast<const float4*>(coef + 4 * c + ch);
    float4 od, ox;
#define SRHIP_BN2(F)                                                                                     \
  {                                                                                                      \
    const float xh = (v.F - mu.F) * is.F;                                                                \
    const float dz = g0.F * mk.F;                                                                        \
    od.F = ga.F * is.F * (uu.F - ub.F - xh * ww.F) * mk.F;                                               \
    ox.F = -ga.F * is.F * is.F *                                                                         \
           (qq.F * (uu.F - ub.F) + ww.F * (dz - pp.F) + xh * (TT.F - 2.f * ww.F * qq.F));                \
  }
    SRHIP_BN2(x) SRHIP_BN2(y) SRHIP_BN2(z) SRHIP_BN2(w)
#undef SRHIP_BN2
    reinterpret_cast<float4*>(g_dy)[i] = od;
    reinterpret_cast<float4*>(g_x)[i] = ox;
  }
}

static long bn_nblk(long rows) {
  long nblk = (rows + 63) / 64;
  return nblk > 1024 ? 1024 : (nblk < 1 ? 1 : nblk);   // 4 blocks per CU: 256 could not saturate HBM
}

}  // namespace srhip

using namespace srhip;

extern "C" {

size_t srhip_bn_workspace(long rows, int c) { return (size_t)bn_nblk(rows) * 2 * c * sizeof(float); }

int srhip_bn_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float* y, float* save_mean, float* save_invstd, void* workspace,
                       size_t workspace_bytes, long rows, int c, float eps, float momentum, float slope, int apply_act,
                       void* stream) {
  SRHIP_REQUIRE(x && gamma && beta && y && save_mean && save_invstd, "bn_train_fwd: null tensor");
  SRHIP_REQUIRE(rows > 0 && c >= 4 && c % 4 == 0 && c <= 1024, "bn_train_fwd: C must be a multiple of 4, <= 1024");
  SRHIP_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_fwd: running stats come in pairs");
  SRHIP_REQUIRE(workspace && workspace_bytes >= srhip_bn_workspace(rows, c), "bn_train_fwd: workspace too small");
  hipStream_t st = as_stream(stream);
  const long nblk = bn_nblk(rows), rpb = (rows + nblk - 1) / nblk;
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(bn_reduce_stage1<0>, dim3((int)nblk), dim3(256), 0, st, nullptr, x, nullptr, nullptr, nullptr, part,
                     rows, c, rpb, 0.f, 0);
  hipLaunchKernelGGL(bn_stats_stage2, dim3(cdiv(c, 16)), dim3(16 * BN_SUBS), 0, st, part, x, save_mean, save_invstd, running_mean,
                     running_var, (int)nblk, c, rows, eps, momentum);
  const long n4 = rows * c / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, st, x, save_mean, save_invstd, gamma, beta, y, n4, c,
                     slope, apply_act);
  return check_launch("bn_train_fwd");
}

int srhip_bn_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, float* y, long rows, int c, float eps, float slope, int apply_act,
                      void* stream) {
  SRHIP_REQUIRE(x && gamma && beta && running_mean && running_var && y, "bn_eval_fwd: null tensor");
  SRHIP_REQUIRE(rows > 0 && c >= 4 && c % 4 == 0, "bn_eval_fwd: C must be a multiple of 4");
  const long n4 = rows * c / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(bn_eval_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, running_mean, running_var, gamma, beta, y,
                     n4, c, eps, slope, apply_act);
  return check_launch("bn_eval_fwd");
}

int srhip_bn_train_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean,
                       const float* save_invstd, float* dx, float* dgamma, float* dbeta, void* workspace,
                       size_t workspace_bytes, long rows, int c, float slope, int apply_act, void* stream) {
  return srhip_bn_train_bwd_acc(dy, x, y, gamma, save_mean, save_invstd, dx, dgamma, dbeta, nullptr, nullptr, workspace,
                                workspace_bytes, rows, c, slope, apply_act, stream);
}

static int bn_train_bwd_impl(const float* dy, const float* x, const float* y, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* acc_gamma,
                             float* acc_beta, void* workspace, size_t workspace_bytes, long rows, int c, float slope,
                             int apply_act, void* stream, const float* addend = nullptr);
int srhip_bn_train_bwd_acc(const float* dy, const float* x, const float* y, const float* gamma, const float* save_mean,
                           const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* acc_gamma,
                           float* acc_beta, void* workspace, size_t workspace_bytes, long rows, int c, float slope,
                           int apply_act, void* stream) {
  return bn_train_bwd_impl(dy, x, y, gamma, nullptr, save_mean, save_invstd, dx, dgamma, dbeta, acc_gamma, acc_beta, workspace,
                           workspace_bytes, rows, c, slope, apply_act, stream);
}
/* ABI 9: the same backward WITHOUT the activation output y: the LeakyReLU mask is the sign of the pre-activation recomputed from x
 * (mean, invstd, gamma, beta: the forward's own expression, bit for bit) -- one tensor read less in each of the two passes */
int srhip_bn_train_bwd_acc_x(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* acc_gamma,
                             float* acc_beta, void* workspace, size_t workspace_bytes, long rows, int c, float slope,
                             int apply_act, void* stream) {
  SRHIP_REQUIRE(beta != nullptr, "bn_train_bwd_acc_x: beta is what replaces y");
  return bn_train_bwd_impl(dy, x, nullptr, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, acc_gamma, acc_beta, workspace,
                           workspace_bytes, rows, c, slope, apply_act, stream);
}
/* The same with dx = (BatchNorm backward) + addend in the apply pass: where x has a second consumer whose gradient is already known (the
 * gradient penalty's double backward reaches a BatchNorm input through the first-order backward's node AND through the forward node,
 * sradsgan.py:621-639), the sum autograd would form with one more pass over three tensors.  addend may be dx (in place). */
int srhip_bn_train_bwd_acc_xa(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                              const float* save_invstd, const float* addend, float* dx, float* dgamma, float* dbeta, float* acc_gamma,
                              float* acc_beta, void* workspace, size_t workspace_bytes, long rows, int c, float slope,
                              int apply_act, void* stream) {
  SRHIP_REQUIRE(beta != nullptr && addend != nullptr, "bn_train_bwd_acc_xa: beta and the addend");
  return bn_train_bwd_impl(dy, x, nullptr, gamma, beta, save_mean, save_invstd, dx, dgamma, dbeta, acc_gamma, acc_beta, workspace,
                           workspace_bytes, rows, c, slope, apply_act, stream, addend);
}
static int bn_train_bwd_impl(const float* dy, const float* x, const float* y, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, float* dx, float* dgamma, float* dbeta, float* acc_gamma,
                             float* acc_beta, void* workspace, size_t workspace_bytes, long rows, int c, float slope,
                             int apply_act, void* stream, const float* addend) {
  SRHIP_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && (y || beta || !apply_act),
                "bn_train_bwd: null tensor");
  SRHIP_REQUIRE(rows > 0 && c >= 4 && c % 4 == 0 && c <= 1024, "bn_train_bwd: C must be a multiple of 4, <= 1024");
  SRHIP_REQUIRE(workspace && workspace_bytes >= srhip_bn_workspace(rows, c), "bn_train_bwd: workspace too small");
  hipStream_t st = as_stream(stream);
  const long nblk = bn_nblk(rows), rpb = (rows + nblk - 1) / nblk;
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(bn_reduce_stage1<1>, dim3((int)nblk), dim3(256), 0, st, dy, x, y, save_mean, save_invstd, part, rows,
                     c, rpb, slope, apply_act, gamma, beta);
  hipLaunchKernelGGL(bn_bwd_stage2, dim3(cdiv(c, 16)), dim3(16 * BN_SUBS), 0, st, part, dgamma, dbeta, (int)nblk, c, acc_gamma, acc_beta);
  const long n4 = rows * c / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, st, dy, x, y, save_mean, save_invstd, gamma, dgamma,
                     dbeta, dx, n4, c, 1.f / (float)rows, slope, apply_act, beta, addend);
  return check_launch("bn_train_bwd");
}

size_t srhip_bn_bwd2_workspace(long rows, int c) { return ((size_t)bn_nblk(rows) * 5 * c + 5 * (size_t)c) * sizeof(float); }

int srhip_bn_train_bwd_bwd(const float* ddx, const float* dy, const float* x, const float* y, const float* gamma,
                           const float* save_mean, const float* save_invstd, float* g_dy, float* g_x, float* g_gamma,
                           void* workspace, size_t workspace_bytes, long rows, int c, float slope, int apply_act,
                           void* stream) {
  return srhip_bn_train_bwd_bwd_acc(ddx, dy, x, y, gamma, save_mean, save_invstd, g_dy, g_x, g_gamma, nullptr, workspace,
                                    workspace_bytes, rows, c, slope, apply_act, stream);
}

static int bn_bwd2_impl(const float* ddx, const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, float* g_dy, float* g_x, float* g_gamma,
                        float* acc_gamma, void* workspace, size_t workspace_bytes, long rows, int c, float slope,
                        int apply_act, void* stream);
int srhip_bn_train_bwd_bwd_acc(const float* ddx, const float* dy, const float* x, const float* y, const float* gamma,
                               const float* save_mean, const float* save_invstd, float* g_dy, float* g_x, float* g_gamma,
                               float* acc_gamma, void* workspace, size_t workspace_bytes, long rows, int c, float slope,
                               int apply_act, void* stream) {
  return bn_bwd2_impl(ddx, dy, x, y, gamma, nullptr, save_mean, save_invstd, g_dy, g_x, g_gamma, acc_gamma, workspace, workspace_bytes,
                      rows, c, slope, apply_act, stream);
}
/* ABI 9: the second-order pass without y (mask from the recomputed pre-activation, like srhip_bn_train_bwd_acc_x) */
int srhip_bn_train_bwd_bwd_acc_x(const float* ddx, const float* dy, const float* x, const float* gamma, const float* beta,
                                 const float* save_mean, const float* save_invstd, float* g_dy, float* g_x, float* g_gamma,
                                 float* acc_gamma, void* workspace, size_t workspace_bytes, long rows, int c, float slope,
                                 int apply_act, void* stream) {
  SRHIP_REQUIRE(beta != nullptr, "bn_train_bwd_bwd_acc_x: beta is what replaces y");
  return bn_bwd2_impl(ddx, dy, x, nullptr, gamma, beta, save_mean, save_invstd, g_dy, g_x, g_gamma, acc_gamma, workspace, workspace_bytes,
                      rows, c, slope, apply_act, stream);
}
static int bn_bwd2_impl(const float* ddx, const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                        const float* save_mean, const float* save_invstd, float* g_dy, float* g_x, float* g_gamma,
                        float* acc_gamma, void* workspace, size_t workspace_bytes, long rows, int c, float slope,
                        int apply_act, void* stream) {
  SRHIP_REQUIRE(ddx && dy && x && gamma && save_mean && save_invstd && g_dy && g_x && g_gamma && (y || beta || !apply_act),
                "bn_train_bwd_bwd: null tensor");
  SRHIP_REQUIRE(rows > 0 && c >= 4 && c % 4 == 0 && c <= 1024, "bn_train_bwd_bwd: C must be a multiple of 4, <= 1024");
  SRHIP_REQUIRE(workspace && workspace_bytes >= srhip_bn_bwd2_workspace(rows, c), "bn_train_bwd_bwd: workspace too small");
  hipStream_t st = as_stream(stream);
  const long nblk = bn_nblk(rows), rpb = (rows + nblk - 1) / nblk;
  float* part = static_cast<float*>(workspace);
  float* coef = part + (size_t)nblk * 5 * c;
  hipLaunchKernelGGL(bn_bwd2_stage1, dim3((int)nblk), dim3(256), 0, st, ddx, dy, x, y, save_mean, save_invstd, part, rows,
                     c, rpb, slope, apply_act, gamma, beta);
  hipLaunchKernelGGL(bn_bwd2_stage2, dim3(cdiv(c, 16)), dim3(16 * BN_SUBS), 0, st, part, save_invstd, coef, g_gamma, (int)nblk, c,
                     rows, acc_gamma);
  const long n4 = rows * c / 4;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(bn_bwd2_apply_kernel, dim3(blocks), dim3(256), 0, st, ddx, dy, x, y, save_mean, save_invstd, gamma,
                     coef, g_dy, g_x, n4, c, slope, apply_act, beta);
  return check_launch("bn_train_bwd_bwd");
}

}  // extern "C"
