// Global attention block of GAB_UP (SRADSGAN/model/sradsgan.py:153-213), C = 64, NHWC fp32:
//   CGAM  (:178-213, light=False)  E = X^T X [64x64], A = softmax(rowmax(E) - E), y = gamma * (X A^T) + x
//   SGAM  (:153-176)               q,k [N][8], v [N][64]; att = softmax_j(q_i . k_j); y = gamma * (att v) + x
// CGAM and SGAM's q.k energies / soft-max are exact fp32 on the matrix pipe (v_mfma_f32_32x32x2_f32 = an fmaf chain): the
// energies are sums that feed an exp().  SGAM's four big products per tile pair (P.V, dP, dV, dQ / dK) follow the conv
// arithmetic mode: split-bf16 (a*b ~= ah*bh + ah*bl + al*bh, fp32 accumulate) outside SRHIP_MATH_FP32 -- second half of this file.
//
// SGAM is flash-style: the N x N energy / attention matrices never exist in HBM (N = 2916 at x4, 11664 at x2: the
// reference materialises 34 MB resp. 544 MB per image, twice).  A wave owns 32 queries and walks the keys in tiles of
// 32 with an online softmax; the backward recomputes the probabilities from the saved log-sum-exp in two passes
// (one owning query tiles -> dq, one owning key tiles -> dk, dv), so there are no atomics and results are deterministic.
//
// Register-level trick used throughout: the 32x32x2 MFMA contracts over k in ANY order as long as A and B agree, and
// its C/D layout puts column (lane & 31) x rows {(r&3) + 8(r>>2) + 4(lane>>5)} in a lane.  Computing the TRANSPOSED
// score tile S^T[key][query] therefore leaves a lane with 16 keys of ONE query: the softmax statistics are per lane
// (one cross-half shuffle per tile), and register r of the probabilities is directly the B operand of MFMA step r of
// the P.V product -- no transposes, no LDS round trip for P.
#include <math.h>

#include "common.h"

namespace srhip {

__device__ inline f32x16 ga_mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
// row of C/D register r in lane half h
__device__ __forceinline__ int ga_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

constexpr int GA_C = 64;        // channels
constexpr int GA_PIX = 128;     // pixels per block (4 waves x 32)
constexpr int GA_MS = 65;       // LDS stride of a 64x64 matrix read column-wise by consecutive lanes
constexpr int GA_VS = 68;       // LDS stride of a [32][64] tile whose rows are read as ds_read_b128 (272 B: all 16 slots)
constexpr int GA_DK = 8;        // q / k channels

// ================================================================================================ //
// CGAM
// ================================================================================================ //
// part[img][split][i][j] = sum over the block's pixels p of a[p][i] * b[p][j]   (a = b = x: energy; a = dy, b = x: dA / gamma)
__global__ __launch_bounds__(256) void ga_gram_partial_kernel(const float* __restrict__ a, const float* __restrict__ bm,
                                                              float* __restrict__ part, int hw, int nsplit) {
  const int split = blockIdx.x, img = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int qi = wave >> 1, qj = wave & 1, h = lane >> 5, l31 = lane & 31;
  const int p0 = split * GA_PIX, p1 = min(p0 + GA_PIX, hw);
  const float* ab = a + (size_t)img * hw * GA_C + qi * 32 + l31;
  const float* bb = bm + (size_t)img * hw * GA_C + qj * 32 + l31;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int p = p0; p < p1; p += 16) {
    float av[8], bv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int pp = p + 2 * s + h;
      const bool ok = pp < p1;
      av[s] = ok ? ab[(size_t)pp * GA_C] : 0.f;
      bv[s] = ok ? bb[(size_t)pp * GA_C] : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = ga_mfma(av[s], bv[s], acc);
  }
  float* o = part + ((size_t)img * nsplit + split) * (GA_C * GA_C);
#pragma unroll
  for (int r = 0; r < 16; ++r) o[(qi * 32 + ga_row(r, h)) * GA_C + qj * 32 + l31] = acc[r];
}

__device__ inline float quad_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 1, 64));
  return fmaxf(v, __shfl_xor(v, 2, 64));
}
__device__ inline float quad_sum(float v) {
  v += __shfl_xor(v, 1, 64);
  return v + __shfl_xor(v, 2, 64);
}

// att[img] = softmax(rowmax(E) - E) with E = sum of the partial tiles, in the reference's operation order (:207-209)
__global__ __launch_bounds__(256) void ga_cgam_softmax_kernel(const float* __restrict__ part, float* __restrict__ att, int nsplit) {
  const int img = blockIdx.x, row = threadIdx.x >> 2, sub = threadIdx.x & 3;
  float e[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) e[c] = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float4* src = reinterpret_cast<const float4*>(part + ((size_t)img * nsplit + s) * (GA_C * GA_C) + row * GA_C + sub * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = src[q];
      e[4 * q] += v.x; e[4 * q + 1] += v.y; e[4 * q + 2] += v.z; e[4 * q + 3] += v.w;
    }
  }
  float mx = e[0];
#pragma unroll
  for (int c = 1; c < 16; ++c) mx = fmaxf(mx, e[c]);
  mx = quad_max(mx);
  float m2 = -INFINITY;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    e[c] = mx - e[c];
    m2 = fmaxf(m2, e[c]);
  }
  m2 = quad_max(m2);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    e[c] = expf(e[c] - m2);
    sum += e[c];
  }
  sum = quad_sum(sum);
  float* o = att + (size_t)img * (GA_C * GA_C) + row * GA_C + sub * 16;
#pragma unroll
  for (int c = 0; c < 16; ++c) o[c] = e[c] / sum;
}

// acc[u][r] += sum over k = 32h .. 32h+31 of arow[k - 32h] * mat[k][32u + l31]; arow = this lane's pixel row (+32h) or nullptr
__device__ __forceinline__ void ga_px_mm(f32x16 (&acc)[2], const float* __restrict__ arow, const float* mat, int h, int l31) {
  float4 a4[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) a4[q] = arow ? *reinterpret_cast<const float4*>(arow + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float av[4] = {a4[q].x, a4[q].y, a4[q].z, a4[q].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* mrow = mat + (32 * h + 4 * q + e) * GA_MS + l31;
      acc[0] = ga_mfma(av[e], mrow[0], acc[0]);
      acc[1] = ga_mfma(av[e], mrow[32], acc[1]);
    }
  }
}

// y[p][c] = gamma * sum_c' att[c][c'] x[p][c'] + x[p][c]
__global__ __launch_bounds__(256) void ga_cgam_apply_kernel(const float* __restrict__ x, const float* __restrict__ att,
                                                            const float* __restrict__ gamma, float* __restrict__ y, int hw) {
  __shared__ float mat[GA_C * GA_MS];
  const int img = blockIdx.y, tid = threadIdx.x;
  for (int e = tid; e < GA_C * GA_C; e += 256) mat[(e & 63) * GA_MS + (e >> 6)] = att[(size_t)img * (GA_C * GA_C) + e];   // B[k = c'][j = c]
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int p0 = blockIdx.x * GA_PIX + wave * 32;
  if (p0 >= hw) return;
  const int pix = p0 + l31;
  const size_t base = (size_t)img * hw;
  f32x16 acc[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  ga_px_mm(acc, pix < hw ? x + (base + pix) * GA_C + 32 * h : nullptr, mat, h, l31);
  const float g = gamma[0];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = p0 + ga_row(r, h);
      if (p < hw) {
        const size_t idx = (base + p) * GA_C + 32 * u + l31;
        y[idx] = g * acc[u][r] + x[idx];
      }
    }
}

// per image: G = sum of partials (G[c][c'] = sum_p dy[p][c] x[p][c']); dgpart[img] = <att, G>;
// dA = gamma G; dE = -att * (dA - rowsum(dA * att)); msym = dE + dE^T
__global__ __launch_bounds__(256) void ga_cgam_bwd_small_kernel(const float* __restrict__ part, const float* __restrict__ att,
                                                                const float* __restrict__ gamma, float* __restrict__ msym,
                                                                float* __restrict__ dgpart, int nsplit) {
  __shared__ float de[GA_C * GA_MS];
  __shared__ float red[4];
  const int img = blockIdx.x, tid = threadIdx.x, row = tid >> 2, sub = tid & 3;
  float gsum[16], a[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) gsum[c] = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float4* src = reinterpret_cast<const float4*>(part + ((size_t)img * nsplit + s) * (GA_C * GA_C) + row * GA_C + sub * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = src[q];
      gsum[4 * q] += v.x; gsum[4 * q + 1] += v.y; gsum[4 * q + 2] += v.z; gsum[4 * q + 3] += v.w;
    }
  }
  const float* ar = att + (size_t)img * (GA_C * GA_C) + row * GA_C + sub * 16;
  float dot = 0.f;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    a[c] = ar[c];
    dot += a[c] * gsum[c];
  }
  const float g = gamma[0];
  const float t = g * quad_sum(dot);                    // rowsum(dA * att)
#pragma unroll
  for (int c = 0; c < 16; ++c) de[row * GA_MS + sub * 16 + c] = -a[c] * (g * gsum[c] - t);
  const float wsum = wave_sum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = wsum;
  __syncthreads();
  if (tid == 0) dgpart[img] = (red[0] + red[1]) + (red[2] + red[3]);
  float* o = msym + (size_t)img * (GA_C * GA_C) + row * GA_C + sub * 16;
#pragma unroll
  for (int c = 0; c < 16; ++c) o[c] = de[row * GA_MS + sub * 16 + c] + de[(sub * 16 + c) * GA_MS + row];
}

// out[0] (+)= sum of v[0..n)   (one wave; fixed order => deterministic)
__global__ __launch_bounds__(64) void ga_sum_small_kernel(const float* __restrict__ v, int n, float* __restrict__ out, int accumulate) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += v[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s : s;
}

// dx[p][c'] = dy[p][c'] + sum_c dy[p][c] gamma att[c][c'] + sum_c x[p][c] msym[c][c']
__global__ __launch_bounds__(256) void ga_cgam_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ att, const float* __restrict__ msym,
                                                                const float* __restrict__ gamma, float* __restrict__ dx, int hw) {
  __shared__ float ma[GA_C * GA_MS];
  __shared__ float mm[GA_C * GA_MS];
  const int img = blockIdx.y, tid = threadIdx.x;
  const float g = gamma[0];
  for (int e = tid; e < GA_C * GA_C; e += 256) {
    ma[(e >> 6) * GA_MS + (e & 63)] = g * att[(size_t)img * (GA_C * GA_C) + e];
    mm[(e >> 6) * GA_MS + (e & 63)] = msym[(size_t)img * (GA_C * GA_C) + e];
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int p0 = blockIdx.x * GA_PIX + wave * 32;
  if (p0 >= hw) return;
  const int pix = p0 + l31;
  const size_t base = (size_t)img * hw;
  f32x16 acc[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
  ga_px_mm(acc, pix < hw ? dy + (base + pix) * GA_C + 32 * h : nullptr, ma, h, l31);
  ga_px_mm(acc, pix < hw ? x + (base + pix) * GA_C + 32 * h : nullptr, mm, h, l31);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = p0 + ga_row(r, h);
      if (p < hw) {
        const size_t idx = (base + p) * GA_C + 32 * u + l31;
        dx[idx] = acc[u][r] + dy[idx];
      }
    }
}

// ================================================================================================ //
// SGAM, flash-style
// ================================================================================================ //
// Staging of one 32-row tile of a [N][64] tensor (optionally scaled) and a [N][8] tensor into LDS through registers.
struct GaStage {
  float4 w[2];      // the thread's two float4 of the [32][64] tile
  float4 n;         // threads 0..63: one float4 of the [32][8] tile
};

__device__ __forceinline__ void ga_stage_load(GaStage& s, const float* __restrict__ wide, const float* __restrict__ narrow,
                                              size_t base, int r0, int hw, int tid, float scale) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i, row = r0 + (e >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < hw) {
      v = *reinterpret_cast<const float4*>(wide + (base + row) * GA_C + (e & 15) * 4);
      v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
    }
    s.w[i] = v;
  }
  if (tid < 64) {
    const int row = r0 + (tid >> 1);
    s.n = row < hw ? *reinterpret_cast<const float4*>(narrow + (base + row) * GA_DK + (tid & 1) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void ga_stage_store(const GaStage& s, float* wt, float* nt, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i;
    *reinterpret_cast<float4*>(wt + (e >> 4) * GA_VS + (e & 15) * 4) = s.w[i];
  }
  if (tid < 64) *reinterpret_cast<float4*>(nt + (tid >> 1) * GA_DK + (tid & 1) * 4) = s.n;
}

// S^T[key][query] for this wave's 32 queries against the staged 32 keys: 4 MFMAs (d = 4h + s on both operands)
__device__ __forceinline__ f32x16 ga_scores_t(const float* ktile, const float4& qf, int h, int l31) {
  const float4 kf = *reinterpret_cast<const float4*>(ktile + l31 * GA_DK + 4 * h);
  f32x16 st;
#pragma unroll
  for (int r = 0; r < 16; ++r) st[r] = 0.f;
  st = ga_mfma(kf.x, qf.x, st);
  st = ga_mfma(kf.y, qf.y, st);
  st = ga_mfma(kf.z, qf.z, st);
  st = ga_mfma(kf.w, qf.w, st);
  return st;
}

// y = gamma * softmax(q k^T) v + x; saves o = softmax(q k^T) v and lse = log sum exp per query
__global__ __launch_bounds__(256) void ga_sgam_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                          const float* __restrict__ v, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, float* __restrict__ y,
                                                          float* __restrict__ osave, float* __restrict__ lse, int hw) {
  __shared__ __attribute__((aligned(16))) float kt[2][32 * GA_DK];
  __shared__ __attribute__((aligned(16))) float vt[2][32 * GA_VS];
  const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const size_t base = (size_t)img * hw;
  const int qi = blockIdx.x * GA_PIX + wave * 32 + l31;
  const bool qvalid = qi < hw;
  const float4 qf = qvalid ? *reinterpret_cast<const float4*>(q + (base + qi) * GA_DK + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int nt = (hw + 31) >> 5;
  GaStage st_;
  ga_stage_load(st_, v, k, base, 0, hw, tid, 1.f);
  ga_stage_store(st_, vt[0], kt[0], tid);
  __syncthreads();
  f32x16 ot[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[u][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int t = 0; t < nt; ++t) {
    const int b = t & 1, k0 = t * 32;
    if (t + 1 < nt) ga_stage_load(st_, v, k, base, k0 + 32, hw, tid, 1.f);
    f32x16 p = ga_scores_t(kt[b], qf, h, l31);
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (k0 + ga_row(r, h) >= hw) p[r] = -INFINITY;
      tmax = fmaxf(tmax, p[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(m, tmax);
    const float alpha = expf(m - mnew);
    m = mnew;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = expf(p[r] - mnew);
      psum += p[r];
    }
    l = l * alpha + psum;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[u][r] *= alpha;
    // O^T[ch][query] += sum_key V[key][ch] P[query][key]: A = V row of key ga_row(s, h), B = register s of P
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float* vr = vt[b] + ga_row(s, h) * GA_VS + l31;
      ot[0] = ga_mfma(vr[0], p[s], ot[0]);
      ot[1] = ga_mfma(vr[32], p[s], ot[1]);
    }
    if (t + 1 < nt) ga_stage_store(st_, vt[b ^ 1], kt[b ^ 1], tid);
    __syncthreads();
  }
  l += __shfl_xor(l, 32, 64);
  if (!qvalid) return;
  const float inv = 1.f / l, g = gamma[0];
  if (h == 0) lse[base + qi] = m + logf(l);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const size_t idx = (base + qi) * GA_C + 32 * u + 8 * gq + 4 * h;
      const float4 o4 = make_float4(ot[u][4 * gq] * inv, ot[u][4 * gq + 1] * inv, ot[u][4 * gq + 2] * inv, ot[u][4 * gq + 3] * inv);
      const float4 x4 = *reinterpret_cast<const float4*>(x + idx);
      *reinterpret_cast<float4*>(osave + idx) = o4;
      *reinterpret_cast<float4*>(y + idx) = make_float4(g * o4.x + x4.x, g * o4.y + x4.y, g * o4.z + x4.z, g * o4.w + x4.w);
    }
}

// rowdot[p] = sum_c dy[p][c] * o[p][c]; dgpart[block] = the block's sum of rowdot (dgamma = sum of all)
__global__ __launch_bounds__(256) void ga_sgam_bwd_prep_kernel(const float* __restrict__ dy, const float* __restrict__ o,
                                                               float* __restrict__ rowdot, float* __restrict__ dgpart, long npix) {
  __shared__ float red[4];
  const int tid = threadIdx.x, sub = tid & 15;
  float tot = 0.f;
  for (long p = (long)blockIdx.x * 16 + (tid >> 4); p < npix; p += (long)gridDim.x * 16) {
    const float4 a = *reinterpret_cast<const float4*>(dy + p * GA_C + sub * 4);
    const float4 b = *reinterpret_cast<const float4*>(o + p * GA_C + sub * 4);
    float d = (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off, 64);
    if (sub == 0) {
      rowdot[p] = d;
      tot += d;
    }
  }
  tot = wave_sum(tot);
  if ((tid & 63) == 0) red[tid >> 6] = tot;
  __syncthreads();
  if (tid == 0) dgpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// query-tile owner: dq[query][d] = sum_key dS[query][key] k[key][d], dS = P * (dP - D), dP = dout . v, D = gamma * rowdot
__global__ __launch_bounds__(256) void ga_sgam_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, const float* __restrict__ dy,
                                                             const float* __restrict__ lse, const float* __restrict__ rowdot,
                                                             const float* __restrict__ gamma, float* __restrict__ dq, int hw) {
  __shared__ __attribute__((aligned(16))) float kt[2][32 * GA_DK];
  __shared__ __attribute__((aligned(16))) float vt[2][32 * GA_VS];
  const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const size_t base = (size_t)img * hw;
  const int qi = blockIdx.x * GA_PIX + wave * 32 + l31;
  const bool qvalid = qi < hw;
  const float g = gamma[0];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 qf = qvalid ? *reinterpret_cast<const float4*>(q + (base + qi) * GA_DK + 4 * h) : z4;
  const float lq = qvalid ? lse[base + qi] : 0.f;
  const float dq_ = qvalid ? g * rowdot[base + qi] : 0.f;
  float4 dof[8];                                         // dout[query][32h + 4i + e] = gamma * dy
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float4 t4 = qvalid ? *reinterpret_cast<const float4*>(dy + (base + qi) * GA_C + 32 * h + 4 * i) : z4;
    t4.x *= g; t4.y *= g; t4.z *= g; t4.w *= g;
    dof[i] = t4;
  }
  const int nt = (hw + 31) >> 5;
  GaStage st_;
  ga_stage_load(st_, v, k, base, 0, hw, tid, 1.f);
  ga_stage_store(st_, vt[0], kt[0], tid);
  __syncthreads();
  f32x16 dqt;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqt[r] = 0.f;
  for (int t = 0; t < nt; ++t) {
    const int b = t & 1, k0 = t * 32;
    if (t + 1 < nt) ga_stage_load(st_, v, k, base, k0 + 32, hw, tid, 1.f);
    f32x16 p = ga_scores_t(kt[b], qf, h, l31);
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = (k0 + ga_row(r, h) < hw) ? expf(p[r] - lq) : 0.f;
    // dP^T[key][query] = sum_ch V[key][ch] dout[query][ch]   (ch = 32h + 4i + e on both operands)
    f32x16 dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) dpt[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v4 = *reinterpret_cast<const float4*>(vt[b] + l31 * GA_VS + 32 * h + 4 * i);
      dpt = ga_mfma(v4.x, dof[i].x, dpt);
      dpt = ga_mfma(v4.y, dof[i].y, dpt);
      dpt = ga_mfma(v4.z, dof[i].z, dpt);
      dpt = ga_mfma(v4.w, dof[i].w, dpt);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = p[r] * (dpt[r] - dq_);       // dS^T
    // dQ^T[d][query] += sum_key K[key][d] dS[query][key]: rows d >= 8 of the tile are don't-care
#pragma unroll
    for (int s = 0; s < 16; ++s) dqt = ga_mfma(kt[b][ga_row(s, h) * GA_DK + (l31 & 7)], p[s], dqt);
    if (t + 1 < nt) ga_stage_store(st_, vt[b ^ 1], kt[b ^ 1], tid);
    __syncthreads();
  }
  if (qvalid) *reinterpret_cast<float4*>(dq + (base + qi) * GA_DK + 4 * h) = make_float4(dqt[0], dqt[1], dqt[2], dqt[3]);
}

// key-tile owner: dv[key][ch] = sum_query P dout, dk[key][d] = sum_query dS q
__global__ __launch_bounds__(256) void ga_sgam_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const float* __restrict__ dy,
                                                              const float* __restrict__ lse, const float* __restrict__ rowdot,
                                                              const float* __restrict__ gamma, float* __restrict__ dk,
                                                              float* __restrict__ dv, int hw) {
  __shared__ __attribute__((aligned(16))) float qt[2][32 * GA_DK];
  __shared__ __attribute__((aligned(16))) float dot[2][32 * GA_VS];
  __shared__ __attribute__((aligned(16))) float lt[2][32];
  __shared__ __attribute__((aligned(16))) float dt[2][32];
  const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const size_t base = (size_t)img * hw;
  const int kj = blockIdx.x * GA_PIX + wave * 32 + l31;
  const bool kvalid = kj < hw;
  const float g = gamma[0];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 kf = kvalid ? *reinterpret_cast<const float4*>(k + (base + kj) * GA_DK + 4 * h) : z4;
  float4 vf[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) vf[i] = kvalid ? *reinterpret_cast<const float4*>(v + (base + kj) * GA_C + 32 * h + 4 * i) : z4;
  const int nt = (hw + 31) >> 5;
  GaStage st_;
  float sl = 0.f;                                        // threads 64..95: lse, 96..127: D = gamma * rowdot
  auto load = [&](int r0) {
    ga_stage_load(st_, dy, q, base, r0, hw, tid, g);
    if (tid >= 64 && tid < 96) {
      const int row = r0 + tid - 64;
      sl = row < hw ? lse[base + row] : INFINITY;      // exp(s - inf) = 0: padded queries carry no probability
    } else if (tid >= 96 && tid < 128) {
      const int row = r0 + tid - 96;
      sl = row < hw ? g * rowdot[base + row] : 0.f;
    }
  };
  auto store = [&](int b) {
    ga_stage_store(st_, dot[b], qt[b], tid);
    if (tid >= 64 && tid < 96) lt[b][tid - 64] = sl;
    else if (tid >= 96 && tid < 128) dt[b][tid - 96] = sl;
  };
  load(0);
  store(0);
  __syncthreads();
  f32x16 dvt[2], dkt;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    dvt[0][r] = 0.f;
    dvt[1][r] = 0.f;
    dkt[r] = 0.f;
  }
  for (int t = 0; t < nt; ++t) {
    const int b = t & 1;
    if (t + 1 < nt) load(t * 32 + 32);
    // S[query][key]: A = Q row of query l31 (d = 4h + s), B = this lane's key
    const float4 q4 = *reinterpret_cast<const float4*>(qt[b] + l31 * GA_DK + 4 * h);
    f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = 0.f;
    p = ga_mfma(q4.x, kf.x, p);
    p = ga_mfma(q4.y, kf.y, p);
    p = ga_mfma(q4.z, kf.z, p);
    p = ga_mfma(q4.w, kf.w, p);
    float dd[16];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {                     // rows 4gq + e of this lane are queries 8gq + 4h + e
      const float4 l4 = *reinterpret_cast<const float4*>(lt[b] + 8 * gq + 4 * h);
      const float4 d4 = *reinterpret_cast<const float4*>(dt[b] + 8 * gq + 4 * h);
      p[4 * gq] = expf(p[4 * gq] - l4.x);
      p[4 * gq + 1] = expf(p[4 * gq + 1] - l4.y);
      p[4 * gq + 2] = expf(p[4 * gq + 2] - l4.z);
      p[4 * gq + 3] = expf(p[4 * gq + 3] - l4.w);
      dd[4 * gq] = d4.x; dd[4 * gq + 1] = d4.y; dd[4 * gq + 2] = d4.z; dd[4 * gq + 3] = d4.w;
    }
    // dP[query][key] = sum_ch dout[query][ch] V[key][ch]
    f32x16 dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 o4 = *reinterpret_cast<const float4*>(dot[b] + l31 * GA_VS + 32 * h + 4 * i);
      dp = ga_mfma(o4.x, vf[i].x, dp);
      dp = ga_mfma(o4.y, vf[i].y, dp);
      dp = ga_mfma(o4.z, vf[i].z, dp);
      dp = ga_mfma(o4.w, vf[i].w, dp);
    }
    // dV^T[ch][key] += sum_query dout[query][ch] P[query][key]
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float* dr = dot[b] + ga_row(s, h) * GA_VS + l31;
      dvt[0] = ga_mfma(dr[0], p[s], dvt[0]);
      dvt[1] = ga_mfma(dr[32], p[s], dvt[1]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = p[r] * (dp[r] - dd[r]);      // dS
    // dK^T[d][key] += sum_query Q[query][d] dS[query][key]
#pragma unroll
    for (int s = 0; s < 16; ++s) dkt = ga_mfma(qt[b][ga_row(s, h) * GA_DK + (l31 & 7)], p[s], dkt);
    if (t + 1 < nt) store(b ^ 1);
    __syncthreads();
  }
  if (!kvalid) return;
  *reinterpret_cast<float4*>(dk + (base + kj) * GA_DK + 4 * h) = make_float4(dkt[0], dkt[1], dkt[2], dkt[3]);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      *reinterpret_cast<float4*>(dv + (base + kj) * GA_C + 32 * u + 8 * gq + 4 * h) =
          make_float4(dvt[u][4 * gq], dvt[u][4 * gq + 1], dvt[u][4 * gq + 2], dvt[u][4 * gq + 3]);
}

// ================================================================================================ //
// SGAM in split-bf16 (SRHIP_MATH_BF16X3 / SRHIP_MATH_HALF): the q.k energies, the soft-max statistics and every
// accumulator stay fp32 (the energies feed an exp()); the four big products per 32 x 32 tile pair -- P.V, dP = dout.V^T,
// dV = P^T.dout, dQ / dK = dS.K / dS^T.Q -- run as a*b ~= ah*bh + ah*bl + al*bh on v_mfma_f32_32x32x16_bf16 like the
// convolutions (fp32 MFMA: 172 instructions of 64 cycles per tile pair over the three kernels; here 12 fp32 + 126 bf16
// of 32 cycles).  The register trick carries over: a C/D register file [16 rows of ONE column per lane] is directly the
// B operand of a bf16 MFMA if the contraction index runs over the tile's rows in the order
//     k slot 16 s + 8 h + j  <->  tile row ga_row(8 s + j, h)          (s = MFMA step, h = lane half, j = 0..7),
// so the OTHER operand -- V^T, dout^T, K^T, Q^T -- is staged in LDS transposed with its 32 rows in exactly that order
// (ga_pos), 8 bf16 "hi" resp. "lo" values per 16-byte fragment read.  Products contracted over channels (dP) read
// row-major bf16 tiles.  hi = bf16(v), lo = bf16(v - hi), computed once per element when a tile is staged.
// ================================================================================================ //
typedef __bf16 ga_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ga_bf16x4 __attribute__((ext_vector_type(4)));
constexpr int GA_TS = 40;       // bf16 stride of a transposed tile row  [ch or d][32 slots]   (80 B: conflict-free b128 reads)
constexpr int GA_RS = 72;       // bf16 stride of a row-major tile row   [32 rows][64 ch]      (144 B: conflict-free b128 reads)

__device__ inline f32x16 ga_mfma16(const ga_bf16x8& a, const ga_bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// k slot of tile row `row` (0..31): the inverse of slot -> ga_row(8 s + j, h)
__device__ __forceinline__ int ga_pos(int row) {
  const int t = row & 15;
  return (row & 16) + 8 * ((t >> 2) & 1) + (t & 3) + 4 * (t >> 3);
}
// registers 8 s .. 8 s + 7 of a C/D file as a split operand
__device__ __forceinline__ void ga_split_regs(const f32x16& p, int s, ga_bf16x8& hi, ga_bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float v = p[8 * s + j];
    const __bf16 hh = (__bf16)v;
    hi[j] = hh;
    lo[j] = (__bf16)(v - (float)hh);
  }
}
__device__ __forceinline__ void ga_split_f4x2(const float4& a, const float4& b, ga_bf16x8& hi, ga_bf16x8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 hh = (__bf16)v[j];
    hi[j] = hh;
    lo[j] = (__bf16)(v[j] - (float)hh);
  }
}
// acc += A.B in split-bf16 (the small cross terms first)
__device__ __forceinline__ f32x16 ga_mma3(const ga_bf16x8& ah, const ga_bf16x8& al, const ga_bf16x8& bh, const ga_bf16x8& bl, f32x16 c) {
  c = ga_mfma16(al, bh, c);
  c = ga_mfma16(ah, bl, c);
  return ga_mfma16(ah, bh, c);
}
// staged wide tile [32][64] -> transposed split tiles th / tl [64][GA_TS] (rows in ga_pos order)
__device__ __forceinline__ void ga_store_wide_T(const GaStage& s, __bf16* th, __bf16* tl, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i, pos = ga_pos(e >> 4), c0 = (e & 15) * 4;
    const float v[4] = {s.w[i].x, s.w[i].y, s.w[i].z, s.w[i].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const __bf16 hh = (__bf16)v[c];
      th[(c0 + c) * GA_TS + pos] = hh;
      tl[(c0 + c) * GA_TS + pos] = (__bf16)(v[c] - (float)hh);
    }
  }
}
// staged wide tile [32][64] -> row-major split tiles rh / rl [32][GA_RS]
__device__ __forceinline__ void ga_store_wide_R(const GaStage& s, __bf16* rh, __bf16* rl, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + 256 * i, row = e >> 4, c0 = (e & 15) * 4;
    const float v[4] = {s.w[i].x, s.w[i].y, s.w[i].z, s.w[i].w};
    ga_bf16x4 hi, lo;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const __bf16 hh = (__bf16)v[c];
      hi[c] = hh;
      lo[c] = (__bf16)(v[c] - (float)hh);
    }
    *reinterpret_cast<ga_bf16x4*>(rh + row * GA_RS + c0) = hi;
    *reinterpret_cast<ga_bf16x4*>(rl + row * GA_RS + c0) = lo;
  }
}
// staged narrow tile [32][8]: fp32 copy nt (energies) and, when th != nullptr, transposed split tiles [8][GA_TS]
__device__ __forceinline__ void ga_store_narrow(const GaStage& s, float* nt, __bf16* th, __bf16* tl, int tid) {
  if (tid < 64) {
    *reinterpret_cast<float4*>(nt + (tid >> 1) * GA_DK + (tid & 1) * 4) = s.n;
    if (th != nullptr) {
      const int pos = ga_pos(tid >> 1), d0 = (tid & 1) * 4;
      const float v[4] = {s.n.x, s.n.y, s.n.z, s.n.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const __bf16 hh = (__bf16)v[c];
        th[(d0 + c) * GA_TS + pos] = hh;
        tl[(d0 + c) * GA_TS + pos] = (__bf16)(v[c] - (float)hh);
      }
    }
  }
}
// exp() of the split-bf16 kernels: v_exp_f32(x * log2(e)) (two instructions; libm's expf is ~15 and the three kernels call it
// 16 times per lane and tile).  Error <= |x| * 2^-24 relative, i.e. < 1e-6 wherever the probability is not negligible.
__device__ __forceinline__ float ga_exp(float x) { return __expf(x); }
#define GA_FRAG(ptr) (*reinterpret_cast<const ga_bf16x8*>(ptr))

__global__ __launch_bounds__(256) void ga_sgam_fwd_x3_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const float* __restrict__ v, const float* __restrict__ x,
                                                             const float* __restrict__ gamma, float* __restrict__ y,
                                                             float* __restrict__ osave, float* __restrict__ lse, int hw) {
  __shared__ __attribute__((aligned(16))) float kt[2][32 * GA_DK];
  __shared__ __attribute__((aligned(16))) __bf16 vth[2][GA_C * GA_TS];
  __shared__ __attribute__((aligned(16))) __bf16 vtl[2][GA_C * GA_TS];
  const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const size_t base = (size_t)img * hw;
  const int qi = blockIdx.x * GA_PIX + wave * 32 + l31;
  const bool qvalid = qi < hw;
  const float4 qf = qvalid ? *reinterpret_cast<const float4*>(q + (base + qi) * GA_DK + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
  const int nt = (hw + 31) >> 5;
  GaStage st_;
  ga_stage_load(st_, v, k, base, 0, hw, tid, 1.f);
  ga_store_wide_T(st_, vth[0], vtl[0], tid);
  ga_store_narrow(st_, kt[0], nullptr, nullptr, tid);
  __syncthreads();
  f32x16 ot[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[u][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  for (int t = 0; t < nt; ++t) {
    const int b = t & 1, k0 = t * 32;
    if (t + 1 < nt) ga_stage_load(st_, v, k, base, k0 + 32, hw, tid, 1.f);
    f32x16 p = ga_scores_t(kt[b], qf, h, l31);
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (k0 + ga_row(r, h) >= hw) p[r] = -INFINITY;
      tmax = fmaxf(tmax, p[r]);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float mnew = fmaxf(m, tmax);
    const float alpha = ga_exp(m - mnew);
    m = mnew;
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = ga_exp(p[r] - mnew);
      psum += p[r];
    }
    l = l * alpha + psum;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[u][r] *= alpha;
    // O^T[ch][query] += sum_key V^T[ch][key] P[key][query]: A = transposed V fragment, B = registers 8 s .. 8 s + 7 of P
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ga_bf16x8 ph, pl;
      ga_split_regs(p, s, ph, pl);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int off = (32 * u + l31) * GA_TS + 16 * s + 8 * h;
        ot[u] = ga_mma3(GA_FRAG(vth[b] + off), GA_FRAG(vtl[b] + off), ph, pl, ot[u]);
      }
    }
    if (t + 1 < nt) {
      ga_store_wide_T(st_, vth[b ^ 1], vtl[b ^ 1], tid);
      ga_store_narrow(st_, kt[b ^ 1], nullptr, nullptr, tid);
    }
    __syncthreads();
  }
  l += __shfl_xor(l, 32, 64);
  if (!qvalid) return;
  const float inv = 1.f / l, g = gamma[0];
  if (h == 0) lse[base + qi] = m + logf(l);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const size_t idx = (base + qi) * GA_C + 32 * u + 8 * gq + 4 * h;
      const float4 o4 = make_float4(ot[u][4 * gq] * inv, ot[u][4 * gq + 1] * inv, ot[u][4 * gq + 2] * inv, ot[u][4 * gq + 3] * inv);
      const float4 x4 = *reinterpret_cast<const float4*>(x + idx);
      *reinterpret_cast<float4*>(osave + idx) = o4;
      *reinterpret_cast<float4*>(y + idx) = make_float4(g * o4.x + x4.x, g * o4.y + x4.y, g * o4.z + x4.z, g * o4.w + x4.w);
    }
}

// query-tile owner, split-bf16: dq[query][d] = sum_key dS[query][key] k[key][d]
__global__ __launch_bounds__(256) void ga_sgam_bwd_dq_x3_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ dy,
                                                                const float* __restrict__ lse, const float* __restrict__ rowdot,
                                                                const float* __restrict__ gamma, float* __restrict__ dq, int hw) {
  __shared__ __attribute__((aligned(16))) float kt[2][32 * GA_DK];
  __shared__ __attribute__((aligned(16))) __bf16 kth[2][GA_DK * GA_TS];
  __shared__ __attribute__((aligned(16))) __bf16 ktl[2][GA_DK * GA_TS];
  __shared__ __attribute__((aligned(16))) __bf16 vrh[2][32 * GA_RS];
  __shared__ __attribute__((aligned(16))) __bf16 vrl[2][32 * GA_RS];
  const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const size_t base = (size_t)img * hw;
  const int qi = blockIdx.x * GA_PIX + wave * 32 + l31;
  const bool qvalid = qi < hw;
  const float g = gamma[0];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 qf = qvalid ? *reinterpret_cast<const float4*>(q + (base + qi) * GA_DK + 4 * h) : z4;
  const float lq = qvalid ? lse[base + qi] : 0.f;
  const float dq_ = qvalid ? g * rowdot[base + qi] : 0.f;
  ga_bf16x8 doh[4], dol[4];                              // dout[query][16 c + 8 h + j] = gamma * dy, split once
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float4 a = qvalid ? *reinterpret_cast<const float4*>(dy + (base + qi) * GA_C + 16 * c + 8 * h) : z4;
    float4 b = qvalid ? *reinterpret_cast<const float4*>(dy + (base + qi) * GA_C + 16 * c + 8 * h + 4) : z4;
    a.x *= g; a.y *= g; a.z *= g; a.w *= g;
    b.x *= g; b.y *= g; b.z *= g; b.w *= g;
    ga_split_f4x2(a, b, doh[c], dol[c]);
  }
  const int nt = (hw + 31) >> 5;
  GaStage st_;
  ga_stage_load(st_, v, k, base, 0, hw, tid, 1.f);
  ga_store_wide_R(st_, vrh[0], vrl[0], tid);
  ga_store_narrow(st_, kt[0], kth[0], ktl[0], tid);
  __syncthreads();
  f32x16 dqt;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqt[r] = 0.f;
  for (int t = 0; t < nt; ++t) {
    const int b = t & 1, k0 = t * 32;
    if (t + 1 < nt) ga_stage_load(st_, v, k, base, k0 + 32, hw, tid, 1.f);
    f32x16 p = ga_scores_t(kt[b], qf, h, l31);
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = (k0 + ga_row(r, h) < hw) ? ga_exp(p[r] - lq) : 0.f;
    // dP^T[key][query] = sum_ch V[key][ch] dout[query][ch]: A = row-major V fragment of key l31, B = this lane's dout
    f32x16 dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) dpt[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int off = l31 * GA_RS + 16 * c + 8 * h;
      dpt = ga_mma3(GA_FRAG(vrh[b] + off), GA_FRAG(vrl[b] + off), doh[c], dol[c], dpt);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = p[r] * (dpt[r] - dq_);       // dS^T
    // dQ^T[d][query] += sum_key K^T[d][key] dS^T[key][query]: rows d >= 8 of the tile are don't-care
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ga_bf16x8 sh, sl;
      ga_split_regs(p, s, sh, sl);
      const int off = (l31 & 7) * GA_TS + 16 * s + 8 * h;
      dqt = ga_mma3(GA_FRAG(kth[b] + off), GA_FRAG(ktl[b] + off), sh, sl, dqt);
    }
    if (t + 1 < nt) {
      ga_store_wide_R(st_, vrh[b ^ 1], vrl[b ^ 1], tid);
      ga_store_narrow(st_, kt[b ^ 1], kth[b ^ 1], ktl[b ^ 1], tid);
    }
    __syncthreads();
  }
  if (qvalid) *reinterpret_cast<float4*>(dq + (base + qi) * GA_DK + 4 * h) = make_float4(dqt[0], dqt[1], dqt[2], dqt[3]);
}

// key-tile owner, split-bf16: dv[key][ch] = sum_query P dout, dk[key][d] = sum_query dS q
__global__ __launch_bounds__(256) void ga_sgam_bwd_dkv_x3_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, const float* __restrict__ dy,
                                                                 const float* __restrict__ lse, const float* __restrict__ rowdot,
                                                                 const float* __restrict__ gamma, float* __restrict__ dk,
                                                                 float* __restrict__ dv, int hw) {
  __shared__ __attribute__((aligned(16))) float qt[2][32 * GA_DK];
  __shared__ __attribute__((aligned(16))) __bf16 qth[2][GA_DK * GA_TS];
  __shared__ __attribute__((aligned(16))) __bf16 qtl[2][GA_DK * GA_TS];
  __shared__ __attribute__((aligned(16))) __bf16 drh[2][32 * GA_RS];     // dout tile, row-major (dP)
  __shared__ __attribute__((aligned(16))) __bf16 drl[2][32 * GA_RS];
  __shared__ __attribute__((aligned(16))) __bf16 dth[2][GA_C * GA_TS];   // dout tile, transposed (dV)
  __shared__ __attribute__((aligned(16))) __bf16 dtl[2][GA_C * GA_TS];
  __shared__ __attribute__((aligned(16))) float lt[2][32];
  __shared__ __attribute__((aligned(16))) float dt[2][32];
  const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const size_t base = (size_t)img * hw;
  const int kj = blockIdx.x * GA_PIX + wave * 32 + l31;
  const bool kvalid = kj < hw;
  const float g = gamma[0];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 kf = kvalid ? *reinterpret_cast<const float4*>(k + (base + kj) * GA_DK + 4 * h) : z4;
  ga_bf16x8 vfh[4], vfl[4];                              // V[key][16 c + 8 h + j], split once
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float4 a = kvalid ? *reinterpret_cast<const float4*>(v + (base + kj) * GA_C + 16 * c + 8 * h) : z4;
    const float4 b = kvalid ? *reinterpret_cast<const float4*>(v + (base + kj) * GA_C + 16 * c + 8 * h + 4) : z4;
    ga_split_f4x2(a, b, vfh[c], vfl[c]);
  }
  const int nt = (hw + 31) >> 5;
  GaStage st_;
  float sl = 0.f;                                        // threads 64..95: lse, 96..127: D = gamma * rowdot
  auto load = [&](int r0) {
    ga_stage_load(st_, dy, q, base, r0, hw, tid, g);
    if (tid >= 64 && tid < 96) {
      const int row = r0 + tid - 64;
      sl = row < hw ? lse[base + row] : INFINITY;      // exp(s - inf) = 0: padded queries carry no probability
    } else if (tid >= 96 && tid < 128) {
      const int row = r0 + tid - 96;
      sl = row < hw ? g * rowdot[base + row] : 0.f;
    }
  };
  auto store = [&](int b) {
    ga_store_wide_R(st_, drh[b], drl[b], tid);
    ga_store_wide_T(st_, dth[b], dtl[b], tid);
    ga_store_narrow(st_, qt[b], qth[b], qtl[b], tid);
    if (tid >= 64 && tid < 96) lt[b][tid - 64] = sl;
    else if (tid >= 96 && tid < 128) dt[b][tid - 96] = sl;
  };
  load(0);
  store(0);
  __syncthreads();
  f32x16 dvt[2], dkt;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    dvt[0][r] = 0.f;
    dvt[1][r] = 0.f;
    dkt[r] = 0.f;
  }
  for (int t = 0; t < nt; ++t) {
    const int b = t & 1;
    if (t + 1 < nt) load(t * 32 + 32);
    // S[query][key]: A = Q row of query l31 (d = 4h + s), B = this lane's key
    const float4 q4 = *reinterpret_cast<const float4*>(qt[b] + l31 * GA_DK + 4 * h);
    f32x16 p;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = 0.f;
    p = ga_mfma(q4.x, kf.x, p);
    p = ga_mfma(q4.y, kf.y, p);
    p = ga_mfma(q4.z, kf.z, p);
    p = ga_mfma(q4.w, kf.w, p);
    float dd[16];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {                     // rows 4gq + e of this lane are queries 8gq + 4h + e
      const float4 l4 = *reinterpret_cast<const float4*>(lt[b] + 8 * gq + 4 * h);
      const float4 d4 = *reinterpret_cast<const float4*>(dt[b] + 8 * gq + 4 * h);
      p[4 * gq] = ga_exp(p[4 * gq] - l4.x);
      p[4 * gq + 1] = ga_exp(p[4 * gq + 1] - l4.y);
      p[4 * gq + 2] = ga_exp(p[4 * gq + 2] - l4.z);
      p[4 * gq + 3] = ga_exp(p[4 * gq + 3] - l4.w);
      dd[4 * gq] = d4.x; dd[4 * gq + 1] = d4.y; dd[4 * gq + 2] = d4.z; dd[4 * gq + 3] = d4.w;
    }
    // dP[query][key] = sum_ch dout[query][ch] V[key][ch]: A = row-major dout fragment of query l31, B = this lane's V row
    f32x16 dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int off = l31 * GA_RS + 16 * c + 8 * h;
      dp = ga_mma3(GA_FRAG(drh[b] + off), GA_FRAG(drl[b] + off), vfh[c], vfl[c], dp);
    }
    // dV^T[ch][key] += sum_query dout^T[ch][query] P[query][key]: A = transposed dout fragment, B = registers of P
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ga_bf16x8 ph, pl;
      ga_split_regs(p, s, ph, pl);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int off = (32 * u + l31) * GA_TS + 16 * s + 8 * h;
        dvt[u] = ga_mma3(GA_FRAG(dth[b] + off), GA_FRAG(dtl[b] + off), ph, pl, dvt[u]);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = p[r] * (dp[r] - dd[r]);      // dS
    // dK^T[d][key] += sum_query Q^T[d][query] dS[query][key]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ga_bf16x8 sh, sl2;
      ga_split_regs(p, s, sh, sl2);
      const int off = (l31 & 7) * GA_TS + 16 * s + 8 * h;
      dkt = ga_mma3(GA_FRAG(qth[b] + off), GA_FRAG(qtl[b] + off), sh, sl2, dkt);
    }
    if (t + 1 < nt) store(b ^ 1);
    __syncthreads();
  }
  if (!kvalid) return;
  *reinterpret_cast<float4*>(dk + (base + kj) * GA_DK + 4 * h) = make_float4(dkt[0], dkt[1], dkt[2], dkt[3]);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq)
      *reinterpret_cast<float4*>(dv + (base + kj) * GA_C + 32 * u + 8 * gq + 4 * h) =
          make_float4(dvt[u][4 * gq], dvt[u][4 * gq + 1], dvt[u][4 * gq + 2], dvt[u][4 * gq + 3]);
}

int g_sgam_cfg = 0;   // srhip_debug_set(4, v): 1 = the exact-fp32 SGAM kernels in every arithmetic mode (A/B and tests)
extern int g_conv_math;
static inline bool ga_split_math() { return g_conv_math >= 1 && g_sgam_cfg != 1; }

static inline int ga_nsplit(int hw) { return cdiv(hw, GA_PIX); }
static inline int ga_prep_blocks(long npix) {
  long b = (npix + 15) / 16;
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace srhip

using namespace srhip;

extern "C" {

size_t srhip_cgam_workspace(int n, int hw) {
  if (n <= 0 || hw <= 0) return 0;
  // gram partials [n][nsplit][64][64] + msym [n][64][64] + dgamma partials [n]
  return ((size_t)n * ga_nsplit(hw) * GA_C * GA_C + (size_t)n * GA_C * GA_C + (size_t)n + 64) * sizeof(float);
}

int srhip_cgam_fwd(const float* x, const float* gamma, float* y, float* att, void* workspace, size_t workspace_bytes,
                   int n, int hw, int c, void* stream) {
  SRHIP_REQUIRE(c == GA_C, "cgam_fwd: C must be 64 (sradsgan.py:194), got %d", c);
  SRHIP_REQUIRE(n > 0 && hw > 0, "cgam_fwd: empty input");
  SRHIP_REQUIRE((((uintptr_t)x | (uintptr_t)y | (uintptr_t)att) & 15) == 0, "cgam_fwd: pointers must be 16-byte aligned");
  if (!workspace || workspace_bytes < srhip_cgam_workspace(n, hw)) {
    set_error("cgam_fwd: workspace %zu bytes < required %zu", workspace_bytes, srhip_cgam_workspace(n, hw));
    return SRHIP_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int ns = ga_nsplit(hw);
  float* part = static_cast<float*>(workspace);
  hipLaunchKernelGGL(ga_gram_partial_kernel, dim3(ns, n), dim3(256), 0, st, x, x, part, hw, ns);
  hipLaunchKernelGGL(ga_cgam_softmax_kernel, dim3(n), dim3(256), 0, st, part, att, ns);
  hipLaunchKernelGGL(ga_cgam_apply_kernel, dim3(ns, n), dim3(256), 0, st, x, att, gamma, y, hw);
  return check_launch("cgam_fwd");
}

int srhip_cgam_bwd(const float* dy, const float* x, const float* att, const float* gamma, float* dx, float* dgamma,
                   int accumulate_dgamma, void* workspace, size_t workspace_bytes, int n, int hw, int c, void* stream) {
  SRHIP_REQUIRE(c == GA_C, "cgam_bwd: C must be 64, got %d", c);
  SRHIP_REQUIRE(n > 0 && hw > 0, "cgam_bwd: empty input");
  SRHIP_REQUIRE((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)att) & 15) == 0, "cgam_bwd: pointers must be 16-byte aligned");
  if (!workspace || workspace_bytes < srhip_cgam_workspace(n, hw)) {
    set_error("cgam_bwd: workspace %zu bytes < required %zu", workspace_bytes, srhip_cgam_workspace(n, hw));
    return SRHIP_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const int ns = ga_nsplit(hw);
  float* part = static_cast<float*>(workspace);
  float* msym = part + (size_t)n * ns * GA_C * GA_C;
  float* dgpart = msym + (size_t)n * GA_C * GA_C;
  hipLaunchKernelGGL(ga_gram_partial_kernel, dim3(ns, n), dim3(256), 0, st, dy, x, part, hw, ns);
  hipLaunchKernelGGL(ga_cgam_bwd_small_kernel, dim3(n), dim3(256), 0, st, part, att, gamma, msym, dgpart, ns);
  if (dgamma) hipLaunchKernelGGL(ga_sum_small_kernel, dim3(1), dim3(64), 0, st, dgpart, n, dgamma, accumulate_dgamma);
  hipLaunchKernelGGL(ga_cgam_bwd_apply_kernel, dim3(ns, n), dim3(256), 0, st, dy, x, att, msym, gamma, dx, hw);
  return check_launch("cgam_bwd");
}

int srhip_sgam_flash_fwd(const float* q, const float* k, const float* v, const float* x, const float* gamma, float* y,
                         float* o, float* lse, int n, int hw, int dk, int c, void* stream) {
  SRHIP_REQUIRE(c == GA_C && dk == GA_DK, "sgam_flash_fwd: C must be 64 and q/k channels 8 (sradsgan.py:157-159), got %d / %d", c, dk);
  SRHIP_REQUIRE(n > 0 && hw > 0, "sgam_flash_fwd: empty input");
  SRHIP_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)x | (uintptr_t)y | (uintptr_t)o) & 15) == 0,
                "sgam_flash_fwd: pointers must be 16-byte aligned");
  if (ga_split_math())
    hipLaunchKernelGGL(ga_sgam_fwd_x3_kernel, dim3(cdiv(hw, GA_PIX), n), dim3(256), 0, as_stream(stream), q, k, v, x, gamma, y, o, lse, hw);
  else
    hipLaunchKernelGGL(ga_sgam_fwd_kernel, dim3(cdiv(hw, GA_PIX), n), dim3(256), 0, as_stream(stream), q, k, v, x, gamma, y, o, lse, hw);
  return check_launch("sgam_flash_fwd");
}

size_t srhip_sgam_flash_bwd_workspace(int n, int hw) {
  if (n <= 0 || hw <= 0) return 0;
  const long npix = (long)n * hw;
  return ((size_t)npix + (size_t)ga_prep_blocks(npix) + 64) * sizeof(float);
}

int srhip_sgam_flash_bwd(const float* dy, const float* q, const float* k, const float* v, const float* o, const float* lse,
                         const float* gamma, float* dq, float* dk_, float* dv, float* dgamma, int accumulate_dgamma,
                         void* workspace, size_t workspace_bytes, int n, int hw, int dk, int c, void* stream) {
  SRHIP_REQUIRE(c == GA_C && dk == GA_DK, "sgam_flash_bwd: C must be 64 and q/k channels 8, got %d / %d", c, dk);
  SRHIP_REQUIRE(n > 0 && hw > 0, "sgam_flash_bwd: empty input");
  SRHIP_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dy | (uintptr_t)o | (uintptr_t)dq | (uintptr_t)dk_ | (uintptr_t)dv) & 15) == 0,
                "sgam_flash_bwd: pointers must be 16-byte aligned");
  if (!workspace || workspace_bytes < srhip_sgam_flash_bwd_workspace(n, hw)) {
    set_error("sgam_flash_bwd: workspace %zu bytes < required %zu", workspace_bytes, srhip_sgam_flash_bwd_workspace(n, hw));
    return SRHIP_ERR_WORKSPACE;
  }
  hipStream_t st = as_stream(stream);
  const long npix = (long)n * hw;
  const int pb = ga_prep_blocks(npix);
  float* rowdot = static_cast<float*>(workspace);
  float* dgpart = rowdot + npix;
  hipLaunchKernelGGL(ga_sgam_bwd_prep_kernel, dim3(pb), dim3(256), 0, st, dy, o, rowdot, dgpart, npix);
  if (dgamma) hipLaunchKernelGGL(ga_sum_small_kernel, dim3(1), dim3(64), 0, st, dgpart, pb, dgamma, accumulate_dgamma);
  const dim3 grid(cdiv(hw, GA_PIX), n);
  if (ga_split_math()) {
    hipLaunchKernelGGL(ga_sgam_bwd_dq_x3_kernel, grid, dim3(256), 0, st, q, k, v, dy, lse, rowdot, gamma, dq, hw);
    hipLaunchKernelGGL(ga_sgam_bwd_dkv_x3_kernel, grid, dim3(256), 0, st, q, k, v, dy, lse, rowdot, gamma, dk_, dv, hw);
  } else {
    hipLaunchKernelGGL(ga_sgam_bwd_dq_kernel, grid, dim3(256), 0, st, q, k, v, dy, lse, rowdot, gamma, dq, hw);
    hipLaunchKernelGGL(ga_sgam_bwd_dkv_kernel, grid, dim3(256), 0, st, q, k, v, dy, lse, rowdot, gamma, dk_, dv, hw);
  }
  return check_launch("sgam_flash_bwd");
}

}  // extern "C"
