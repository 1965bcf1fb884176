// Implicit-GEMM 2-D convolution for gfx950 on the exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD, bitwise an fmaf chain -> fp32 parity with the
// reference's ATen path at ~1e-6, see DESIGN.md "precision").
//
//   fprop / dgrad :  D[m][n] = sum_k A[m][k] * B[k][n]
//        m = output pixel (n,ho,wo)      n = destination channel      k = (kh,kw,c_src)
//        A is gathered on the fly from the NHWC source (im2col never materialised),
//        B is the packed weight [KH*KW*Csrc][ld] (srhip_pack_weight).
//        dgrad is the same kernel with flipped/transposed weights and a "divide by stride"
//        gather (hnum = o - (KH-1-pad) + kh must be a multiple of the stride).
//   wgrad         :  dW[co][(kh,kw,ci)] = sum_pixels dy[pix][co] * window(x)[pix][(kh,kw,ci)]
//        both operands are pixel-major (k-major) in NHWC, so they stream straight into the
//        [k][m] / [k][n] LDS images; split-K over pixels, deterministic two-pass reduce.
//
// Tiling: 256 threads = 4 wave64s; block tile BM x BN, K step 16, double-buffered LDS with the next
// tile's global loads in flight during the MFMAs; every wave owns a (BM/WM)x(BN/WN) sub-tile built
// from 32x32 MFMA tiles.  LDS images are k-major so that a wave's MFMA operand fetch is 32
// consecutive dwords per half-wave (conflict-free ds_read_b32).
#include "common.h"

namespace srhip {
int g_headconv_rows = 4;   // srhip_debug_set(18, rows): image rows a block of headconv_fwd_kernel walks
}
namespace srhip {

constexpr int BK = 16;

struct ConvGeom {
  int N, H, W, C;    // source tensor: N x H x W x C (row stride ldx)
  int Ho, Wo, K;     // destination tensor: N x Ho x Wo x K (row stride ldy)
  int KH, KW;
  int so, pad, dv;   // source coordinate = (o*so - pad + k) / dv, valid only when divisible
  int ldx, ldy, ldr, ldw;
  int Ktot;          // KH*KW*C
  int M;             // N*Ho*Wo
  float slope;
  int flags;
  int accumulate;
};

__device__ inline f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// -------------------------------------------------------------------------------------------- //
template <int BM, int BN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(256) void igemm_fprop_kernel(const float* __restrict__ src,
                                                           const float* __restrict__ wt,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ residual,
                                                           const float* __restrict__ rowscale,
                                                           float* __restrict__ dst, ConvGeom g) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LDA = BM + 2, LDB = BN + 4;
  constexpr int AR = BM / 64;
  constexpr int BVEC = BK * BN / 4;            // float4s in one B tile
  constexpr int BV = (BVEC + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * LDA + 2 * BK * LDB];
  float* As = lds;
  float* Bs = lds + 2 * BK * LDA;

  const int tid = threadIdx.x;
  const int ntn = (g.K + BN - 1) / BN;
  const int tile_n = blockIdx.x % ntn;
  const int tile_m = blockIdx.x / ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-thread im2col row bookkeeping (fixed for the whole K loop) ----
  const int arow = tid >> 2, kq = tid & 3;
  int hb[AR], wb[AR], pb[AR];
  const int HoWo = g.Ho * g.Wo;
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    int m = m0 + arow + 64 * i;
    if (m < g.M) {
      int n = m / HoWo;
      int rem = m - n * HoWo;
      int ho = rem / g.Wo;
      int wo = rem - ho * g.Wo;
      hb[i] = ho * g.so - g.pad;
      wb[i] = wo * g.so - g.pad;
      pb[i] = n * g.H * g.W;
    } else {
      hb[i] = -(1 << 28);
      wb[i] = 0;
      pb[i] = 0;
    }
  }

  float4 ra[AR];
  float4 rb[BV];

  auto gather1 = [&](int i, int k) -> float {
    if (k >= g.Ktot) return 0.f;
    int tap = k / g.C;
    int c = k - tap * g.C;
    int kh = tap / g.KW;
    int kw = tap - kh * g.KW;
    int hn = hb[i] + kh, wn = wb[i] + kw;
    if (hn < 0 || wn < 0) return 0.f;
    int hs = hn, ws = wn;
    if (g.dv > 1) {
      hs = hn / g.dv;
      ws = wn / g.dv;
      if (hs * g.dv != hn || ws * g.dv != wn) return 0.f;
    }
    if (hs >= g.H || ws >= g.W) return 0.f;
    return src[(size_t)(pb[i] + hs * g.W + ws) * g.ldx + c];
  };

  auto load_tiles = [&](int kc) {
    const int k = kc * BK + kq * 4;
    if (VEC) {
      int tap = k / g.C;
      int c = k - tap * g.C;
      int kh = tap / g.KW;
      int kw = tap - kh * g.KW;
      const bool kok = k < g.Ktot;
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        int hn = hb[i] + kh, wn = wb[i] + kw;
        bool ok = kok && hn >= 0 && wn >= 0;
        int hs = hn, ws = wn;
        if (g.dv > 1) {
          hs = hn / g.dv;
          ws = wn / g.dv;
          ok = ok && (hs * g.dv == hn) && (ws * g.dv == wn);
        }
        ok = ok && hs < g.H && ws < g.W;
        ra[i] = ok ? *reinterpret_cast<const float4*>(src + (size_t)(pb[i] + hs * g.W + ws) * g.ldx + c)
                   : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        ra[i].x = gather1(i, k);
        ra[i].y = gather1(i, k + 1);
        ra[i].z = gather1(i, k + 2);
        ra[i].w = gather1(i, k + 3);
      }
    }
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      int idx = tid + 256 * j;
      int brow = idx / (BN / 4), bc = idx - brow * (BN / 4);
      int kk = kc * BK + brow, col = n0 + bc * 4;
      bool ok = (BVEC % 256 == 0 || idx < BVEC) && kk < g.Ktot && col < g.ldw;
      rb[j] = ok ? *reinterpret_cast<const float4*>(wt + (size_t)kk * g.ldw + col)
                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  auto store_tiles = [&](int buf) {
    float* a = As + buf * BK * LDA + (kq * 4) * LDA + arow;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      a[0 * LDA + 64 * i] = ra[i].x;
      a[1 * LDA + 64 * i] = ra[i].y;
      a[2 * LDA + 64 * i] = ra[i].z;
      a[3 * LDA + 64 * i] = ra[i].w;
    }
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      int idx = tid + 256 * j;
      if (BVEC % 256 == 0 || idx < BVEC) {
        int brow = idx / (BN / 4), bc = idx - brow * (BN / 4);
        *reinterpret_cast<float4*>(Bs + buf * BK * LDB + brow * LDB + bc * 4) = rb[j];
      }
    }
  };

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int kl = lane >> 5, l31 = lane & 31;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  const int nk = (g.Ktot + BK - 1) / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) load_tiles(kc + 1);
    const float* a = As + buf * BK * LDA + kl * LDA + wm * WTM + l31;
    const float* b = Bs + buf * BK * LDB + kl * LDB + wn * WTN + l31;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float av[TM], bv[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) av[t] = a[kk * 2 * LDA + t * 32];
#pragma unroll
      for (int u = 0; u < TN; ++u) bv[u] = b[kk * 2 * LDB + u * 32];
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) acc[t][u] = mfma32(av[t], bv[u], acc[t][u]);
    }
    if (kc + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D map of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int n = n0 + wn * WTN + u * 32 + l31;
    const bool nok = n < g.K;
    const float bval = (nok && (g.flags & SRHIP_EPI_BIAS)) ? bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
        if (nok && m < g.M) {
          float v = acc[t][u][r];
          if (g.flags & SRHIP_EPI_ROWSCALE) v *= rowscale[m];
          v += bval;
          if (g.flags & SRHIP_EPI_LRELU) v = v > 0.f ? v : v * g.slope;
          if (g.flags & SRHIP_EPI_RESIDUAL) v += residual[(size_t)m * g.ldr + n];
          float* o = dst + (size_t)m * g.ldy + n;
          if (g.accumulate) v += *o;
          *o = v;
        }
      }
    }
  }
}

// -------------------------------------------------------------------------------------------- //
// wgrad: rows = co, cols = kcol = (kh,kw,ci), reduction = pixels [p_begin, p_end) of this split.
template <int BM, int BN, int WM, int WN, bool VA, bool VB>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ dy,
                                                           float* __restrict__ partial, ConvGeom g,
                                                           int nsplit, int chunks_per_split) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int AVEC = BK * BM / 4, BVEC = BK * BN / 4;
  constexpr int AV = (AVEC + 255) / 256, BV = (BVEC + 255) / 256;
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * LDA + 2 * BK * LDB];
  float* As = lds;
  float* Bs = lds + 2 * BK * LDA;

  const int tid = threadIdx.x;
  const int ntn = (g.Ktot + BN - 1) / BN;
  const int ntm = (g.K + BM - 1) / BM;
  int bid = blockIdx.x;
  const int tile_n = bid % ntn;
  bid /= ntn;
  const int tile_m = bid % ntm;
  const int split = bid / ntm;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int HoWo = g.Ho * g.Wo;

  // loop-invariant decode of this thread's B columns (kcol -> tap, ci)
  int b_kh[BV][4], b_kw[BV][4], b_ci[BV][4];
#pragma unroll
  for (int j = 0; j < BV; ++j) {
    int idx = tid + 256 * j;
    int brow = idx / (BN / 4), bc = idx - brow * (BN / 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int kcol = n0 + bc * 4 + e;
      if (kcol < g.Ktot) {
        int tap = kcol / g.C;
        b_ci[j][e] = kcol - tap * g.C;
        b_kh[j][e] = tap / g.KW;
        b_kw[j][e] = tap - b_kh[j][e] * g.KW;
      } else {
        b_ci[j][e] = 0;
        b_kh[j][e] = -(1 << 28);
        b_kw[j][e] = 0;
      }
    }
  }

  float4 ra[AV], rb[BV];
  const int c_begin = split * chunks_per_split;
  const int nchunks_total = (g.M + BK - 1) / BK;
  const int c_end = min(c_begin + chunks_per_split, nchunks_total);

  auto load_tiles = [&](int kc) {
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      int idx = tid + 256 * j;
      int arow = idx / (BM / 4), ac = idx - arow * (BM / 4);
      int p = kc * BK + arow, co = m0 + ac * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((AVEC % 256 == 0 || idx < AVEC) && p < g.M) {
        const float* s = dy + (size_t)p * g.ldy + co;
        if (VA) {
          if (co < g.K) v = *reinterpret_cast<const float4*>(s);
        } else {
          if (co + 0 < g.K) v.x = s[0];
          if (co + 1 < g.K) v.y = s[1];
          if (co + 2 < g.K) v.z = s[2];
          if (co + 3 < g.K) v.w = s[3];
        }
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      int idx = tid + 256 * j;
      int brow = idx / (BN / 4);
      int p = kc * BK + brow;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((BVEC % 256 == 0 || idx < BVEC) && p < g.M) {
        int n = p / HoWo;
        int rem = p - n * HoWo;
        int ho = rem / g.Wo;
        int wo = rem - ho * g.Wo;
        int hbase = ho * g.so - g.pad, wbase = wo * g.so - g.pad;
        int pbase = n * g.H * g.W;
        if (VB) {
          int hi = hbase + b_kh[j][0], wi = wbase + b_kw[j][0];
          if (hi >= 0 && wi >= 0 && hi < g.H && wi < g.W)
            v = *reinterpret_cast<const float4*>(x + (size_t)(pbase + hi * g.W + wi) * g.ldx + b_ci[j][0]);
        } else {
          float e4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int hi = hbase + b_kh[j][e], wi = wbase + b_kw[j][e];
            e4[e] = (hi >= 0 && wi >= 0 && hi < g.H && wi < g.W)
                        ? x[(size_t)(pbase + hi * g.W + wi) * g.ldx + b_ci[j][e]]
                        : 0.f;
          }
          v = make_float4(e4[0], e4[1], e4[2], e4[3]);
        }
      }
      rb[j] = v;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      int idx = tid + 256 * j;
      if (AVEC % 256 == 0 || idx < AVEC) {
        int arow = idx / (BM / 4), ac = idx - arow * (BM / 4);
        *reinterpret_cast<float4*>(As + buf * BK * LDA + arow * LDA + ac * 4) = ra[j];
      }
    }
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      int idx = tid + 256 * j;
      if (BVEC % 256 == 0 || idx < BVEC) {
        int brow = idx / (BN / 4), bc = idx - brow * (BN / 4);
        *reinterpret_cast<float4*>(Bs + buf * BK * LDB + brow * LDB + bc * 4) = rb[j];
      }
    }
  };

  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int kl = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  if (c_begin < c_end) {
    load_tiles(c_begin);
    store_tiles(0);
    __syncthreads();
    for (int kc = c_begin; kc < c_end; ++kc) {
      const int buf = (kc - c_begin) & 1;
      if (kc + 1 < c_end) load_tiles(kc + 1);
      const float* a = As + buf * BK * LDA + kl * LDA + wm * WTM + l31;
      const float* b = Bs + buf * BK * LDB + kl * LDB + wn * WTN + l31;
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        float av[TM], bv[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) av[t] = a[kk * 2 * LDA + t * 32];
#pragma unroll
        for (int u = 0; u < TN; ++u) bv[u] = b[kk * 2 * LDB + u * 32];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32(av[t], bv[u], acc[t][u]);
      }
      if (kc + 1 < c_end) store_tiles(buf ^ 1);
      __syncthreads();
    }
  }
  float* out = partial + (size_t)split * g.K * g.Ktot;
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int n = n0 + wn * WTN + u * 32 + l31;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * WTM + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kl;
        if (n < g.Ktot && m < g.K) out[(size_t)m * g.Ktot + n] = acc[t][u][r];
      }
  }
}


// ---- forward of 3x3 stride-1 pad-1 convs with <= 3 input channels (discriminator / VGG / generator head) ---------- //
// K = 9*Cin <= 27 pads to 32 = sixteen 32x32x2 fp32 MFMA steps; the whole weight matrix (32 x 64) lives in registers.
// A block takes one image row: the three input rows (zero halo) are staged in LDS, a wave owns 32 consecutive output
// pixels x 64 channels, gathers its A values from LDS and stores 128-byte channel runs.  Write-bound by design.
__global__ __launch_bounds__(256) void headconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ packed,
                                                            const float* __restrict__ bias, float* __restrict__ y,
                                                            int N, int H, int W, int cin, int cout, int ldx, int ldy,
                                                            int ldw, float slope, int flags, int rows_per_block) {
  extern __shared__ float xs[];                      // [3][W + 2][cin]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int co0 = blockIdx.y * 64;
  const int ktot = 9 * cin;
  const int rowlen = (W + 2) * cin;
  int koff[16];
  float bw0[16], bw1[16];
#pragma unroll
  for (int s_ = 0; s_ < 16; ++s_) {
    const int k = 2 * s_ + half;
    koff[s_] = -1;
    bw0[s_] = bw1[s_] = 0.f;
    if (k < ktot) {
      const int tap = k / cin, ci = k - tap * cin;
      const int kh = tap / 3, kw = tap - kh * 3;
      koff[s_] = kh * rowlen + kw * cin + ci;
      if (co0 + l31 < cout) bw0[s_] = packed[(size_t)k * ldw + co0 + l31];
      if (co0 + 32 + l31 < cout) bw1[s_] = packed[(size_t)k * ldw + co0 + 32 + l31];
    }
  }
  float b0 = 0.f, b1 = 0.f;
  if (flags & SRHIP_EPI_BIAS) {
    if (co0 + l31 < cout) b0 = bias[co0 + l31];
    if (co0 + 32 + l31 < cout) b1 = bias[co0 + 32 + l31];
  }
  // a block walks `rows_per_block` consecutive image rows: the 32 weight loads per lane and the tap offsets above are paid once
  for (int r = blockIdx.x * rows_per_block; r < min((int)(blockIdx.x + 1) * rows_per_block, N * H); ++r) {
  const int n = r / H, oh = r - n * H;
  __syncthreads();                                   // the previous row's gathers are done
  for (int e = tid; e < 3 * rowlen; e += 256) {
    const int kh = e / rowlen, rem = e - kh * rowlen;
    const int col = rem / cin, ci = rem - col * cin;
    const int ih = oh + kh - 1, iw = col - 1;
    xs[e] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[((size_t)(n * H + ih) * W + iw) * ldx + ci] : 0.f;
  }
  __syncthreads();
  for (int ow0 = wave * 32; ow0 < W; ow0 += 128) {
    const int p = min(ow0 + l31, W - 1);             // lanes past the row end compute a duplicate, never stored
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) {
      const float a = koff[s_] >= 0 ? xs[p * cin + koff[s_]] : 0.f;
      acc0 = mfma32(a, bw0[s_], acc0);
      acc1 = mfma32(a, bw1[s_], acc1);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ow = ow0 + (i & 3) + 8 * (i >> 2) + 4 * half;
      if (ow < W) {
        float v0 = acc0[i] + b0, v1 = acc1[i] + b1;
        if (flags & SRHIP_EPI_LRELU) {
          v0 = v0 > 0.f ? v0 : v0 * slope;
          v1 = v1 > 0.f ? v1 : v1 * slope;
        }
        float* o = y + ((size_t)r * W + ow) * ldy + co0 + l31;
        if (co0 + l31 < cout) o[0] = v0;
        if (co0 + 32 + l31 < cout) o[32] = v1;
      }
    }
  }
  }
}

// ---- weight gradient of 3x3 stride-1 pad-1 convs with <= 3 input channels (discriminator / VGG head, 3 -> 64) ---- //
// dW[64][27] = dy^T [64 x pixels] . xcol [pixels x 27]: with the PIXELS as the contraction the exact-fp32 MFMA
// (32x32x2) fits well -- M = 64 channels = two tiles, N = 27 columns padded to 32 -- and costs ~40 us of matrix time
// for 1.5 M pixels; the generic kernel spent 1.1 ms on gather bookkeeping.  A block walks whole image rows: the
// three input rows (zero halo) are staged in LDS; per pixel pair a wave issues two coalesced dy loads (A operand:
// lane = channel, half-wave = pixel), one LDS gather (B operand: lane = column (tap, ci), half-wave = pixel) and two
// MFMAs.  The four waves take interleaved pixel pairs and are summed through LDS at the end.
// bias_partial != nullptr: the bias gradient (column sums of dy) comes out of the same pass -- every dy value is in a
// register here anyway; the separate colsum pass re-read the 382 MB of the 216 x 216 x 64 gradient (123 us, 7 per step).
// ymask != nullptr: dy is the gradient at the conv's ACTIVATED output y = LeakyReLU(conv): the activation's backward
// (dy * (y > 0 ? 1 : slope)) is applied to the values as they are loaded, so the separate lrelu-backward pass over the
// 382 MB gradient (read dy, read y, write g: 150 us in the discriminator's serial chain, three passes per step) is not
// needed when only the weight gradient of the layer is wanted.
__global__ __launch_bounds__(256) void wgrad_smallcin_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ ymask, float slope,
                                                              float* __restrict__ partial, float* __restrict__ bias_partial,
                                                              int N, int H, int W, int cin,
                                                              int cout, int ldx, int ldy, int rows_per_split) {
  extern __shared__ float xs[];                      // [3][W + 2][cin] (zero outside the image), then the reduce area
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int co0 = blockIdx.y * 64;
  const int ktot = 9 * cin;
  const int rowlen = (W + 2) * cin;
  int koff = -1;                                     // LDS offset of this lane's column for output pixel 0
  if (l31 < ktot) {
    const int tap = l31 / cin, ci = l31 - tap * cin;
    const int kh = tap / 3, kw = tap - kh * 3;
    koff = kh * rowlen + kw * cin + ci;
  }
  const bool c0ok = co0 + l31 < cout, c1ok = co0 + 32 + l31 < cout;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  float bs0 = 0.f, bs1 = 0.f;                        // this lane's share of the bias gradient of channels co0 + l31 (+ 32)
  const int r_begin = blockIdx.x * rows_per_split, r_end = min(r_begin + rows_per_split, N * H);
  for (int r = r_begin; r < r_end; ++r) {
    const int n = r / H, oh = r - n * H;
    __syncthreads();
    for (int e = tid; e < 3 * rowlen; e += 256) {
      const int kh = e / rowlen, rem = e - kh * rowlen;
      const int col = rem / cin, ci = rem - col * cin;
      const int ih = oh + kh - 1, iw = col - 1;
      xs[e] = (ih >= 0 && ih < H && iw >= 0 && iw < W) ? x[((size_t)(n * H + ih) * W + iw) * ldx + ci] : 0.f;
    }
    __syncthreads();
    const float* drow = dy + (size_t)r * W * ldy + co0 + l31;
    const float* yrow = ymask ? ymask + (size_t)r * W * ldy + co0 + l31 : nullptr;
#pragma unroll 4
    for (int ow = 2 * wave; ow < W; ow += 8) {       // this wave's pixel pairs (ow, ow + 1)
      const int p = ow + half;
      const bool pok = p < W;
      float d0 = (pok && c0ok) ? drow[(size_t)p * ldy] : 0.f;
      float d1 = (pok && c1ok) ? drow[(size_t)p * ldy + 32] : 0.f;
      if (yrow != nullptr) {
        if (pok && c0ok && !(yrow[(size_t)p * ldy] > 0.f)) d0 *= slope;
        if (pok && c1ok && !(yrow[(size_t)p * ldy + 32] > 0.f)) d1 *= slope;
      }
      const float b = (pok && koff >= 0) ? xs[p * cin + koff] : 0.f;
      bs0 += d0;
      bs1 += d1;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(d0, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(d1, b, acc1, 0, 0, 0);
    }
  }
  // sum the four waves' tiles through LDS, then one thread per (channel, column)
  __syncthreads();
  float* red = xs;                                   // [4][64][32]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    red[(wave * 64 + row) * 32 + l31] = acc0[r];
    red[(wave * 64 + 32 + row) * 32 + l31] = acc1[r];
  }
  __syncthreads();
  for (int e = tid; e < 64 * 32; e += 256) {
    const int co = e >> 5, k = e & 31;
    if (k < ktot && co0 + co < cout)
      partial[((size_t)blockIdx.x * cout + co0 + co) * ktot + k] =
          (red[e] + red[2048 + e]) + (red[4096 + e] + red[6144 + e]);
  }
  if (bias_partial != nullptr) {                     // 8 contributions per channel (4 waves x 2 pixel parities), fixed order
    __syncthreads();
    red[(wave * 2 + half) * 64 + l31] = bs0;
    red[(wave * 2 + half) * 64 + 32 + l31] = bs1;
    __syncthreads();
    if (tid < 64 && co0 + tid < cout) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += red[j * 64 + tid];
      bias_partial[(size_t)blockIdx.x * cout + co0 + tid] = s;
    }
  }
}

// ---- weight gradient of 3x3 stride-1 pad-1 convs with <= 4 OUTPUT channels at full image size (generator tail conv 64 -> 3
// at 216 x 216: 1.5 M pixels) ---- //
// dW[co][tap][ci] = sum over pixels of dy[p][co] * x[p + tap][ci]: 1728 outputs, each a reduction over every pixel, 5 GFLOP
// in all.  The generic kernel pads the 3 channels to a 32-wide MFMA tile and spends 1.1 ms on gather bookkeeping; an
// MFMA formulation wastes 10x the arithmetic on that padding.  Here a block of 9 x 64 threads owns one 64-channel slice:
// thread = (tap, ci) keeps its <= 4 sums in registers and walks 16 x 16 pixel patches; the 18 x 18 x 64 halo of a patch and
// its 256 dy vectors are staged in LDS (zero outside the image), so the inner step is one conflict-free LDS read of x,
// one broadcast read of dy and <= 4 FMAs.  One partial [co][tap][ci] tile per block, summed by wgrad_reduce_kernel.
template <int ND>
__global__ __launch_bounds__(576) void wgrad_narrow_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ partial, int N, int H, int W, int cin,
                                                            int ldx, int ldy, int tiles_h, int tiles_w) {
  // Round 5: thread = (row group rg, filter row kh, ci) instead of (tap, ci): it reads the 18 staged x values of one halo row ONCE and
  // uses them for the three kw taps (x[px + kw]), keeping 3 x ND sums; the three row groups split the 16 rows of a patch (6 / 5 / 5)
  // and are summed through LDS at the end in a fixed order.  LDS traffic per patch: 864 row reads + 768 broadcasts per 64 lanes
  // instead of 2304 + 2304 (the kernel was LDS-bound: 350 us for one pass over the 382 MB tensor).
  constexpr int PW = 16, PH = 16, PWP = PW + 2, NPX = (PH + 2) * PWP;   // 324 halo pixels
  constexpr int NLD = (NPX * 16 + 575) / 576;        // float4 loads per thread and patch (9)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ float xs[];                      // [324][64], then the patch's 256 dy vectors (float4, zero outside the image)
  float4* dys = reinterpret_cast<float4*>(xs + NPX * 64);
  const int tid = threadIdx.x;
  const int rg = tid / 192, kh = (tid - rg * 192) >> 6, ci = tid & 63;     // wave-uniform rg, kh
  const int row0 = rg == 0 ? 0 : (rg == 1 ? 6 : 11), row1 = rg == 0 ? 6 : (rg == 1 ? 11 : 16);
  const int ci0 = blockIdx.y * 64;
  const int npatch = N * tiles_h * tiles_w;
  const int tpi = tiles_h * tiles_w;
  f32x2 acc01[3], acc23[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) acc01[k] = acc23[k] = f32x2{0.f, 0.f};
  float4 stage[NLD];
  float4 dstage = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int pid) {                         // the halo of patch pid and its dy vectors: global -> registers
    const int n = pid / tpi, prem = pid - n * tpi;
    const int ty = prem / tiles_w, tx = prem - ty * tiles_w;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + i * 576;
      const int px = e >> 4, q = e & 15;
      const int pi = px / PWP, pj = px - pi * PWP;
      const int ih = ty * PH - 1 + pi, iw = tx * PW - 1 + pj;
      stage[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < NPX * 16 && ih >= 0 && ih < H && iw >= 0 && iw < W)
        stage[i] = *reinterpret_cast<const float4*>(x + ((size_t)(n * H + ih) * W + iw) * ldx + ci0 + q * 4);
    }
    dstage = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < PH * PW) {
      const int oh = ty * PH + (tid >> 4), ow = tx * PW + (tid & 15);
      if (oh < H && ow < W) {
        const float* dp = dy + ((size_t)(n * H + oh) * W + ow) * ldy;
        dstage.x = dp[0];
        if (ND > 1) dstage.y = dp[1];
        if (ND > 2) dstage.z = dp[2];
        if (ND > 3) dstage.w = dp[3];
      }
    }
  };
  int pid = blockIdx.x;
  if (pid < npatch) fetch(pid);
  for (; pid < npatch; pid += gridDim.x) {
    __syncthreads();                                  // the previous patch is consumed
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int e = tid + i * 576;
      if (e < NPX * 16) *reinterpret_cast<float4*>(xs + (e >> 4) * 64 + (e & 15) * 4) = stage[i];
    }
    if (tid < PH * PW) dys[tid] = dstage;
    __syncthreads();
    if (pid + (int)gridDim.x < npatch) fetch(pid + gridDim.x);   // next patch's loads fly under this patch's arithmetic
    for (int py = row0; py < row1; ++py) {
      const float* xr = xs + ((py + kh) * PWP) * 64 + ci;        // halo row py + kh: x[py + kh - 1][-1 .. 16]
      float xv[PWP];
#pragma unroll
      for (int j = 0; j < PWP; ++j) xv[j] = xr[j * 64];
#pragma unroll
      for (int px = 0; px < PW; ++px) {
        const float4 d = dys[py * PW + px];           // uniform address: one broadcast read
        const f32x2 d01 = {d.x, d.y}, d23 = {d.z, d.w};
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const f32x2 vv = {xv[px + kw], xv[px + kw]};
          acc01[kw] = __builtin_elementwise_fma(vv, d01, acc01[kw]);
          if (ND > 2) acc23[kw] = __builtin_elementwise_fma(vv, d23, acc23[kw]);
        }
      }
    }
  }
  // the three row groups' sums: red[rg][co][tap][ci] in the (consumed) halo area, then one thread per (co, tap, ci)
  __syncthreads();
  float* red = xs;
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    const int tap = kh * 3 + kw;
    red[((rg * 4 + 0) * 9 + tap) * 64 + ci] = acc01[kw].x;
    if (ND > 1) red[((rg * 4 + 1) * 9 + tap) * 64 + ci] = acc01[kw].y;
    if (ND > 2) red[((rg * 4 + 2) * 9 + tap) * 64 + ci] = acc23[kw].x;
    if (ND > 3) red[((rg * 4 + 3) * 9 + tap) * 64 + ci] = acc23[kw].y;
  }
  __syncthreads();
  const int ktot = 9 * cin;
  for (int e = tid; e < ND * 9 * 64; e += 576) {
    const int c = e & 63, tap = (e >> 6) % 9, co = e / (9 * 64);
    const int o = (co * 9 + tap) * 64 + c;
    const float v = (red[o] + red[4 * 9 * 64 + o]) + red[2 * 4 * 9 * 64 + o];
    partial[(size_t)blockIdx.x * ND * ktot + (size_t)co * ktot + tap * cin + ci0 + c] = v;
  }
}

// partial[s][co][(kh,kw,ci)] --sum over s--> dw[co][ci][kh][kw]
// 16 outputs per block x 16 split lanes (the small-channel convs have few outputs and ~1000 splits: one thread per
// output walked them serially, 235 us for 1728 outputs)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw,
                                                            int nsplit, int cout, int cin, int khkw, int ktot) {
  __shared__ float red[256];
  const int o = threadIdx.x & 15, sub = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + o;
  const int total = cout * ktot;
  float s0 = 0.f, s1 = 0.f;
  if (idx < total) {
    int i = sub;
    for (; i + 16 < nsplit; i += 32) {
      s0 += partial[(size_t)i * total + idx];
      s1 += partial[(size_t)(i + 16) * total + idx];
    }
    for (; i < nsplit; i += 16) s0 += partial[(size_t)i * total + idx];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (sub == 0 && idx < total) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j * 16 + o];
    const int co = idx / ktot, kcol = idx - co * ktot;
    const int tap = kcol / cin, ci = kcol - tap * cin;
    dw[((size_t)co * cin + ci) * khkw + tap] = s;
  }
}

// OIHW -> packed GEMM operand (see header)
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ packed, int cout, int cin,
                                   int kh, int kw, int mode, int ld) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int csrc = mode == 0 ? cin : cout;
  int cdst = mode == 0 ? cout : cin;
  int rows = kh * kw * csrc;
  if (idx >= rows * ld) return;
  int row = idx / ld, col = idx - row * ld;
  float v = 0.f;
  if (col < cdst) {
    int tap = row / csrc, cs = row - tap * csrc;
    int a = tap / kw, b = tap - a * kw;
    if (mode == 0)
      v = w[(((size_t)col * cin + cs) * kh + a) * kw + b];
    else
      v = w[(((size_t)cs * cin + col) * kh + (kh - 1 - a)) * kw + (kw - 1 - b)];
  }
  packed[idx] = v;
}

// -------------------------------------------------------------------------------------------- //
template <int BM, int BN, int WM, int WN>
static int launch_fprop(const float* src, const float* wt, const float* bias, const float* residual,
                        const float* rowscale, float* dst, const ConvGeom& g, bool vec, hipStream_t st) {
  int blocks = cdiv(g.M, BM) * cdiv(g.K, BN);
  if (vec)
    hipLaunchKernelGGL((igemm_fprop_kernel<BM, BN, WM, WN, true>), dim3(blocks), dim3(256), 0, st, src, wt, bias,
                       residual, rowscale, dst, g);
  else
    hipLaunchKernelGGL((igemm_fprop_kernel<BM, BN, WM, WN, false>), dim3(blocks), dim3(256), 0, st, src, wt, bias,
                       residual, rowscale, dst, g);
  return check_launch("igemm_fprop");
}

static int run_fprop(const float* src, const float* wt, const float* bias, const float* residual,
                     const float* rowscale, float* dst, const ConvGeom& g, hipStream_t st) {
  const bool vec = (g.C % 4 == 0) && (g.ldx % 4 == 0);
  if (g.M <= 0) return SRHIP_OK;
  if (g.K <= 32) return launch_fprop<128, 32, 4, 1>(src, wt, bias, residual, rowscale, dst, g, vec, st);
  // keep >= ~2 waves of blocks over the 256 CUs; shrink the tile when the problem is small
  long blocks128 = (long)cdiv(g.M, 128) * cdiv(g.K, 128);
  if (g.K >= 128 && blocks128 >= 512) return launch_fprop<128, 128, 2, 2>(src, wt, bias, residual, rowscale, dst, g, vec, st);
  long blocks64n = (long)cdiv(g.M, 128) * cdiv(g.K, 64);
  if (blocks64n >= 512) return launch_fprop<128, 64, 2, 2>(src, wt, bias, residual, rowscale, dst, g, vec, st);
  return launch_fprop<64, 64, 2, 2>(src, wt, bias, residual, rowscale, dst, g, vec, st);
}

template <int BM, int BN, int WM, int WN>
static int launch_wgrad(const float* x, const float* dy, float* partial, const ConvGeom& g, int nsplit, int cps,
                        bool va, bool vb, hipStream_t st) {
  int blocks = cdiv(g.K, BM) * cdiv(g.Ktot, BN) * nsplit;
#define SRHIP_WG(VA_, VB_)                                                                                     \
  hipLaunchKernelGGL((igemm_wgrad_kernel<BM, BN, WM, WN, VA_, VB_>), dim3(blocks), dim3(256), 0, st, x, dy, \
                     partial, g, nsplit, cps)
  if (va && vb)
    SRHIP_WG(true, true);
  else if (va)
    SRHIP_WG(true, false);
  else if (vb)
    SRHIP_WG(false, true);
  else
    SRHIP_WG(false, false);
#undef SRHIP_WG
  return check_launch("igemm_wgrad");
}

struct WgradPlan {
  int bm, bn, nsplit, chunks_per_split;
};

static WgradPlan plan_wgrad(int M, int cout, int ktot) {
  WgradPlan p;
  p.bm = cout > 64 ? 128 : (cout > 32 ? 64 : 32);
  p.bn = p.bm == 32 ? 128 : 64;
  long tiles = (long)cdiv(cout, p.bm) * cdiv(ktot, p.bn);
  int nchunks = cdiv(M, BK);
  long want = (1024 + tiles - 1) / tiles;          // ~4 blocks per CU overall
  if (cout <= 4 && want < 256) want = 256;         // wgrad_narrow_kernel: one 576-thread block per CU, one partial tile per block
  long maxsplit = (nchunks + 15) / 16;             // at least 16 chunks (256 pixels) per split
  long ns = want < 1 ? 1 : want;
  if (ns > maxsplit) ns = maxsplit;
  if (ns < 1) ns = 1;
  p.chunks_per_split = (int)((nchunks + ns - 1) / ns);
  p.nsplit = cdiv(nchunks, p.chunks_per_split);
  return p;
}

}  // namespace srhip

namespace srhip {

int legacy_packed_ld(int cdst) { return ((cdst + 31) / 32) * 32; }

int legacy_pack_weight(const float* w, float* packed, int cout, int cin, int kh, int kw, int mode, void* stream) {
  SRHIP_REQUIRE(w && packed && cout > 0 && cin > 0 && kh > 0 && kw > 0 && (mode == 0 || mode == 1),
                "pack_weight: bad argument");
  int csrc = mode == 0 ? cin : cout, cdst = mode == 0 ? cout : cin;
  int ld = legacy_packed_ld(cdst);
  long total = (long)kh * kw * csrc * ld;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), w, packed, cout,
                     cin, kh, kw, mode, ld);
  return check_launch("pack_weight");
}

int legacy_conv2d_fwd(const float* x, const float* packed, const float* bias, const float* residual,
                     const float* rowscale, float* y, int n, int h, int w, int cin, int cout, int kh, int kw,
                     int stride, int pad, int ldx, int ldy, int ldr, float slope, int flags, void* stream) {
  SRHIP_REQUIRE(x && packed && y, "conv2d_fwd: null tensor");
  SRHIP_REQUIRE(n >= 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                "conv2d_fwd: bad geometry");
  SRHIP_REQUIRE(ldx >= cin && ldy >= cout, "conv2d_fwd: row stride smaller than channel count");
  SRHIP_REQUIRE(!(flags & SRHIP_EPI_BIAS) || bias, "conv2d_fwd: EPI_BIAS without bias");
  SRHIP_REQUIRE(!(flags & SRHIP_EPI_RESIDUAL) || (residual && ldr >= cout), "conv2d_fwd: EPI_RESIDUAL without residual");
  SRHIP_REQUIRE(!(flags & SRHIP_EPI_ROWSCALE) || rowscale, "conv2d_fwd: EPI_ROWSCALE without rowscale");
  ConvGeom g;
  g.N = n; g.H = h; g.W = w; g.C = cin;
  g.Ho = (h + 2 * pad - kh) / stride + 1;
  g.Wo = (w + 2 * pad - kw) / stride + 1;
  SRHIP_REQUIRE(g.Ho > 0 && g.Wo > 0, "conv2d_fwd: empty output");
  g.K = cout; g.KH = kh; g.KW = kw; g.so = stride; g.pad = pad; g.dv = 1;
  g.ldx = ldx; g.ldy = ldy; g.ldr = ldr; g.ldw = legacy_packed_ld(cout);
  g.Ktot = kh * kw * cin;
  long M = (long)n * g.Ho * g.Wo;
  SRHIP_REQUIRE(M < (1L << 31) && (long)n * h * w < (1L << 31), "conv2d_fwd: pixel count overflows int32");
  g.M = (int)M; g.slope = slope; g.flags = flags; g.accumulate = 0;
  if (cin <= 3 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && M >= 65536 &&
      !(flags & ~(SRHIP_EPI_BIAS | SRHIP_EPI_LRELU)) && (size_t)3 * (w + 2) * cin * sizeof(float) <= 32 * 1024) {
    const int rpb = g_headconv_rows > 0 ? g_headconv_rows : 1;
    hipLaunchKernelGGL(headconv_fwd_kernel, dim3(cdiv((long)n * h, rpb), cdiv(cout, 64)), dim3(256), (size_t)3 * (w + 2) * cin * sizeof(float),
                       as_stream(stream), x, packed, bias, y, n, h, w, cin, cout, ldx, ldy, g.ldw, slope, flags, rpb);
    return check_launch("headconv_fwd");
  }
  return run_fprop(x, packed, bias, residual, rowscale, y, g, as_stream(stream));
}

int legacy_conv2d_dgrad(const float* dy, const float* packed, float* dx, int n, int h, int w, int cin, int cout,
                       int kh, int kw, int stride, int pad, int ldy, int ldx, int accumulate, void* stream) {
  SRHIP_REQUIRE(dy && packed && dx, "conv2d_dgrad: null tensor");
  SRHIP_REQUIRE(n >= 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                "conv2d_dgrad: bad geometry");
  SRHIP_REQUIRE(pad <= kh - 1 && pad <= kw - 1, "conv2d_dgrad: pad > kernel-1 unsupported");
  SRHIP_REQUIRE(ldy >= cout && ldx >= cin, "conv2d_dgrad: row stride smaller than channel count");
  ConvGeom g;
  // source = dy (N x Ho x Wo x Cout), destination = dx (N x H x W x Cin)
  g.N = n;
  g.H = (h + 2 * pad - kh) / stride + 1;
  g.W = (w + 2 * pad - kw) / stride + 1;
  SRHIP_REQUIRE(g.H > 0 && g.W > 0, "conv2d_dgrad: empty dy");
  g.C = cout; g.Ho = h; g.Wo = w; g.K = cin; g.KH = kh; g.KW = kw;
  g.so = 1; g.pad = kh - 1 - pad; g.dv = stride;
  SRHIP_REQUIRE(kh == kw, "conv2d_dgrad: square kernels only");
  g.ldx = ldy; g.ldy = ldx; g.ldr = 0; g.ldw = legacy_packed_ld(cin);
  g.Ktot = kh * kw * cout;
  long M = (long)n * h * w;
  SRHIP_REQUIRE(M < (1L << 31), "conv2d_dgrad: pixel count overflows int32");
  g.M = (int)M; g.slope = 0.f; g.flags = 0; g.accumulate = accumulate ? 1 : 0;
  return run_fprop(dy, packed, nullptr, nullptr, nullptr, dx, g, as_stream(stream));
}

size_t legacy_conv2d_wgrad_workspace(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
  long M = (long)n * ho * wo;
  if (M <= 0) return 0;
  WgradPlan p = plan_wgrad((int)M, cout, kh * kw * cin);
  return (size_t)p.nsplit * cout * kh * kw * cin * sizeof(float);
}

size_t legacy_conv2d_wgrad_bias_workspace(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
  const long M = (long)n * ho * wo;
  if (M <= 0) return 0;
  return (size_t)plan_wgrad((int)M, cout, kh * kw * cin).nsplit * cout * sizeof(float);
}

// db / bias_ws / bias_done: kernels that see every dy value anyway also produce the bias gradient (bias_ws >=
// legacy_conv2d_wgrad_bias_workspace bytes) and set *bias_done; otherwise the caller runs the column-sum pass.
int legacy_conv2d_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int n,
                       int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int ldx, int ldy,
                       void* stream, float* db, float* bias_ws, int* bias_done, const float* ymask, float slope) {
  if (bias_done) *bias_done = 0;
  SRHIP_REQUIRE(!ymask || (db && bias_ws && bias_done && cin <= 3 && kh == 3 && kw == 3 && stride == 1 && pad == 1 &&
                           (long)n * h * w >= 65536 && (size_t)3 * (w + 2) * cin * sizeof(float) <= 32 * 1024),
                "conv2d_wgrad: the fused activation mask is built for the 3-channel 3x3 head convs at full image size (with bias)");
  SRHIP_REQUIRE(x && dy && dw, "conv2d_wgrad: null tensor");
  SRHIP_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0,
                "conv2d_wgrad: bad geometry");
  SRHIP_REQUIRE(ldx >= cin && ldy >= cout, "conv2d_wgrad: row stride smaller than channel count");
  ConvGeom g;
  g.N = n; g.H = h; g.W = w; g.C = cin;
  g.Ho = (h + 2 * pad - kh) / stride + 1;
  g.Wo = (w + 2 * pad - kw) / stride + 1;
  SRHIP_REQUIRE(g.Ho > 0 && g.Wo > 0, "conv2d_wgrad: empty output");
  g.K = cout; g.KH = kh; g.KW = kw; g.so = stride; g.pad = pad; g.dv = 1;
  g.ldx = ldx; g.ldy = ldy; g.ldr = 0; g.ldw = 0; g.Ktot = kh * kw * cin;
  long M = (long)n * g.Ho * g.Wo;
  SRHIP_REQUIRE(M < (1L << 31) && (long)n * h * w < (1L << 31), "conv2d_wgrad: pixel count overflows int32");
  g.M = (int)M; g.slope = 0.f; g.flags = 0; g.accumulate = 0;
  WgradPlan p = plan_wgrad(g.M, cout, g.Ktot);
  size_t need = (size_t)p.nsplit * cout * g.Ktot * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    set_error("conv2d_wgrad: workspace %zu bytes < required %zu", workspace_bytes, need);
    return SRHIP_ERR_WORKSPACE;
  }
  float* partial = static_cast<float*>(workspace);
  hipStream_t st = as_stream(stream);
  if (cin <= 3 && kh == 3 && kw == 3 && stride == 1 && pad == 1 && g.M >= 65536 &&
      (size_t)3 * (w + 2) * cin * sizeof(float) <= 32 * 1024) {
    const int rows = n * h;
    int ns = p.nsplit < rows ? p.nsplit : rows;                       // splits the workspace was sized for
    const int rps = cdiv(rows, ns);
    ns = cdiv(rows, rps);
    const bool fuse_bias = db != nullptr && bias_ws != nullptr && bias_done != nullptr;
    hipLaunchKernelGGL(wgrad_smallcin_kernel, dim3(ns, cdiv(cout, 64)), dim3(256), (size_t)32 * 1024, st,
                       x, dy, ymask, slope, partial, fuse_bias ? bias_ws : nullptr, n, h, w, cin, cout, ldx, ldy, rps);
    int rc0 = check_launch("wgrad_smallcin");
    if (rc0) return rc0;
    long total0 = (long)cout * g.Ktot;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total0, 16)), dim3(256), 0, st, partial, dw, ns, cout, cin, kh * kw,
                       g.Ktot);
    if (fuse_bias) {                                  // bias_ws[split][cout] -> db[cout]: the same reduce with a 1-column matrix
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(cout, 16)), dim3(256), 0, st, bias_ws, db, ns, cout, 1, 1, 1);
      *bias_done = 1;
    }
    return check_launch("wgrad_reduce");
  }
  if (cout <= 4 && cin % 64 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0 && kh == 3 && kw == 3 && stride == 1 && pad == 1 &&
      g.M >= 65536 && p.nsplit >= 64) {
    const int th = cdiv(h, 16), tw = cdiv(w, 16);
    const int ns = p.nsplit < 256 ? p.nsplit : 256;                   // one 576-thread block per CU; the workspace holds p.nsplit tiles
    const size_t lds_bytes = (size_t)(18 * 18 * 64 + 256 * 4) * sizeof(float);
#define SRHIP_WN(ND_)                                                                                                    \
  do {                                                                                                                   \
    static bool attr_set = false;                                                                                        \
    if (!attr_set) {                                                                                                     \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_narrow_kernel<ND_>),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                             \
      attr_set = true;                                                                                                   \
    }                                                                                                                    \
    hipLaunchKernelGGL(wgrad_narrow_kernel<ND_>, dim3(ns, cin / 64), dim3(576), lds_bytes, st, x, dy, partial, n, h, w,  \
                       cin, ldx, ldy, th, tw);                                                                           \
  } while (0)
    if (cout == 1) SRHIP_WN(1);
    else if (cout == 2) SRHIP_WN(2);
    else if (cout == 3) SRHIP_WN(3);
    else SRHIP_WN(4);
#undef SRHIP_WN
    int rc1 = check_launch("wgrad_narrow");
    if (rc1) return rc1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((long)cout * g.Ktot, 16)), dim3(256), 0, st, partial, dw, ns, cout, cin,
                       kh * kw, g.Ktot);
    return check_launch("wgrad_reduce");
  }
  const bool va = (cout % 4 == 0) && (ldy % 4 == 0);
  const bool vb = (cin % 4 == 0) && (ldx % 4 == 0);
  int rc;
  if (p.bm == 128)
    rc = launch_wgrad<128, 64, 2, 2>(x, dy, partial, g, p.nsplit, p.chunks_per_split, va, vb, st);
  else if (p.bm == 64)
    rc = launch_wgrad<64, 64, 2, 2>(x, dy, partial, g, p.nsplit, p.chunks_per_split, va, vb, st);
  else
    rc = launch_wgrad<32, 128, 1, 4>(x, dy, partial, g, p.nsplit, p.chunks_per_split, va, vb, st);
  if (rc) return rc;
  long total = (long)cout * g.Ktot;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv(total, 16)), dim3(256), 0, st, partial, dw, p.nsplit, cout, cin,
                     kh * kw, g.Ktot);
  return check_launch("wgrad_reduce");
}

}  // namespace srhip
