// Fast implicit-GEMM convolution kernels for gfx950 (source channels % 16 == 0, <= 32 taps): the
// shapes that carry >95 % of the SRADSGAN step's FLOPs (RAB 3x3 64<->256, 1x1 tails, upsampler,
// discriminator and VGG 3x3 convs).  Exact-fp32 matrix pipe (v_mfma_f32_32x32x2_f32).
//
// What makes them fast compared with the generic kernel (conv_igemm.hip):
//   * both GEMM operands are "row = M/N index, K contiguous" in global memory (NHWC activations,
//     n-major packed weights), so tiles go global -> VGPR -> LDS as 16-byte vectors with NO
//     transpose, and MFMA fragments come back as ds_read_b128: the K order inside an 8-wide group is
//     permuted identically for A and B (lanes 0-31 take k 0..3, lanes 32-63 take k 4..7), which a
//     contraction does not care about.  LDS rows are 80 bytes apart => conflict-free b128 reads;
//   * a K chunk of 16 never straddles a filter tap, so the im2col address of a chunk is
//     "per-thread pixel base + one scalar tap offset"; padding/stride holes are handled by the
//     buffer-load bounds check (offset >= num_records returns 0): no branches, ~3 VALU per load;
//   * strided backward-data is decomposed into stride^2 phase classes, each a dense GEMM over only
//     the taps that hit it (a 3x3 stride-2 dgrad does 9 tap-GEMMs instead of 36);
//   * XCD-aware tile order: each of the 8 XCDs walks a contiguous range of output tiles, so the
//     halo rows and the weights of neighbouring tiles are served by that XCD's L2;
//   * wgrad also emits the bias gradient (column sums of dy) from the tiles it already stages.
#include "conv_dev.h"

namespace srhip {

int g_fast_ablate = 0;   // srhip_debug_set(3, bits): 0x100 = no in-loop loads, 0x200 = no barrier (timing only, wrong results); 0x400 = plain epilogue stores (correct results)


// ================================================================================================ //
// fprop / dgrad
// ================================================================================================ //

// MATH >= 1 (BK 16 only): 16-bit products (PROD = MATH - 1), B read from the pre-split / fp16 section of the packed weights
template <int BM, int BN, int WM, int WN, int BK, int MATH = 0>
__global__ __launch_bounds__(WM* WN * 64) void fast_conv_kernel(const float* __restrict__ src,
                                                                 const float* __restrict__ wt,
                                                                 const float* __restrict__ bias,
                                                                 const float* __restrict__ residual,
                                                                 const float* __restrict__ rowscale,
                                                                 const float* __restrict__ chanscale,
                                                                 const float* __restrict__ actmask,
                                                                 float* __restrict__ dst, FastGeom g, int nblk_m,
                                                                 int nblk_n) {
  constexpr int NT = WM * WN * 64;                 // threads
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LS = BK + 4;                       // LDS row stride (floats): (LS/4) odd => conflict-free b128 reads
  constexpr int QPR = BK / 4;                      // 16-byte quads per row
  constexpr int RPP = NT / QPR;                    // rows covered by one pass of all threads
  constexpr int AR = (BM + RPP - 1) / RPP, BR = (BN + RPP - 1) / RPP;
  constexpr int STAGE = (BM + BN) * LS;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int tile = xcd_tile(blockIdx.x, nblk_m * nblk_n);
  const int tile_n = tile % nblk_n, tile_m = tile / nblk_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int lrow = tid / QPR, kq = tid % QPR;

  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, g.src_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, g.w_bytes, 0x00020000);

  // ---- per-thread row bookkeeping, fixed for the whole K loop ----
  int abase[AR], aimg[AR];
  unsigned amask[AR];
  const int OHOW = g.OH * g.OW;
  const bool cscale = (g.flags & SRHIP_EPI_CHANSCALE) != 0;      // A[m][k] *= chanscale[image(m)][c(k)]
#pragma unroll
  for (int i = 0; i < AR; ++i) {
    const int m = m0 + lrow + RPP * i;
    abase[i] = 0;
    amask[i] = 0;
    aimg[i] = 0;
    if (m < g.M && (BM % RPP == 0 || lrow + RPP * i < BM)) {
      const int n = m / OHOW;
      aimg[i] = n;
      const int rem = m - n * OHOW;
      const int oh = rem / g.OW;
      const int ow = rem - oh * g.OW;
      const int sh0 = oh * g.ss, sw0 = ow * g.ss;
      abase[i] = ((n * g.Hs + sh0) * g.Ws + sw0) * g.lds + kq * 4;
      unsigned mk = 0;
      for (int th = 0; th < g.TH; ++th) {
        const int sh = sh0 + g.dh0 + th * g.dhs;
        for (int tw = 0; tw < g.TW; ++tw) {
          const int sw = sw0 + g.dw0 + tw * g.dws;
          if (sh >= 0 && sh < g.Hs && sw >= 0 && sw < g.Ws) mk |= 1u << (th * g.TW + tw);
        }
      }
      amask[i] = mk;
    }
  }
  int bbase[BR];
  bool bval[BR];
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int n = n0 + lrow + RPP * j;
    bval[j] = (BN % RPP == 0 || lrow + RPP * j < BN) && n < g.K;
    bbase[j] = n * g.ldw + kq * 4;
  }

  // ---- K-loop state (wave-uniform): tap (th,tw), channel chunk cc ----
  const int CC = g.C / BK;
  const int nk = g.TH * g.TW * CC;
  int th = 0, tw = 0, cc = 0;
  float4 ra[AR], rb[BR], rsc[AR];

  auto load_tiles = [&]() {
    const int tapoff = ((g.dh0 + th * g.dhs) * g.Ws + (g.dw0 + tw * g.dws)) * g.lds + cc * BK;
    if (cscale) {
#pragma unroll
      for (int i = 0; i < AR; ++i)
        rsc[i] = *reinterpret_cast<const float4*>(chanscale + (size_t)aimg[i] * g.C + cc * BK + kq * 4);
    }
    const int wk = ((g.kh0 + th * g.khs) * g.KW + (g.kw0 + tw * g.kws)) * g.C + cc * BK;
    const int bit = th * g.TW + tw;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
      const unsigned off = ((amask[i] >> bit) & 1u) ? (unsigned)(abase[i] + tapoff) * 4u : F_OOB;
      ra[i] = bufload4(rs, off);
    }
#pragma unroll
    for (int j = 0; j < BR; ++j) {
      const unsigned off = bval[j] ? (unsigned)(bbase[j] + wk) * 4u : F_OOB;
      rb[j] = bufload4(rw, off);
    }
    // taps innermost: the 9 taps of one 16-channel chunk re-read the same 64-byte segments of ~3 image
    // rows back to back (L1/L2 hits); with taps outermost a 256-channel input was re-fetched 9x from
    // beyond L2 (FETCH_SIZE 788 MB vs 96 MB algorithmic, profiles/r01_conv_pmc_summary.txt)
    if (++tw == g.TW) {
      tw = 0;
      if (++th == g.TH) {
        th = 0;
        ++cc;
      }
    }
  };
  auto store_tiles = [&](int stage) {
    float* a = lds + stage * STAGE + lrow * LS + kq * 4;
    if (cscale) {
#pragma unroll
      for (int i = 0; i < AR; ++i) {
        ra[i].x *= rsc[i].x;
        ra[i].y *= rsc[i].y;
        ra[i].z *= rsc[i].z;
        ra[i].w *= rsc[i].w;
      }
    }
#pragma unroll
    for (int i = 0; i < AR; ++i)
      if (BM % RPP == 0 || lrow + RPP * i < BM) *reinterpret_cast<float4*>(a + RPP * i * LS) = ra[i];
    float* b = lds + stage * STAGE + BM * LS + lrow * LS + kq * 4;
#pragma unroll
    for (int j = 0; j < BR; ++j)
      if (BN % RPP == 0 || lrow + RPP * j < BN) *reinterpret_cast<float4*>(b + RPP * j * LS) = rb[j];
  };

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int khalf = lane >> 5, l31 = lane & 31;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  if (nk > 0) {
    load_tiles();
    store_tiles(0);
    __syncthreads();
    const bool abl_noload = (g.flags & 0x100) != 0, abl_nobar = (g.flags & 0x200) != 0;   // timing ablations only
    for (int kc = 0; kc < nk; ++kc) {
      const int stage = abl_noload ? 0 : (kc & 1);
      if (kc + 1 < nk && !abl_noload) load_tiles();
      if (MATH >= 1) {
        constexpr int PROD = MATH >= 1 ? MATH - 1 : 0;
        const float* a = lds + stage * STAGE + (wm * WTM + l31) * LS + khalf * 8;
        const float* b = lds + stage * STAGE + BM * LS + (wn * WTN + l31) * LS + khalf * 8;
        bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const float4 a0 = *reinterpret_cast<const float4*>(a + t * 32 * LS), a1 = *reinterpret_cast<const float4*>(a + t * 32 * LS + 4);
          if (PROD == 0) split_bf16x8(a0, a1, ah[t], al[t]);
          else ah[t] = al[t] = round16x8<PROD>(a0, a1);
        }
#pragma unroll
        for (int u = 0; u < TN; ++u) {
          bh[u] = *reinterpret_cast<const bf16x8_t*>(b + u * 32 * LS);
          bl[u] = PROD == 0 ? *reinterpret_cast<const bf16x8_t*>(b + u * 32 * LS + 4) : bh[u];
        }
#pragma unroll
        for (int i = 0; i < nprod<PROD>() * TM * TN; ++i) {
          const int grp = PROD == 0 ? i / (TM * TN) : 2, t = (i % (TM * TN)) / TN, u = i % TN;
          acc[t][u] = mma16<PROD>(grp == 0 ? al[t] : ah[t], grp == 1 ? bl[u] : bh[u], acc[t][u]);
        }
        if (kc + 1 < nk && !abl_noload) store_tiles(stage ^ 1);
        if (!abl_nobar) __syncthreads();
        continue;
      }
      const float* a = lds + stage * STAGE + (wm * WTM + l31) * LS + khalf * 4;
      const float* b = lds + stage * STAGE + BM * LS + (wn * WTN + l31) * LS + khalf * 4;
      float4 af[2][TM], bf[2][TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) af[0][t] = *reinterpret_cast<const float4*>(a + t * 32 * LS);
#pragma unroll
      for (int u = 0; u < TN; ++u) bf[0][u] = *reinterpret_cast<const float4*>(b + u * 32 * LS);
#pragma unroll
      for (int ks = 0; ks < BK / 8; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks + 1 < BK / 8) {
#pragma unroll
          for (int t = 0; t < TM; ++t) af[nxt][t] = *reinterpret_cast<const float4*>(a + t * 32 * LS + (ks + 1) * 8);
#pragma unroll
          for (int u = 0; u < TN; ++u) bf[nxt][u] = *reinterpret_cast<const float4*>(b + u * 32 * LS + (ks + 1) * 8);
        }
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(af[cur][t].x, bf[cur][u].x, acc[t][u]);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(af[cur][t].y, bf[cur][u].y, acc[t][u]);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(af[cur][t].z, bf[cur][u].z, acc[t][u]);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(af[cur][t].w, bf[cur][u].w, acc[t][u]);
      }
      if (kc + 1 < nk && !abl_noload) store_tiles(stage ^ 1);
      if (!abl_nobar) __syncthreads();
    }
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
#pragma unroll
  for (int t = 0; t < TM; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * WTM + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
      if (m >= g.M) continue;
      size_t dpix = (size_t)m;
      if (!g.dst_identity) {
        const int n = m / OHOW;
        const int rem = m - n * OHOW;
        const int oh = rem / g.OW;
        const int ow = rem - oh * g.OW;
        dpix = ((size_t)n * g.Hd + (oh * g.dsd + g.ph)) * g.Wd + (ow * g.dsd + g.pw);
      }
      const float rsc = (g.flags & SRHIP_EPI_ROWSCALE) ? rowscale[dpix] : 1.f;
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        const int n = n0 + wn * WTN + u * 32 + l31;
        if (n >= g.K) continue;
        float v = acc[t][u][r];
        if (g.flags & SRHIP_EPI_ROWSCALE) v *= rsc;
        if (g.flags & SRHIP_EPI_BIAS) v += bias[n];
        if (g.flags & SRHIP_EPI_LRELU) v = v > 0.f ? v : v * g.slope;
        if (g.flags & SRHIP_EPI_ACTMASK) v = actmask[dpix * g.ldd + n] > 0.f ? v : v * g.slope;
        if (g.flags & SRHIP_EPI_RESIDUAL) v += residual[dpix * g.ldr + n];
        float* o = dst + dpix * g.ldd + n;
        if (g.accumulate) v += *o;
        *o = v;
      }
    }
  }
}

// ================================================================================================ //
// fprop / dgrad, LDS-DMA variant: tiles go global -> LDS with global_load_lds_dwordx4 (no VGPR
// staging, no ds_write), 3-stage ring, prefetch distance 2, ONE raw s_barrier per K chunk and a
// counted s_waitcnt vmcnt (the loads of the next chunk stay in flight across the barrier).
//   * the DMA writes lane-linear (wave base + lane*16 B), so an LDS row is 64 B unpadded and the
//     bank spread comes from an XOR swizzle applied on the SOURCE side: 16-byte slot (row, s) holds
//     global quad q = s ^ ((row>>2)&3); fragment reads apply the same involution => conflict-free
//     ds_read_b128 (every 16-lane service group covers all 16 slots of a 256 B bank row once);
//   * padding / stride holes: lanes whose tap falls outside the image DMA from a 16-byte zero block;
//   * the DMA and its wait are inline asm (hipcc would otherwise drain vmcnt(0) before every ds_read
//     that may alias an in-flight LDS-DMA); the fragment reads stay ordinary loads and are ordered
//     behind the asm wait + barrier by their "memory" clobbers;
//   * EPI >= 0 fixes the epilogue flags at compile time (no per-element branches); EPI < 0 = dynamic.
// ================================================================================================ //
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};


// epilogue of one float4 of output (4 consecutive channels n.. of destination pixel dpix)
// Round 4: the epilogue of a tile row group is TWO loops -- every global operand of its NRD output quads is fetched first
// (epi_fetch), then the arithmetic and the stores follow (epi_finish).  Loads and stores retire in order on this memory pipeline:
// with one fused "load, wait, store" per quad, hipcc's wait for quad k's operands also waited for quad k-1's store to complete
// (8 chained store latencies per wave and tile in the residual / activation-mask / row-scale epilogues, worst inside the step
// where three streams share the memory system).  Same operations in the same order per element: results unchanged.
struct EpiOps {
  float4 bb, a4, r4, p4;
  float rsc;
};
__device__ inline EpiOps epi_fetch(size_t dpix, int n, int flags, const FastGeom& g, const float* __restrict__ bias,
                                   const float* __restrict__ residual, const float* __restrict__ rowscale,
                                   const float* __restrict__ actmask, const float* __restrict__ dst, bool accumulate) {
  EpiOps o;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  // (real branches: written as selects, hipcc turns a run-time-flagged load into a load through a pointer to a zero in scratch)
  o.rsc = 1.f;
  o.bb = z;
  o.a4 = z;
  o.r4 = z;
  if (flags & SRHIP_EPI_ROWSCALE) {
    asm volatile("" ::: "memory");
    o.rsc = rowscale[dpix];
  }
  if (flags & SRHIP_EPI_BIAS) {
    asm volatile("" ::: "memory");
    o.bb = *reinterpret_cast<const float4*>(bias + n);
  }
  if (flags & SRHIP_EPI_ACTMASK) {
    asm volatile("" ::: "memory");
    o.a4 = *reinterpret_cast<const float4*>(actmask + dpix * g.ldd + n);
  }
  if (flags & SRHIP_EPI_RESIDUAL) {
    asm volatile("" ::: "memory");
    o.r4 = *reinterpret_cast<const float4*>(residual + dpix * g.ldr + n);
  }
  o.p4 = z;
  if (accumulate) {                                   // a real branch: as a select hipcc loads through a pointer to a zero in scratch
    asm volatile("" ::: "memory");
    o.p4 = *reinterpret_cast<const float4*>(dst + dpix * g.ldd + n);
  }
  return o;
}
__device__ inline void epi_finish(float4 v, const EpiOps& e, size_t dpix, int n, int flags, const FastGeom& g, float* __restrict__ dst,
                                  bool accumulate) {
  if (flags & SRHIP_EPI_ROWSCALE) {
    v.x *= e.rsc; v.y *= e.rsc; v.z *= e.rsc; v.w *= e.rsc;
  }
  if (flags & SRHIP_EPI_BIAS) {
    v.x += e.bb.x; v.y += e.bb.y; v.z += e.bb.z; v.w += e.bb.w;
  }
  if (flags & SRHIP_EPI_LRELU) {
    v.x = v.x > 0.f ? v.x : v.x * g.slope;
    v.y = v.y > 0.f ? v.y : v.y * g.slope;
    v.z = v.z > 0.f ? v.z : v.z * g.slope;
    v.w = v.w > 0.f ? v.w : v.w * g.slope;
  }
  if (flags & SRHIP_EPI_ACTMASK) {
    v.x = e.a4.x > 0.f ? v.x : v.x * g.slope;
    v.y = e.a4.y > 0.f ? v.y : v.y * g.slope;
    v.z = e.a4.z > 0.f ? v.z : v.z * g.slope;
    v.w = e.a4.w > 0.f ? v.w : v.w * g.slope;
  }
  if (flags & SRHIP_EPI_RESIDUAL) {
    v.x += e.r4.x; v.y += e.r4.y; v.z += e.r4.z; v.w += e.r4.w;
  }
  float4* o = reinterpret_cast<float4*>(dst + dpix * g.ldd + n);
  if (accumulate) {
    v.x += e.p4.x; v.y += e.p4.y; v.z += e.p4.z; v.w += e.p4.w;
  }
  // Conv outputs are streamed out with the non-temporal hint: nothing in this kernel reads them back, and keeping them
  // out of the L2's way is worth 3-4 % on the 64 -> 256 fprop (95.6 MB written) and 0.4 % on the step.
  if (g.dst2_pp != nullptr) {                       // the same four channels as padded planes: octet n / 8, hi half at + (n & 4) * 2, lo 16 bytes further
    const unsigned hwd = (unsigned)(g.Hd * g.Wd);
    const unsigned img = (unsigned)dpix / hwd, rem = (unsigned)dpix - img * hwd;
    const unsigned oy = rem / (unsigned)g.Wd, ox = rem - oy * (unsigned)g.Wd;
    const size_t row = (size_t)g.dst2_guard + ((size_t)img * (g.Hd + 1) + oy) * (g.Wd + 1) + ox;
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t h01 = {(__bf16)v.x, (__bf16)v.y}, h23 = {(__bf16)v.z, (__bf16)v.w};
    const unsigned uh01 = __builtin_bit_cast(unsigned, h01), uh23 = __builtin_bit_cast(unsigned, h23);
    const bf16x2_t l01 = {(__bf16)(v.x - __uint_as_float(uh01 << 16)), (__bf16)(v.y - __uint_as_float(uh01 & 0xffff0000u))};
    const bf16x2_t l23 = {(__bf16)(v.z - __uint_as_float(uh23 << 16)), (__bf16)(v.w - __uint_as_float(uh23 & 0xffff0000u))};
    // The two lanes of an octet (adjacent lanes, adjacent quads: the row-group loop's mapping) trade halves so that each stores ONE whole
    // 16-byte piece -- the even lane the octet's 8 hi halves, the odd lane its 8 lo halves: a store instruction then writes whole rows
    // (two 8-byte stores per lane wrote 16 bytes of every 32 per instruction and the memory side counted them as partial lines).
    const bool odd = ((n >> 2) & 1) != 0;
    const unsigned ul01 = __builtin_bit_cast(unsigned, l01), ul23 = __builtin_bit_cast(unsigned, l23);
    const unsigned r0 = pair_swap(odd ? uh01 : ul01), r1 = pair_swap(odd ? uh23 : ul23);   // even lane receives the odd lane's hi, odd lane the even lane's lo
    unsigned* o2 = static_cast<unsigned*>(g.dst2_pp) + row * g.K + (n >> 3) * 8 + (odd ? 4 : 0);
    *reinterpret_cast<uint4*>(o2) = odd ? make_uint4(r0, r1, ul01, ul23) : make_uint4(uh01, uh23, r0, r1);
  }
  if (g.flags & 0x400) {                            // srhip_debug_set(3, 0x400): plain stores (A/B)
    *o = v;
    return;
  }
  __builtin_nontemporal_store(v.x, &o->x);
  __builtin_nontemporal_store(v.y, &o->y);
  __builtin_nontemporal_store(v.z, &o->z);
  __builtin_nontemporal_store(v.w, &o->w);
}
// one output quad, fetch and finish together (kernels whose epilogue is not a row-group loop)
__device__ inline void epi_apply_store(float4 v, size_t dpix, int n, int flags, const FastGeom& g,
                                       const float* __restrict__ bias, const float* __restrict__ residual,
                                       const float* __restrict__ rowscale, const float* __restrict__ actmask,
                                       float* __restrict__ dst) {
  const EpiOps e = epi_fetch(dpix, n, flags, g, bias, residual, rowscale, actmask, dst, g.accumulate != 0);
  epi_finish(v, e, dpix, n, flags, g, dst, g.accumulate != 0);
}

// ================================================================================================ //
template <int BM, int BN, int EPI, int MATH>
__global__ __launch_bounds__(256) void fast_conv_dma_kernel(const float* __restrict__ src, const float* __restrict__ wt,
                                                             const float* __restrict__ bias,
                                                             const float* __restrict__ residual,
                                                             const float* __restrict__ rowscale,
                                                             const float* __restrict__ chanscale,
                                                             const float* __restrict__ actmask,
                                                             float* __restrict__ dst, FastGeom g_in, int nblk_m,
                                                             int nblk_n, PhaseSet ps) {
  constexpr int WM = 2, WN = 2, BK = 16;
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int AI = BM / 64, BI = BN / 64;            // DMA instructions per wave per chunk (16 rows each)
  constexpr int STAGE_B = (BM + BN) * 64;              // bytes per stage (64 B per row)
  __shared__ __attribute__((aligned(1024))) char lds[3 * STAGE_B];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps per-wave control flow on the scalar unit
  FastGeom g = g_in;
  int bid = blockIdx.x;
  if (ps.n > 1) {                                      // the phases of a strided data gradient in one launch (PhaseSet, conv_dev.h)
    int k = 0;
    if (bid >= ps.first[1]) k = 1;
    if (ps.n > 2 && bid >= ps.first[2]) k = 2;
    if (ps.n > 3 && bid >= ps.first[3]) k = 3;
    bid -= ps.first[k];
    const PhaseSet::P q = ps.p[k];
    g.ph = q.ph; g.pw = q.pw; g.OH = q.OH; g.OW = q.OW; g.kh0 = q.kh0; g.kw0 = q.kw0; g.TH = q.TH; g.TW = q.TW;
    g.dh0 = q.dh0; g.dw0 = q.dw0; g.M = q.M;
    nblk_m = q.nblk_m;
  }
  const int tile = xcd_tile(bid, nblk_m * nblk_n);
  const int tile_n = tile % nblk_n, tile_m = tile / nblk_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  // ---- DMA bookkeeping: this lane feeds slot (row, s = lane&3) of rows wave*16*AI + 16*i + (lane>>2) ----
  const int OHOW = g.OH * g.OW;
  int abase[AI];
  unsigned amask[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int row = wave * 16 * AI + 16 * i + (lane >> 2);
    const int q = (lane & 3) ^ ((row >> 2) & 3);
    const int m = m0 + row;
    abase[i] = 0;
    amask[i] = 0;
    if (m < g.M) {
      const int n = m / OHOW;
      const int rem = m - n * OHOW;
      const int oh = rem / g.OW;
      const int ow = rem - oh * g.OW;
      const int sh0 = oh * g.ss, sw0 = ow * g.ss;
      abase[i] = ((n * g.Hs + sh0) * g.Ws + sw0) * g.lds + q * 4;
      unsigned mk = 0;
      for (int th = 0; th < g.TH; ++th) {
        const int sh = sh0 + g.dh0 + th * g.dhs;
        for (int tw = 0; tw < g.TW; ++tw) {
          const int sw = sw0 + g.dw0 + tw * g.dws;
          if (sh >= 0 && sh < g.Hs && sw >= 0 && sw < g.Ws) mk |= 1u << (th * g.TW + tw);
        }
      }
      amask[i] = mk;
    }
  }
  int bbase[BI];
  bool bval[BI];
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int row = wave * 16 * BI + 16 * j + (lane >> 2);
    const int q = (lane & 3) ^ ((row >> 2) & 3);
    const int n = n0 + row;
    bval[j] = n < g.K;
    bbase[j] = n * g.ldw + q * 4;
  }
  const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + wave * 16 * AI * 64);
  const unsigned b_dst = __builtin_amdgcn_readfirstlane(lds_base + BM * 64 + wave * 16 * BI * 64);

  const int CC = g.C / BK;
  const int nk = g.TH * g.TW * CC;
  int th = 0, tw = 0, cc = 0;
  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, g.src_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, g.w_bytes, 0x00020000);
  auto issue = [&](int stage) {
    const int tapoff = ((g.dh0 + th * g.dhs) * g.Ws + (g.dw0 + tw * g.dws)) * g.lds + cc * BK;
    const int wk = ((g.kh0 + th * g.khs) * g.KW + (g.kw0 + tw * g.kws)) * g.C + cc * BK;
    const int bit = th * g.TW + tw;
    const unsigned so = stage * STAGE_B;
    // buffer-descriptor DMA: a lane that feeds padding carries an out-of-range offset and the hardware writes zeros
#pragma unroll
    for (int i = 0; i < AI; ++i)
      lds_dma16_buf(((amask[i] >> bit) & 1u) ? (unsigned)(abase[i] + tapoff) * 4u : F_OOB, rs_a, a_dst + so + i * 1024);
#pragma unroll
    for (int j = 0; j < BI; ++j) lds_dma16_buf(bval[j] ? (unsigned)(bbase[j] + wk) * 4u : F_OOB, rs_b, b_dst + so + j * 1024);
    // taps innermost: the 9 taps of one 16-channel chunk re-read the same 64-byte segments of ~3 image
    // rows back to back (L1/L2 hits); with taps outermost a 256-channel input was re-fetched 9x from
    // beyond L2 (FETCH_SIZE 788 MB vs 96 MB algorithmic, profiles/r01_conv_pmc_summary.txt)
    if (++tw == g.TW) {
      tw = 0;
      if (++th == g.TH) {
        th = 0;
        ++cc;
      }
    }
  };

  // ---- fragment addresses (bytes inside a stage): slot (row, q ^ ((row>>2)&3)), q = ks*2 + khalf ----
  const int wm = wave >> 1, wn = wave & 1;
  const int khalf = lane >> 5, l31 = lane & 31;
  // MATH 0 (fp32 MFMA 32x32x2, 4 k per read): lane half h takes quads q = h (ks 0) and q = 2 + h (ks 1)
  // MATH 1 (bf16 MFMA 32x32x16, 8 k per lane):  lane half h takes quads q = 2h and 2h + 1
  int aoff[TM], boff[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int row = wm * WTM + t * 32 + l31;
    aoff[t] = row * 64 + ((((MATH ? 2 : 1) * khalf) ^ ((row >> 2) & 3)) << 4);
  }
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int row = wn * WTN + u * 32 + l31;
    boff[u] = BM * 64 + row * 64 + ((((MATH ? 2 : 1) * khalf) ^ ((row >> 2) & 3)) << 4);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  // A-operand channel scale (CLAM's s folded into the attention tail's 1x1 conv): fragment value
  // A[row][k] *= chanscale[image(row)][channel(k)], applied to the registers right after the LDS read
  const bool cscale = ((EPI >= 0 ? EPI : g.flags) & SRHIP_EPI_CHANSCALE) != 0;
  const float* csrow[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int m = m0 + wm * WTM + t * 32 + l31;
    csrow[t] = chanscale + (size_t)((m < g.M ? m : 0) / OHOW) * g.C + khalf * 4;
  }
  const int T_taps = g.TH * g.TW;
  int c_tap = 0, c_cc = 0;                           // compute-side position in the (cc, tap) loop nest

  auto chunk_sync = [&](int kc) {
    if (kc + 1 < nk)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");   // chunk kc landed; kc+1 may still fly
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // everyone's chunk kc visible; everyone done with chunk kc-1
    asm volatile("" ::: "memory");
  };
  auto read_frags = [&](int stage, float4 (&af)[2][TM], float4 (&bf)[2][TN]) {
    const char* sb = lds + stage * STAGE_B;
#pragma unroll
    for (int t = 0; t < TM; ++t) af[0][t] = *reinterpret_cast<const float4*>(sb + aoff[t]);
#pragma unroll
    for (int u = 0; u < TN; ++u) bf[0][u] = *reinterpret_cast<const float4*>(sb + boff[u]);
#pragma unroll
    for (int t = 0; t < TM; ++t) af[1][t] = *reinterpret_cast<const float4*>(sb + (aoff[t] ^ (MATH ? 16 : 32)));
#pragma unroll
    for (int u = 0; u < TN; ++u) bf[1][u] = *reinterpret_cast<const float4*>(sb + (boff[u] ^ (MATH ? 16 : 32)));
    if (cscale) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < TM; ++t) {
          const float4 sc = *reinterpret_cast<const float4*>(csrow[t] + c_cc * BK + (MATH ? ks * 4 + khalf * 4 : ks * 8));
          af[ks][t].x *= sc.x;
          af[ks][t].y *= sc.y;
          af[ks][t].z *= sc.z;
          af[ks][t].w *= sc.w;
        }
      if (++c_tap == T_taps) {
        c_tap = 0;
        ++c_cc;
      }
    }
  };

  if (MATH == 0 && nk > 0) {
    issue(0);
    if (nk > 1) issue(1);
    int stage = 0, nstage = 2;                       // nstage: where chunk kc+2 goes
    for (int kc = 0; kc < nk; ++kc) {
      chunk_sync(kc);
      if (kc + 2 < nk) issue(nstage);
      float4 af[2][TM], bf[2][TN];
      read_frags(stage, af, bf);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(af[ks][t].x, bf[ks][u].x, acc[t][u]);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(af[ks][t].y, bf[ks][u].y, acc[t][u]);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(af[ks][t].z, bf[ks][u].z, acc[t][u]);
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(af[ks][t].w, bf[ks][u].w, acc[t][u]);
      }
      stage = stage == 2 ? 0 : stage + 1;
      nstage = nstage == 2 ? 0 : nstage + 1;
    }
  }
  if (MATH >= 1 && nk > 0) {
    constexpr int PROD = MATH >= 1 ? MATH - 1 : 0;
    // split-bf16: A fragments are split in registers, B was split when it was packed.  (Interleaving the split of
    // chunk kc+1 with the MFMAs of chunk kc by hand measured the same: the loop is bound by LDS-DMA issue and the
    // per-chunk barrier, not by VALU/MFMA overlap -- DESIGN.md.)
    issue(0);
    if (nk > 1) issue(1);
    int stage = 0, nstage = 2;
    for (int kc = 0; kc < nk; ++kc) {
      chunk_sync(kc);
      if (kc + 2 < nk) issue(nstage);
      float4 af[2][TM], bf[2][TN];
      read_frags(stage, af, bf);
      bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        if (PROD == 0) split_bf16x8(af[0][t], af[1][t], ah[t], al[t]);
        else ah[t] = al[t] = round16x8<PROD>(af[0][t], af[1][t]);
      }
#pragma unroll
      for (int u = 0; u < TN; ++u) {   // weights were split / rounded when they were packed (fast_pack_store)
        bh[u] = __builtin_bit_cast(bf16x8_t, bf[0][u]);
        bl[u] = PROD == 0 ? __builtin_bit_cast(bf16x8_t, bf[1][u]) : bh[u];
      }
#pragma unroll
      for (int i = 0; i < nprod<PROD>() * TM * TN; ++i) {   // product order al*bh, ah*bl, ah*bh; accumulator chains interleaved
        const int grp = PROD == 0 ? i / (TM * TN) : 2, t = (i % (TM * TN)) / TN, u = i % TN;
        acc[t][u] = mma16<PROD>(grp == 0 ? al[t] : ah[t], grp == 1 ? bl[u] : bh[u], acc[t][u]);
      }
      stage = stage == 2 ? 0 : stage + 1;
      nstage = nstage == 2 ? 0 : nstage + 1;
    }
  }

  // ---- epilogue: accumulators -> LDS (per-wave region of the now idle ring) -> row-contiguous float4s:
  // 16-byte loads of bias / mask / residual and 16-byte stores (4 rows x 256 B per wave instruction)
  // instead of 64 scalar stores per lane.  C/D map of the 32x32 MFMA: col = lane&31,
  // row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const int flags = EPI >= 0 ? EPI : g.flags;
  __syncthreads();                                   // every wave is done reading the ring
  float* wl = reinterpret_cast<float*>(lds) + wave * (32 * WTN);
  constexpr int QPRW = WTN / 4;                      // float4 per tile row
  constexpr int NRD = 32 * QPRW / 64;                // float4 reads per lane per 32-row half
#pragma unroll
  for (int t = 0; t < TM; ++t) {
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * khalf) * WTN + u * 32 + l31] = acc[t][u][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // wave-private region: no block barrier needed
    constexpr int EB = NRD < 4 ? NRD : 4;               // quads per fetch / finish batch
#pragma unroll
    for (int i0 = 0; i0 < NRD; i0 += EB) {
      float4 vq[EB];
      EpiOps eo[EB];
      unsigned dpx[EB];                                   // pixel index (< 2^31: checked on the host side)
      int nq[EB];
      bool okq[EB];
#pragma unroll
      for (int j = 0; j < EB; ++j) {                    // every global operand of the batch first ...
        const int idx = (i0 + j) * 64 + lane;
        const int row = idx / QPRW, cq = idx - row * QPRW;
        vq[j] = *reinterpret_cast<const float4*>(wl + row * WTN + cq * 4);
        const int m = m0 + wm * WTM + t * 32 + row;
        nq[j] = n0 + wn * WTN + cq * 4;
        okq[j] = m < g.M && nq[j] < g.K;
        size_t dpix = (size_t)(okq[j] ? m : 0);
        if (!g.dst_identity) {
          const int mm = okq[j] ? m : 0;
          const int nimg = mm / OHOW;
          const int rem = mm - nimg * OHOW;
          const int oh = rem / g.OW;
          const int ow = rem - oh * g.OW;
          dpix = ((size_t)nimg * g.Hd + (oh * g.dsd + g.ph)) * g.Wd + (ow * g.dsd + g.pw);
        }
        dpx[j] = (unsigned)dpix;
        if (!okq[j]) nq[j] = 0;
        eo[j] = epi_fetch(dpix, nq[j], flags, g, bias, residual, rowscale, actmask, dst, EPI < 0 && g.accumulate != 0);
      }
#pragma unroll
      for (int j = 0; j < EB; ++j)                      // ... then the arithmetic and the stores (see epi_fetch)
        if (okq[j]) epi_finish(vq[j], eo[j], dpx[j], nq[j], flags, g, dst, EPI < 0 && g.accumulate != 0);
    }
    if (t + 1 < TM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // reads done before the region is rewritten
  }
}

// ================================================================================================ //
// fprop / dgrad of stride-1 3x3 convolutions in split-bf16 arithmetic: "patch" kernel.
// In fast_conv_dma_kernel every tap re-fetches its own 128-row A tile from L2; at the bf16 MFMA rate the kernel
// is then bound by LDS-DMA issue (4 x 1 KiB per wave per 12 MFMAs), not by the matrix pipe (ablation in
// DESIGN.md).  Here a block owns a PH x PW patch of output pixels (PH*PW <= 128) and keeps the (PH+2) x (PW+2)
// input halo of one 16-channel chunk in LDS; all nine taps read their A fragments from that one patch at shifted
// row addresses, so A traffic drops ~6x and only the B (weight) tile is streamed per tap.  Each wave converts the
// patch pieces it fetched itself from fp32 to the split hi|lo layout IN PLACE (once per element, instead of once
// per fragment read in every tap and wave), so fragments of both operands come out of LDS ready for the MFMA.
//   LDS: 2 patch buffers (chunk cc / cc+1) + a 3-stage ring of B tiles (one tap each), one barrier per tap.
//   Results are bit-identical to fast_conv_dma_kernel<.., MATH 1>: same split, same product and chunk order.
// ================================================================================================ //

template <int BN, int EPI, int PROD = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv_patch_kernel(const float* __restrict__ src, const float* __restrict__ wt,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ residual,
                                                          const float* __restrict__ actmask, float* __restrict__ dst,
                                                          FastGeom g, PatchGeom pg, int nblk_m, int nblk_n) {
  constexpr int NW = 4, BK = 16;
  constexpr int WTM = 64, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int BPW = BN / 64;                      // B DMA pieces per wave per tap
  constexpr int MAXP = 3;                           // A patch pieces per wave: 12 pieces = 192 rows per patch
  constexpr int PATCH_B = 12 * 1024;
  constexpr int BSTAGE_B = BN * 64;
  constexpr int EPI_B = NW * 32 * WTN * 4;
  constexpr int LDS_B = 2 * PATCH_B + 3 * BSTAGE_B > EPI_B ? 2 * PATCH_B + 3 * BSTAGE_B : EPI_B;
  __shared__ __attribute__((aligned(1024))) char lds[LDS_B];
  __shared__ int pix_tab[128];                      // lane-row -> (orow << 16 | ocol): patch_pixel (two integer divisions) once per row

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_tile(blockIdx.x, nblk_m * nblk_n);
  const int tile_n = tile % nblk_n, pid = tile / nblk_n;
  const int n0 = tile_n * BN;
  const int tpi = pg.tiles_h * pg.tiles_w;
  const int img = pid / tpi;
  const int prem = pid - img * tpi;
  const int ty = prem / pg.tiles_w, tx = prem - ty * pg.tiles_w;
  const int oh0 = ty * pg.PH, ow0 = tx * pg.PW;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  if (tid < 128) {
    int pr_, pc_;
    patch_pixel(tid, pg.PH, pg.PW, pg.gmap, pr_, pc_);
    pix_tab[tid] = (pr_ << 16) | pc_;
  }
  __syncthreads();

  // ---- A patch DMA: piece p = k*NW + wave covers patch rows 16p .. 16p+15; this lane feeds (row, slot lane&3).
  // Every wave always moves MAXP pieces (rows past the patch come from the zero block and are never read), so the
  // number of DMAs in flight is the same compile-time constant for all waves.
  const int swz = (lane >> 4) & 3;                  // ((16p + lane/4) >> 2) & 3
  const int aq = (lane & 3) ^ swz;                  // global 16-byte quad held by this lane's slot
  int abase[MAXP];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int row = (k * NW + wave) * 16 + (lane >> 2);
    abase[k] = -1;
    if (row < pg.PR) {
      const int pi = row / pg.PWP, pj = row - pi * pg.PWP;
      const int sh = oh0 + pg.lo_h + pi, sw = ow0 + pg.lo_w + pj;
      if (sh >= 0 && sh < g.Hs && sw >= 0 && sw < g.Ws) abase[k] = ((img * g.Hs + sh) * g.Ws + sw) * g.lds + aq * 4;
    }
  }
  int bbase[BPW];
  bool bval[BPW];
#pragma unroll
  for (int j = 0; j < BPW; ++j) {
    const int n = n0 + wave * 16 * BPW + 16 * j + (lane >> 2);
    bval[j] = n < g.K;
    bbase[j] = n * g.ldw + aq * 4;                  // ((row >> 2) & 3) == swz here too (16-row pieces)
  }
  const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
  const unsigned b_dst = __builtin_amdgcn_readfirstlane(lds_base + 2 * PATCH_B + wave * BPW * 1024);

  const int CC = g.C / BK;
  int wtap[9];                                      // packed-weight column of tap t (scalar registers)
#pragma unroll
  for (int t = 0; t < 9; ++t) wtap[t] = ((g.kh0 + (t / 3) * g.khs) * g.KW + (g.kw0 + (t % 3) * g.kws)) * g.C;

  // DMA through buffer descriptors: lanes that feed padding (outside the image / past the last destination channel)
  // carry an out-of-range offset and the hardware delivers zeros -- one 32-bit add per DMA instead of a 64-bit
  // address and a pointer select (measured -2.5 % on the 64 -> 256 fprop, bit-identical)
  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, g.src_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, g.w_bytes, 0x00020000);
  unsigned aoffb[MAXP], boffb[BPW];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) aoffb[k] = abase[k] >= 0 ? (unsigned)abase[k] * 4u : F_OOB;
#pragma unroll
  for (int j = 0; j < BPW; ++j) boffb[j] = bval[j] ? (unsigned)bbase[j] * 4u : F_OOB;
  auto issue_a = [&](int buf, int k, int cc) {      // one 1 KiB piece of the patch of chunk cc
    lds_dma16_buf(aoffb[k] + (unsigned)(cc * BK * 4), rs_a, a_dst + buf * PATCH_B + k * (NW * 1024));
  };
  auto issue_b = [&](int stage, int tap, int cc) {  // the B tile of (chunk cc, tap)
    const int wk = wtap[tap] + cc * BK;
#pragma unroll
    for (int j = 0; j < BPW; ++j) lds_dma16_buf(boffb[j] + (unsigned)(wk * 4), rs_b, b_dst + stage * BSTAGE_B + j * 1024);
  };
  // fp32 -> split bf16 in place for one piece this wave fetched: lanes 2i, 2i+1 hold the two quads (8 consecutive
  // channels) of a half row; the lane with the even quad keeps the 8 hi halves, the odd one the 8 lo halves
  auto convert_piece = [&](int buf, int k) {
    float4* slot = reinterpret_cast<float4*>(lds + buf * PATCH_B + (k * NW + wave) * 1024 + lane * 16);
    const float4 own = *slot;
    float4 oth;
    oth.x = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.x), 0xB1, 0xF, 0xF, true));
    oth.y = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.y), 0xB1, 0xF, 0xF, true));
    oth.z = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.z), 0xB1, 0xF, 0xF, true));
    oth.w = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.w), 0xB1, 0xF, 0xF, true));
    const bool odd = aq & 1;                        // this lane's quad is the second half of the 8-group
    bf16x8_t hi, lo;
    if (PROD == 0) {
      split_bf16x8(odd ? oth : own, odd ? own : oth, hi, lo);
      *reinterpret_cast<bf16x8_t*>(slot) = odd ? lo : hi;
    } else if (!odd) {                              // single product: the even lane's slot takes the 8 rounded values
      *reinterpret_cast<bf16x8_t*>(slot) = round16x8<PROD>(own, oth);
    }
  };

  // prologue DMAs first (whole patch of chunk 0, B tiles of taps 0 and 1): the address tables below are computed under their latency
#pragma unroll
  for (int k = 0; k < MAXP; ++k) issue_a(0, k, 0);
  issue_b(0, 0, 0);
  issue_b(1, 1, 0);

  // ---- fragment addressing: everything but the patch-buffer parity is fixed for the whole kernel ----
  const int wm = wave >> 1, wn = wave & 1;
  const int khalf = lane >> 5, l31 = lane & 31;
  int aoff[9][TM];                                  // byte offset (buffer 0) of this lane's hi quad for tap t
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int r = wm * WTM + t * 32 + l31;
    const int pt = pix_tab[r];
    const int orow = pt >> 16, ocol = pt & 0xffff;
    const int arow = orow < pg.PH ? orow * pg.PWP + ocol : 0;   // dead rows read pixel 0, never stored
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int a_th = (g.dh0 + (tap / 3) * g.dhs) - pg.lo_h, a_tw = (g.dw0 + (tap % 3) * g.dws) - pg.lo_w;
      const int pr = arow + a_th * pg.PWP + a_tw;
      aoff[tap][t] = pr * 64 + (((2 * khalf) ^ ((pr >> 2) & 3)) << 4);
    }
  }
  int boff[TN];
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int row = wn * WTN + u * 32 + l31;
    boff[u] = 2 * PATCH_B + row * 64 + (((2 * khalf) ^ ((row >> 2) & 3)) << 4);
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  // One tap of one chunk.  TAP and LAST (= this is the final chunk) are compile-time, so which DMAs are issued,
  // which piece is converted, the ring slots and the vmcnt count are all immediates: the loop body is a straight
  // line of [wait, barrier, <= 3 DMAs, 8 ds_read_b128, 12 MFMAs].
  //   DMA order inside a tap: the A piece (taps 0..2, for chunk cc+1), then the B tile of tap+2.
  //   At tap t the B tile of t (issued at t-2) must have landed; issued after it: the A piece of tap t-1 (if any)
  //   and the B tile of t+1 (if any) -> that many DMAs may stay in flight.
  auto do_tap = [&](auto tapc, auto lastc, int cc) {
    constexpr int TAP = decltype(tapc)::value;
    constexpr bool LAST = decltype(lastc)::value != 0;
    constexpr int NEWER = ((LAST && TAP == 8) ? 0 : BPW) + ((!LAST && TAP >= 1 && TAP <= MAXP) ? 1 : 0);
    wait_vmcnt<NEWER>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's in-place conversions are in LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int pbuf = cc & 1;
    if (!LAST && TAP < MAXP) issue_a(pbuf ^ 1, TAP, cc + 1);
    if (TAP + 2 < 9) issue_b((TAP + 2) % 3, TAP + 2, cc);
    else if (!LAST) issue_b((TAP + 2) % 3, TAP + 2 - 9, cc + 1);
    if (!LAST && TAP >= 2 && TAP - 2 < MAXP) convert_piece(pbuf ^ 1, TAP - 2);   // landed: it is older than B tile TAP
    const char* pb = lds + pbuf * PATCH_B;
    const char* sb = lds + (TAP % 3) * BSTAGE_B;
    bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      ah[t] = *reinterpret_cast<const bf16x8_t*>(pb + aoff[TAP][t]);
      al[t] = PROD == 0 ? *reinterpret_cast<const bf16x8_t*>(pb + (aoff[TAP][t] ^ 16)) : ah[t];
    }
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      bh[u] = *reinterpret_cast<const bf16x8_t*>(sb + boff[u]);
      bl[u] = PROD == 0 ? *reinterpret_cast<const bf16x8_t*>(sb + (boff[u] ^ 16)) : bh[u];
    }
#pragma unroll
    for (int i = 0; i < nprod<PROD>() * TM * TN; ++i) {   // same product order as fast_conv_dma_kernel
      const int grp = PROD == 0 ? i / (TM * TN) : 2, t = (i % (TM * TN)) / TN, u = i % TN;
      acc[t][u] = mma16<PROD>(grp == 0 ? al[t] : ah[t], grp == 1 ? bl[u] : bh[u], acc[t][u]);
    }
  };
  auto do_chunk = [&](auto lastc, int cc) {
    do_tap(IC<0>(), lastc, cc);
    do_tap(IC<1>(), lastc, cc);
    do_tap(IC<2>(), lastc, cc);
    do_tap(IC<3>(), lastc, cc);
    do_tap(IC<4>(), lastc, cc);
    do_tap(IC<5>(), lastc, cc);
    do_tap(IC<6>(), lastc, cc);
    do_tap(IC<7>(), lastc, cc);
    do_tap(IC<8>(), lastc, cc);
  };

  // (prologue DMAs: issued above, before the fragment tables) convert the patch once B tile 0 (issued after it) is in
  wait_vmcnt<BPW>();
#pragma unroll
  for (int k = 0; k < MAXP; ++k) convert_piece(0, k);
  for (int cc = 0; cc + 1 < CC; ++cc) do_chunk(IC<0>(), cc);
  do_chunk(IC<1>(), CC - 1);

  // ---- epilogue (as fast_conv_dma_kernel): accumulators -> wave-private LDS -> row-contiguous float4s ----
  const int flags = EPI >= 0 ? EPI : g.flags;
  __syncthreads();
  float* wl = reinterpret_cast<float*>(lds) + wave * (32 * WTN);
  constexpr int QPRW = WTN / 4;
  constexpr int NRD = 32 * QPRW / 64;
#pragma unroll
  for (int t = 0; t < TM; ++t) {
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * khalf) * WTN + u * 32 + l31] = acc[t][u][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    constexpr int EB = NRD < 4 ? NRD : 4;               // quads per fetch / finish batch (register budget: 17 values per quad at most)
#pragma unroll
    for (int i0 = 0; i0 < NRD; i0 += EB) {
      float4 vq[EB];
      EpiOps eo[EB];
      unsigned dpx[EB];                                   // pixel index (< 2^31: checked on the host side)
      int nq[EB];
      bool okq[EB];
#pragma unroll
      for (int j = 0; j < EB; ++j) {                    // every global operand of the batch first (epi_fetch) ...
        const int idx = (i0 + j) * 64 + lane;
        const int row = idx / QPRW, cq = idx - row * QPRW;
        vq[j] = *reinterpret_cast<const float4*>(wl + row * WTN + cq * 4);
        const int r = wm * WTM + t * 32 + row;
        const int pt = pix_tab[r];
        const int orow = pt >> 16, ocol = pt & 0xffff;
        const int oh = oh0 + orow, ow = ow0 + ocol;
        const int n = n0 + wn * WTN + cq * 4;
        okq[j] = !(orow >= pg.PH || oh >= g.OH || ow >= g.OW || n >= g.K);
        dpx[j] = okq[j] ? (unsigned)((img * g.Hd + oh) * g.Wd + ow) : 0u;
        nq[j] = okq[j] ? n : 0;
        eo[j] = epi_fetch(dpx[j], nq[j], flags, g, bias, residual, nullptr, actmask, dst, false);   // (the patch path never accumulates)
      }
#pragma unroll
      for (int j = 0; j < EB; ++j)                      // ... then the arithmetic and the stores
        if (okq[j]) epi_finish(vq[j], eo[j], dpx[j], nq[j], flags, g, dst, false);
    }
    if (t + 1 < TM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// ================================================================================================ //
// K-split form of the one-tile patch kernel for 64 destination channels (round 4): conv2's fprop (256 -> 64) and conv1's dgrad.
// With BN = 64 the 2 x 2 wave grid above gives a wave 64 pixels x 32 channels: 6 MFMAs for 6 ds_read_b128 per tap, against 12 for
// 8 in the 128-wide tile -- the fragment reads of three blocks per CU then take as long as their MFMAs (12 waves x 6 x 8 clk of
// LDS against 3 waves x 6 x 32 clk per SIMD), which is why these convs sat at 0.40 while the wide ones reached 0.44.  Here the
// second wave column splits K instead of N: wave (wm, wk) owns 64 pixels x ALL 64 channels and every second tap of the
// (chunk, tap) sequence -- T = 2 s + wk at step s --, so a step is one barrier, 8 fragment reads and 12 MFMAs per wave, like the
// wide tile.  The tap sequence of a PAIR of 16-channel chunks (18 taps, 9 steps) is the unit of the compile-time schedule: the
// patch of the pair's second chunk arrives during steps 0-1, the next pair's first patch during steps 5-6, weight tiles run
// four taps ahead in a six-slot ring (T mod 6: the pattern repeats per pair).  At the end the two K halves are added through LDS
// (each wave hands its partner the 32 channels the partner stores), then the usual epilogue.  The sum is grouped differently
// from fast_conv_dma_kernel's (two partial sums per output), so results agree with it to rounding, not bit for bit.
// Requires an even number of chunks (C % 32 == 0).
// ================================================================================================ //
template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv_patch_ks_kernel(
    const float* __restrict__ src, const float* __restrict__ wt, const float* __restrict__ bias, const float* __restrict__ residual,
    const float* __restrict__ actmask, float* __restrict__ dst, FastGeom g, PatchGeom pg, int nblk_m, int nblk_n) {
  constexpr int BN = 64, NW = 4, BK = 16, TM = 2, TN = 2, MAXP = 3, NST = 6;
  constexpr int PATCH_B = 12 * 1024, BSTAGE_B = BN * 64;
  constexpr int LDS_B = 2 * PATCH_B + NST * BSTAGE_B;          // 48 KB (the K exchange takes 32 KB of it, the epilogue 16 KB)
  __shared__ __attribute__((aligned(1024))) char lds[LDS_B];
  __shared__ int pix_tab[128];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_tile(blockIdx.x, nblk_m * nblk_n);
  const int tile_n = tile % nblk_n, pid = tile / nblk_n;
  const int n0 = tile_n * BN;
  const int tpi = pg.tiles_h * pg.tiles_w;
  const int img = pid / tpi;
  const int prem = pid - img * tpi;
  const int ty = prem / pg.tiles_w, tx = prem - ty * pg.tiles_w;
  const int oh0 = ty * pg.PH, ow0 = tx * pg.PW;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  if (tid < 128) {
    int pr_, pc_;
    patch_pixel(tid, pg.PH, pg.PW, pg.gmap, pr_, pc_);
    pix_tab[tid] = (pr_ << 16) | pc_;
  }
  __syncthreads();

  const int swz = (lane >> 4) & 3;
  const int aq = (lane & 3) ^ swz;
  unsigned aoffb[MAXP];
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int row = (k * NW + wave) * 16 + (lane >> 2);
    aoffb[k] = F_OOB;
    if (row < pg.PR) {
      const int pi = row / pg.PWP, pj = row - pi * pg.PWP;
      const int sh = oh0 + pg.lo_h + pi, sw = ow0 + pg.lo_w + pj;
      if (sh >= 0 && sh < g.Hs && sw >= 0 && sw < g.Ws) aoffb[k] = (unsigned)(((img * g.Hs + sh) * g.Ws + sw) * g.lds + aq * 4) * 4u;
    }
  }
  unsigned boffb;
  {
    const int n = n0 + wave * 16 + (lane >> 2);
    boffb = n < g.K ? (unsigned)(n * g.ldw + aq * 4) * 4u : F_OOB;
  }
  const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
  const unsigned b_dst = __builtin_amdgcn_readfirstlane(lds_base + 2 * PATCH_B + wave * 1024);
  const int CC = g.C / BK;
  int wtap[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wtap[t] = ((g.kh0 + (t / 3) * g.khs) * g.KW + (g.kw0 + (t % 3) * g.kws)) * g.C;
  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, g.src_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, g.w_bytes, 0x00020000);
  auto issue_a = [&](int buf, int k, int cc) {
    lds_dma16_buf(aoffb[k] + (unsigned)(cc * BK * 4), rs_a, a_dst + buf * PATCH_B + k * (NW * 1024));
  };
  auto issue_b = [&](int stage, int tap, int cc) {
    lds_dma16_buf(boffb + (unsigned)((wtap[tap] + cc * BK) * 4), rs_b, b_dst + stage * BSTAGE_B);
  };
  auto convert_piece = [&](int buf, int k) {          // as in conv_patch_kernel (split-bf16)
    float4* slot = reinterpret_cast<float4*>(lds + buf * PATCH_B + (k * NW + wave) * 1024 + lane * 16);
    const float4 own = *slot;
    float4 oth;
    oth.x = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.x), 0xB1, 0xF, 0xF, true));
    oth.y = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.y), 0xB1, 0xF, 0xF, true));
    oth.z = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.z), 0xB1, 0xF, 0xF, true));
    oth.w = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.w), 0xB1, 0xF, 0xF, true));
    const bool odd = aq & 1;
    bf16x8_t hi, lo;
    split_bf16x8(odd ? oth : own, odd ? own : oth, hi, lo);
    *reinterpret_cast<bf16x8_t*>(slot) = odd ? lo : hi;
  };

  // prologue DMAs first -- the patch of chunk 0, the weight tiles of taps 0..3 --, so that the address tables below (54 offsets,
  // two integer divisions per row) are computed under their latency: every block of a one-round launch pays this prologue at
  // the same time, with nothing else on the chip to hide it
#pragma unroll
  for (int k = 0; k < MAXP; ++k) issue_a(0, k, 0);
  issue_b(0, 0, 0);
  issue_b(1, 1, 0);
  issue_b(2, 2, 0);
  issue_b(3, 3, 0);

  // ---- fragment addressing by STEP: this wave's tap at step s of a chunk pair is T = 2 s + wk (chunk T / 9, tap T % 9) ----
  const int wm = wave >> 1, wk = wave & 1;
  const int khalf = lane >> 5, l31 = lane & 31;
  int aoffS[9][TM];                                 // byte offset of the hi quad, patch buffer (= chunk parity) included
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int r = wm * 64 + t * 32 + l31;
    const int pt = pix_tab[r];
    const int orow = pt >> 16, ocol = pt & 0xffff;
    const int arow = orow < pg.PH ? orow * pg.PWP + ocol : 0;
#pragma unroll
    for (int sidx = 0; sidx < 9; ++sidx) {
      const int T = 2 * sidx + wk;
      const int par = T >= 9 ? 1 : 0, tap = T - 9 * par;
      const int th = tap / 3, tw = tap - 3 * th;
      const int a_th = (g.dh0 + th * g.dhs) - pg.lo_h, a_tw = (g.dw0 + tw * g.dws) - pg.lo_w;
      const int pr = arow + a_th * pg.PWP + a_tw;
      aoffS[sidx][t] = par * PATCH_B + pr * 64 + (((2 * khalf) ^ ((pr >> 2) & 3)) << 4);
    }
  }
  int boffk[TN];                                    // ring base + this K group's slot of a step's pair + the fragment row
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int row = u * 32 + l31;
    boffk[u] = 2 * PATCH_B + wk * BSTAGE_B + row * 64 + (((2 * khalf) ^ ((row >> 2) & 3)) << 4);
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  // One step of a chunk pair (compile-time step index and "last pair" flag: every DMA, conversion, ring slot and vmcnt count is
  // an immediate).  DMAs per wave and step: s0 A A B B | s1 A B B | s2-s4 B B | s5 A A B B | s6 A B B | s7, s8 B B; the last pair
  // drops the next pair's patch (s5, s6) and weight tiles (s7, s8).  At step s the weight tiles of taps 2 s and 2 s + 1 (the last
  // two DMAs of step s - 2) must have landed: everything issued at step s - 1 may stay in flight.
  auto do_step = [&](auto sc, auto lastc, int cc0) {
    constexpr int S = decltype(sc)::value;
    constexpr bool LAST = decltype(lastc)::value != 0;
    constexpr int NEWER = S == 0 ? 2 : S == 1 ? 4 : S == 2 ? 3 : (S >= 3 && S <= 5) ? 2 : S == 6 ? (LAST ? 2 : 4) : S == 7 ? (LAST ? 2 : 3) : (LAST ? 0 : 2);
    wait_vmcnt<NEWER>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (S == 0) {
      issue_a(1, 0, cc0 + 1);
      issue_a(1, 1, cc0 + 1);
    }
    if (S == 1) issue_a(1, 2, cc0 + 1);
    if (!LAST && S == 5) {
      issue_a(0, 0, cc0 + 2);
      issue_a(0, 1, cc0 + 2);
    }
    if (!LAST && S == 6) issue_a(0, 2, cc0 + 2);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      constexpr int T0 = 2 * S + 4;
      const int T = T0 + d;
      if (T < 18) issue_b(T % NST, T % 9, cc0 + T / 9);
      else if (!LAST) issue_b(T % NST, (T - 18) % 9, cc0 + 2 + (T - 18) / 9);
    }
    if (S == 2) {
      convert_piece(1, 0);
      convert_piece(1, 1);
    }
    if (S == 3) convert_piece(1, 2);
    if (!LAST && S == 7) {
      convert_piece(0, 0);
      convert_piece(0, 1);
    }
    if (!LAST && S == 8) convert_piece(0, 2);
    const char* sb = lds + ((2 * S) % NST) * BSTAGE_B;
    bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      ah[t] = *reinterpret_cast<const bf16x8_t*>(lds + aoffS[S][t]);
      al[t] = *reinterpret_cast<const bf16x8_t*>(lds + (aoffS[S][t] ^ 16));
    }
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      bh[u] = *reinterpret_cast<const bf16x8_t*>(sb + boffk[u]);
      bl[u] = *reinterpret_cast<const bf16x8_t*>(sb + (boffk[u] ^ 16));
    }
#pragma unroll
    for (int i = 0; i < 3 * TM * TN; ++i) {         // same product order as the other split-bf16 kernels: al*bh, ah*bl, ah*bh
      const int grp = i / (TM * TN), t = (i % (TM * TN)) / TN, u = i % TN;
      acc[t][u] = mma16<0>(grp == 0 ? al[t] : ah[t], grp == 1 ? bl[u] : bh[u], acc[t][u]);
    }
  };
  auto do_pair = [&](auto lastc, int cc0) {
    do_step(IC<0>(), lastc, cc0);
    do_step(IC<1>(), lastc, cc0);
    do_step(IC<2>(), lastc, cc0);
    do_step(IC<3>(), lastc, cc0);
    do_step(IC<4>(), lastc, cc0);
    do_step(IC<5>(), lastc, cc0);
    do_step(IC<6>(), lastc, cc0);
    do_step(IC<7>(), lastc, cc0);
    do_step(IC<8>(), lastc, cc0);
  };

  // (prologue DMAs: issued above, before the fragment tables) the patch is converted once it is in; the weight tiles stay in flight
  wait_vmcnt<4>();
#pragma unroll
  for (int k = 0; k < MAXP; ++k) convert_piece(0, k);
  for (int cc0 = 0; cc0 + 2 < CC; cc0 += 2) do_pair(IC<0>(), cc0);
  do_pair(IC<1>(), CC - 2);

  // ---- add the two K halves: a wave hands its partner (same pixels, other K group) the 32 channels the partner stores ----
  __syncthreads();
  {
    float* xl = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) xl[((wave * TM + t) * 16 + r) * 64 + lane] = wk ? acc[t][0][r] : acc[t][1][r];
    __syncthreads();
    const int pw = wave ^ 1;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float o = xl[((pw * TM + t) * 16 + r) * 64 + lane];
        acc[t][0][r] = (wk ? acc[t][1][r] : acc[t][0][r]) + o;
      }
    __syncthreads();
  }

  // ---- epilogue: as conv_patch_kernel<64> with wn = wk (wave = 64 pixels x 32 channels) ----
  const int flags = EPI >= 0 ? EPI : g.flags;
  constexpr int WTN = 32;
  float* wl = reinterpret_cast<float*>(lds) + wave * (32 * WTN);
  constexpr int QPRW = WTN / 4;
  constexpr int NRD = 32 * QPRW / 64;
#pragma unroll
  for (int t = 0; t < TM; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * khalf) * WTN + l31] = acc[t][0][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float4 vq[NRD];
    EpiOps eo[NRD];
    unsigned dpx[NRD];
    int nq[NRD];
    bool okq[NRD];
#pragma unroll
    for (int j = 0; j < NRD; ++j) {
      const int idx = j * 64 + lane;
      const int row = idx / QPRW, cq = idx - row * QPRW;
      vq[j] = *reinterpret_cast<const float4*>(wl + row * WTN + cq * 4);
      const int r = wm * 64 + t * 32 + row;
      const int pt = pix_tab[r];
      const int orow = pt >> 16, ocol = pt & 0xffff;
      const int oh = oh0 + orow, ow = ow0 + ocol;
      const int n = n0 + wk * WTN + cq * 4;
      okq[j] = !(orow >= pg.PH || oh >= g.OH || ow >= g.OW || n >= g.K);
      dpx[j] = okq[j] ? (unsigned)((img * g.Hd + oh) * g.Wd + ow) : 0u;
      nq[j] = okq[j] ? n : 0;
      eo[j] = epi_fetch(dpx[j], nq[j], flags, g, bias, residual, nullptr, actmask, dst, false);
    }
#pragma unroll
    for (int j = 0; j < NRD; ++j)
      if (okq[j]) epi_finish(vq[j], eo[j], dpx[j], nq[j], flags, g, dst, false);
    if (t + 1 < TM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// ================================================================================================ //
// Stride-1 3x3 convolutions with <= 4 destination channels (generator tail conv 64 -> 3, discriminator head dgrad
// 64 -> 3) at full image size.  A 32-wide MFMA tile wastes 10x the arithmetic there and the op is HBM-bound
// (one read of the source); this kernel does the 1728 multiply-adds per pixel on the VALU in exact fp32:
//   * a block owns a 16 x 16 patch of output pixels, one thread per pixel, ND accumulators each;
//   * the 18 x 18 halo of a 16-channel chunk is staged in LDS (double-buffered, coalesced 16-byte loads,
//     16-byte slots XOR-swizzled by the pixel index so the nine shifted ds_read_b128 streams are conflict-free);
//   * weights are indexed uniformly, so they arrive through scalar loads and enter the FMAs as SGPR operands.
// ================================================================================================ //
template <int ND>
__global__ __launch_bounds__(256) void narrow_conv_kernel(const float* __restrict__ src, const float* __restrict__ wt,
                                                           const float* __restrict__ bias, float* __restrict__ dst,
                                                           FastGeom g, int tiles_h, int tiles_w, int lo_h, int lo_w) {
  constexpr int PW = 16, PH = 16, PWP = PW + 2, PR = (PH + 2) * PWP;   // 324 patch rows of 64 B
  __shared__ __attribute__((aligned(16))) float4 lds[2][PR * 4];
  const int tid = threadIdx.x;
  const int tpi = tiles_h * tiles_w;
  const int img = blockIdx.x / tpi;
  const int prem = blockIdx.x - img * tpi;
  const int ty = prem / tiles_w, tx = prem - ty * tiles_w;
  const int oh0 = ty * PH, ow0 = tx * PW;
  const int py = tid >> 4, px = tid & 15;

  // staging: 324 rows x 4 quads = 1296 float4 per chunk, 256 threads -> 6 per thread (last partially)
  int soff[6];
  int sdst[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int e = tid + i * 256;
    soff[i] = -1;
    sdst[i] = 0;
    if (e < PR * 4) {
      const int row = e >> 2, q = e & 3;
      const int pi = row / PWP, pj = row - pi * PWP;
      const int sh = oh0 + lo_h + pi, sw = ow0 + lo_w + pj;
      sdst[i] = row * 4 + (q ^ (row & 3));
      if (sh >= 0 && sh < g.Hs && sw >= 0 && sw < g.Ws) soff[i] = ((img * g.Hs + sh) * g.Ws + sw) * g.lds + q * 4;
      else soff[i] = -2;                               // inside the patch, outside the image: zero
    }
  }
  const int CC = g.C / 16;
  float4 stage[6];
  auto fetch = [&](int cc) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      stage[i] = soff[i] >= 0 ? *reinterpret_cast<const float4*>(src + (size_t)soff[i] + cc * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 6; ++i)
      if (soff[i] != -1) lds[buf][sdst[i]] = stage[i];
  };
  float acc[ND];
#pragma unroll
  for (int n = 0; n < ND; ++n) acc[n] = 0.f;

  fetch(0);
  commit(0);
  __syncthreads();
  for (int cc = 0; cc < CC; ++cc) {
    const int buf = cc & 1;
    if (cc + 1 < CC) fetch(cc + 1);
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {              // not unrolled: one tap's 16 x ND weights fit the SGPR file
      {
        const int th = tap / 3, tw = tap - th * 3;
        const int a_th = (g.dh0 + th * g.dhs) - lo_h, a_tw = (g.dw0 + tw * g.dws) - lo_w;
        const int row = (py + a_th) * PWP + px + a_tw;
        const int wk = ((g.kh0 + th * g.khs) * g.KW + (g.kw0 + tw * g.kws)) * g.C + cc * 16;   // uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = lds[buf][row * 4 + (q ^ (row & 3))];
#pragma unroll
          for (int n = 0; n < ND; ++n) {
            const float* w = wt + (size_t)n * g.ldw + wk + q * 4;                              // scalar loads
            acc[n] = fmaf(v.x, w[0], acc[n]);
            acc[n] = fmaf(v.y, w[1], acc[n]);
            acc[n] = fmaf(v.z, w[2], acc[n]);
            acc[n] = fmaf(v.w, w[3], acc[n]);
          }
        }
      }
    }
    if (cc + 1 < CC) commit(buf ^ 1);
    __syncthreads();
  }
  const int oh = oh0 + py, ow = ow0 + px;
  if (oh < g.OH && ow < g.OW) {
    const size_t dpix = ((size_t)img * g.Hd + oh) * g.Wd + ow;
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      if (n < g.K) {
        float v = acc[n];
        if (g.flags & SRHIP_EPI_BIAS) v += bias[n];
        if (g.flags & SRHIP_EPI_LRELU) v = v > 0.f ? v : v * g.slope;
        dst[dpix * g.ldd + n] = v;
      }
    }
  }
}

// ================================================================================================ //
// wgrad: dW[co][(tap,ci)] = sum_p dy[p][co] * xwin[p][(tap,ci)], split over pixel ranges.
// Both operands are pixel-major, so the LDS images are k-major ([pixel][channel]) and fragments are
// conflict-free ds_read_b32 of consecutive dwords.  Blocks with tile_n == 0 also emit the column sums
// of their dy tiles (bias gradient partials).
// ================================================================================================ //
struct WgradGeom {
  int N, H, W, C, ldx;       // x
  int Ho, Wo, K, ldy;        // dy
  int KH, KW, stride, pad;
  int P;                     // N*Ho*Wo pixels
  int Ktot;                  // KH*KW*C
  int nsplit, chunks_per_split;
  unsigned x_bytes, dy_bytes;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void fast_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ partial,
                                                          float* __restrict__ bias_partial,
                                                          const float* __restrict__ xrow,
                                                          const float* __restrict__ xchan, WgradGeom g) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int AV = (FBK * BM / 4) / 256, BV = (FBK * BN / 4) / 256;   // float4 per thread per chunk
  static_assert(AV >= 1 && BV >= 1, "tile too small");
  constexpr int STAGE = FBK * (LDA + LDB);
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int ntn = (g.Ktot + BN - 1) / BN;
  const int ntm = (g.K + BM - 1) / BM;
  // XCD-aware order: every tile (tile_m, tile_n) of one pixel split runs on the same XCD (blocks b, b+8,
  // b+16, ... share an L2), so a dy / x chunk is fetched from HBM once per split instead of once per tile
  int tile_n, tile_m, split;
  {
    const int tps = ntm * ntn;
    int bid = blockIdx.x;
    if (g.nsplit % 8 == 0) {
      const int j = bid >> 3;
      split = (j / tps) * 8 + (bid & 7);
      bid = j % tps;
    } else {
      split = bid / tps;
      bid -= split * tps;
    }
    tile_n = bid % ntn;
    tile_m = bid / ntn;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, g.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, g.dy_bytes, 0x00020000);

  const int c_begin = split * g.chunks_per_split;
  const int nchunks_total = (g.P + FBK - 1) / FBK;
  const int c_end = min(c_begin + g.chunks_per_split, nchunks_total);

  // A (dy) thread mapping: idx = tid + 256*j -> pixel row arow = idx / (BM/4), column quad ac
  constexpr int AQ = BM / 4, BQ = BN / 4;
  const int a_row0 = tid / AQ, a_c = tid - a_row0 * AQ;          // rows a_row0 + j*(256/AQ)
  const int b_row0 = tid / BQ, b_c = tid - b_row0 * BQ;          // rows b_row0 + j*(256/BQ)
  const bool a_colok = (m0 + a_c * 4) < g.K;                     // K % 4 == 0 on this path
  // this thread's B columns (a quad of input channels of ONE filter tap; C % 4 == 0)
  const int kcol = n0 + b_c * 4;
  const bool b_colok = kcol < g.Ktot;
  const int tap = kcol / g.C, ci0 = kcol - tap * g.C;
  const int kh = tap / g.KW, kw = tap - kh * g.KW;

  // B (x window) per-row pixel coordinates, advanced incrementally by FBK pixels per chunk
  int bn[BV], bho[BV], bwo[BV];
#pragma unroll
  for (int j = 0; j < BV; ++j) {
    const int p = c_begin * FBK + b_row0 + j * (256 / BQ);
    const int HoWo = g.Ho * g.Wo;
    bn[j] = p / HoWo;
    const int rem = p - bn[j] * HoWo;
    bho[j] = rem / g.Wo;
    bwo[j] = rem - bho[j] * g.Wo;
  }

  float4 ra[AV], rb[BV], rxs[BV];
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool xscale = xrow != nullptr || xchan != nullptr;
  auto load_tiles = [&](int kc) {
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      const int p = kc * FBK + a_row0 + j * (256 / AQ);
      const unsigned off = (p < g.P && a_colok) ? ((unsigned)p * g.ldy + m0 + a_c * 4) * 4u : F_OOB;
      ra[j] = bufload4(ry, off);
    }
#pragma unroll
    for (int j = 0; j < BV; ++j) {
      const int hi = bho[j] * g.stride - g.pad + kh, wi = bwo[j] * g.stride - g.pad + kw;
      const bool ok = b_colok && bn[j] < g.N && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
      const unsigned off = ok ? ((unsigned)((bn[j] * g.H + hi) * g.W + wi) * g.ldx + ci0) * 4u : F_OOB;
      rb[j] = bufload4(rx, off);
      if (xscale) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
        if (ok) {
          if (xchan) sc = *reinterpret_cast<const float4*>(xchan + (size_t)bn[j] * g.C + ci0);
          if (xrow) {
            const float r = xrow[(size_t)(bn[j] * g.H + hi) * g.W + wi];
            sc.x *= r; sc.y *= r; sc.z *= r; sc.w *= r;
          }
        }
        rxs[j] = sc;
      }
      bwo[j] += FBK;
      while (bwo[j] >= g.Wo) {
        bwo[j] -= g.Wo;
        if (++bho[j] == g.Ho) {
          bho[j] = 0;
          ++bn[j];
        }
      }
    }
  };
  auto store_tiles = [&](int stage) {
    float* a = lds + stage * STAGE;
#pragma unroll
    for (int j = 0; j < AV; ++j) {
      *reinterpret_cast<float4*>(a + (a_row0 + j * (256 / AQ)) * LDA + a_c * 4) = ra[j];
      bsum.x += ra[j].x;
      bsum.y += ra[j].y;
      bsum.z += ra[j].z;
      bsum.w += ra[j].w;
    }
    float* b = lds + stage * STAGE + FBK * LDA;
    if (xscale) {
#pragma unroll
      for (int j = 0; j < BV; ++j) {
        rb[j].x *= rxs[j].x;
        rb[j].y *= rxs[j].y;
        rb[j].z *= rxs[j].z;
        rb[j].w *= rxs[j].w;
      }
    }
#pragma unroll
    for (int j = 0; j < BV; ++j) *reinterpret_cast<float4*>(b + (b_row0 + j * (256 / BQ)) * LDB + b_c * 4) = rb[j];
  };

  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const int khalf = lane >> 5, l31 = lane & 31;
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
      const int stage = (kc - c_begin) & 1;
      if (kc + 1 < c_end) load_tiles(kc + 1);
      const float* a = lds + stage * STAGE + khalf * LDA + wm * WTM + l31;
      const float* b = lds + stage * STAGE + FBK * LDA + khalf * LDB + wn * WTN + l31;
      float av[2][TM], bv[2][TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) av[0][t] = a[t * 32];
#pragma unroll
      for (int u = 0; u < TN; ++u) bv[0][u] = b[u * 32];
#pragma unroll
      for (int kk = 0; kk < FBK / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < FBK / 2) {
#pragma unroll
          for (int t = 0; t < TM; ++t) av[nxt][t] = a[(kk + 1) * 2 * LDA + t * 32];
#pragma unroll
          for (int u = 0; u < TN; ++u) bv[nxt][u] = b[(kk + 1) * 2 * LDB + u * 32];
        }
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(av[cur][t], bv[cur][u], acc[t][u]);
      }
      if (kc + 1 < c_end) store_tiles(stage ^ 1);
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
        const int m = m0 + wm * WTM + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (m < g.K && n < g.Ktot) out[(size_t)m * g.Ktot + n] = acc[t][u][r];
      }
  }

  // bias-gradient partial: column sums of every dy tile this block staged (only the tile_n == 0 blocks)
  if (bias_partial != nullptr && tile_n == 0) {
    __syncthreads();
    float* red = lds;                                  // [256/AQ][BM]
    *reinterpret_cast<float4*>(red + a_row0 * BM + a_c * 4) = bsum;
    __syncthreads();
    if (tid < BM && m0 + tid < g.K) {
      float s = 0.f;
#pragma unroll
      for (int rr = 0; rr < 256 / AQ; ++rr) s += red[rr * BM + tid];
      bias_partial[(size_t)split * g.K + m0 + tid] = s;
    }
  }
}

// ---- wgrad, LDS-DMA variant: same ring / counted-vmcnt structure as fast_conv_dma_kernel.  Both
// operands are pixel-major, so a stage is simply [16 pixels][BM] + [16 pixels][BN] floats, written
// lane-linear by the DMA and read back as conflict-free ds_read_b32 (consecutive dwords) -- no swizzle.
// MATH 1 (SRHIP_MATH_BF16X3): a lane gathers its 8 consecutive pixels of one channel with 8 ds_read_b32 (still
// consecutive dwords across lanes), splits them into bf16 hi/lo and issues three 32x32x16 MFMAs per tile pair.
template <int BM, int BN, int WM, int WN, int WBK, int MATH>
__global__ __launch_bounds__(256) void fast_wgrad_dma_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                              float* __restrict__ partial,
                                                              float* __restrict__ bias_partial, WgradGeom g) {
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int AI = BM * WBK / 1024, BI = BN * WBK / 1024;   // DMA instructions per wave per chunk
  constexpr int ALR = BM / 4, BLR = BN / 4;              // lanes per pixel row
  constexpr int ARPI = 64 / ALR, BRPI = 64 / BLR;        // pixel rows per DMA instruction
  constexpr int STAGE_B = (BM + BN) * WBK * 4;
  __shared__ __attribute__((aligned(1024))) char lds[3 * STAGE_B];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps per-wave control flow on the scalar unit
  const int ntn = (g.Ktot + BN - 1) / BN;
  const int ntm = (g.K + BM - 1) / BM;
  // XCD-aware order: every tile (tile_m, tile_n) of one pixel split runs on the same XCD (blocks b, b+8,
  // b+16, ... share an L2), so a dy / x chunk is fetched from HBM once per split instead of once per tile
  int tile_n, tile_m, split;
  {
    const int tps = ntm * ntn;
    int bid = blockIdx.x;
    if (g.nsplit % 8 == 0) {
      const int j = bid >> 3;
      split = (j / tps) * 8 + (bid & 7);
      bid = j % tps;
    } else {
      split = bid / tps;
      bid -= split * tps;
    }
    tile_n = bid % ntn;
    tile_m = bid / ntn;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  const int c_begin = split * g.chunks_per_split;
  const int nchunks_total = (g.P + WBK - 1) / WBK;
  const int c_end = min(c_begin + g.chunks_per_split, nchunks_total);
  const int nk = c_end - c_begin;

  // A (dy): this lane feeds pixel row arow[i], channels m0 + acol*4 ..
  const int acol = lane % ALR;
  const bool a_colok = (m0 + acol * 4) < g.K;
  int arow[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) arow[i] = (wave * AI + i) * ARPI + lane / ALR;
  // B (x window): pixel row brow[j], columns kcol..kcol+3 of ONE tap
  const int bcolq = lane % BLR;
  const int kcol = n0 + bcolq * 4;
  const bool b_colok = kcol < g.Ktot;
  const int tap = kcol / g.C, ci0 = kcol - tap * g.C;
  const int kh = tap / g.KW, kw = tap - kh * g.KW;
  int bn[BI], bho[BI], bwo[BI];
  const int HoWo = g.Ho * g.Wo;
#pragma unroll
  for (int j = 0; j < BI; ++j) {
    const int p = c_begin * WBK + (wave * BI + j) * BRPI + lane / BLR;
    bn[j] = p / HoWo;
    const int rem = p - bn[j] * HoWo;
    bho[j] = rem / g.Wo;
    bwo[j] = rem - bho[j] * g.Wo;
  }
  const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + wave * AI * 1024);
  const unsigned b_dst = __builtin_amdgcn_readfirstlane(lds_base + BM * WBK * 4 + wave * BI * 1024);

  int kc_issue = c_begin;
  auto issue = [&](int stage) {
    const unsigned so = stage * STAGE_B;
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int p = kc_issue * WBK + arow[i];
      const float* ptr = (p < g.P && a_colok) ? dy + ((long)p * g.ldy + m0 + acol * 4) : g_zero16;
      lds_dma16(ptr, a_dst + so + i * 1024);
    }
#pragma unroll
    for (int j = 0; j < BI; ++j) {
      const int hi = bho[j] * g.stride - g.pad + kh, wi = bwo[j] * g.stride - g.pad + kw;
      const bool ok = b_colok && bn[j] < g.N && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
      const float* ptr = ok ? x + ((long)((bn[j] * g.H + hi) * g.W + wi) * g.ldx + ci0) : g_zero16;
      lds_dma16(ptr, b_dst + so + j * 1024);
      bwo[j] += WBK;
      while (bwo[j] >= g.Wo) {
        bwo[j] -= g.Wo;
        if (++bho[j] == g.Ho) {
          bho[j] = 0;
          ++bn[j];
        }
      }
    }
    ++kc_issue;
  };

  const int wm = wave / WN, wn = wave - wm * WN;
  const int khalf = lane >> 5, l31 = lane & 31;
  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  float bsum = 0.f;
  const bool want_bias = bias_partial != nullptr && tile_n == 0 && tid < BM;

  if (nk > 0) {
    issue(0);
    if (nk > 1) issue(1);
    int stage = 0, nstage = 2;
    for (int kc = 0; kc < nk; ++kc) {
      if (kc + 1 < nk)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AI + BI) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kc + 2 < nk) issue(nstage);
      const float* a = reinterpret_cast<const float*>(lds + stage * STAGE_B) + khalf * BM + wm * WTM + l31;
      const float* b = reinterpret_cast<const float*>(lds + stage * STAGE_B) + WBK * BM + khalf * BN + wn * WTN + l31;
      if (MATH == 0) {
#pragma unroll
        for (int kk = 0; kk < WBK / 2; ++kk) {
          float av[TM], bv[TN];
#pragma unroll
          for (int t = 0; t < TM; ++t) av[t] = a[kk * 2 * BM + t * 32];
#pragma unroll
          for (int u = 0; u < TN; ++u) bv[u] = b[kk * 2 * BN + u * 32];
#pragma unroll
          for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u) acc[t][u] = mfma32f(av[t], bv[u], acc[t][u]);
        }
      } else {
        // pixel rows khalf*8 .. khalf*8+7 of each 16-pixel step (a/b above start at row khalf: rebase to khalf*8)
        const float* a8 = a + 7 * khalf * BM;
        const float* b8 = b + 7 * khalf * BN;
        constexpr bool SPLIT = MATH == 1;               // MATH 2: one bf16 product (SRHIP_MATH_HALF)
#pragma unroll
        for (int ks = 0; ks < WBK / 16; ++ks) {
          bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
          for (int t = 0; t < TM; ++t) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = a8[(ks * 16 + j) * BM + t * 32];
            if (SPLIT) split_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), ah[t], al[t]);
            else ah[t] = al[t] = round16x8<1>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
          }
#pragma unroll
          for (int u = 0; u < TN; ++u) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = b8[(ks * 16 + j) * BN + u * 32];
            if (SPLIT) split_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), bh[u], bl[u]);
            else bh[u] = bl[u] = round16x8<1>(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
          }
          if (SPLIT) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
              for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh[u], acc[t][u], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
              for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl[u], acc[t][u], 0, 0, 0);
          }
#pragma unroll
          for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh[u], acc[t][u], 0, 0, 0);
        }
      }
      if (want_bias) {
        const float* col = reinterpret_cast<const float*>(lds + stage * STAGE_B) + tid;
#pragma unroll
        for (int r = 0; r < WBK; ++r) bsum += col[r * BM];
      }
      stage = stage == 2 ? 0 : stage + 1;
      nstage = nstage == 2 ? 0 : nstage + 1;
    }
  }

  // partial tile -> LDS (wave-private region of the idle ring) -> row-contiguous 16-byte stores
  float* out = partial + (size_t)split * g.K * g.Ktot;
  __syncthreads();
  float* wl = reinterpret_cast<float*>(lds) + wave * (32 * WTN);
  constexpr int QPRW = WTN / 4;
  constexpr int NRD = 32 * QPRW / 64;
#pragma unroll
  for (int t = 0; t < TM; ++t) {
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * khalf) * WTN + u * 32 + l31] = acc[t][u][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NRD; ++i) {
      const int idx = i * 64 + lane;
      const int row = idx / QPRW, cq = idx - row * QPRW;
      const float4 v = *reinterpret_cast<const float4*>(wl + row * WTN + cq * 4);
      const int m = m0 + wm * WTM + t * 32 + row;
      const int n = n0 + wn * WTN + cq * 4;
      if (m < g.K && n < g.Ktot) *reinterpret_cast<float4*>(out + (size_t)m * g.Ktot + n) = v;   // Ktot % 4 == 0
    }
    if (t + 1 < TM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (want_bias && m0 + tid < g.K) bias_partial[(size_t)split * g.K + m0 + tid] = bsum;
}

// ================================================================================================ //
// wgrad of stride-1 pad-1 3x3 convolutions in split-bf16, "row-tap" form.
// fast_wgrad_dma_kernel<.., MATH 1> is VALU-issue bound (14-22 VALU per MFMA: every wave splits every fragment,
// and for Cin = 64 the nine tap tiles each re-fetch and re-split the same dy pixels).  Here
//   * a K chunk is 16 output pixels of ONE image row, so (image, row, first column) are scalars;
//   * a block owns [BM output channels] x [one filter row kh, a slice of 64 input channels, ALL THREE kw]:
//     the three kw taps read the same input row shifted by one pixel, so the x operand is staged once as an
//     18-pixel segment and a lane builds its three B fragments from 10 gathered values (one split, two funnel
//     shifts) instead of 24; the dy operand is fetched and split once for the three taps.
//   18 MFMAs per wave and chunk for ~75 VALU (fast_wgrad_dma_kernel: 6 MFMAs for ~90).
// LDS: 3-slot ring of [16 px][BM] dy + [20 px][64] x (fp32, lane-linear LDS-DMA images); same split-K partial
// layout, reduce kernel and XCD mapping as the other wgrad kernels.
// ================================================================================================ //
// GROUPED launches (round 3): up to 4 weight gradients of the SAME shape in one launch.  The chip wants one full wave of blocks
// (768) whatever the number of convolutions behind it, so G problems run with nsplit / G splits each: the split-K partial
// tiles (the 2 x 76 MB per convolution that made the single launch move 2.5 x its algorithmic bytes), the end-of-kernel
// write burst and the reduce shrink by G, and every block's K loop gets G times longer.  Blocks [p * bpp, (p + 1) * bpp)
// serve problem p (bpp % 8 == 0 keeps a block's XCD = its split lane).
struct WgradBatch {
  const float* x[4];
  const float* dy[4];
  float* partial[4];
  float* bias_partial[4];
  int nprob, bpp;
};

template <int BM, int CIS, bool SPLIT = true, int ADDR = 1, bool PIPE = false>     // SPLIT false: one bf16 product per multiply (SRHIP_MATH_HALF)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PIPE ? 2 : 3))) void wgrad_rowtap_kernel(
    const float* __restrict__ x_, const float* __restrict__ dy_, float* __restrict__ partial_,
    float* __restrict__ bias_partial_, WgradGeom g, int nseg, int chunks_per_split, int tail_rem, WgradBatch bt) {
  const float* x = x_;
  const float* dy = dy_;
  float* partial = partial_;
  float* bias_partial = bias_partial_;
  int bid0 = blockIdx.x;
  if (bt.nprob > 1) {
    const int prob = __builtin_amdgcn_readfirstlane((int)blockIdx.x / bt.bpp);
    bid0 = (int)blockIdx.x - prob * bt.bpp;
    x = bt.x[prob];
    dy = bt.dy[prob];
    partial = bt.partial[prob];
    bias_partial = bt.bias_partial[prob];
  }
  // tail_rem > 0 ("paired tails", rows of 16 q + tail_rem pixels with tail_rem <= 8): the chunks are enumerated per PAIR of
  // image rows -- q full 16-pixel segments of row A, q of row B, then ONE chunk that holds both rows' tails: MFMA K index
  // k < 8 is pixel 16 q + k of row A, k >= 8 pixel 16 q + (k - 8) of row B (slots past the tail carry dy = 0).  Each half
  // is staged with its own halo (staged rows 0..9 / 10..19), so a lane's gather only swaps the base row of its K half
  // (8 khalf -> 10 khalf).  54-pixel rows: 7 chunks per two rows instead of 8 (an eighth of the MFMAs, DMAs and splits
  // of the 4 x 16 layout fell on padding).
  // tile = BM output channels x (one kh, CIS input channels, three kw); 128 x 64 for wide layers, 64 x 128 for Cout = 64
  constexpr int WM = BM / 64, WN = 4 / WM;          // a wave owns 64 co x (3 kw x 32 ci)
  constexpr int NCI = CIS / WN;
  static_assert((BM == 128 && CIS == 64) || (BM == 64 && CIS == 128), "tile shapes");
  static_assert(NCI == 32, "32 input channels per wave");
  constexpr int TM = 2, TN = 3;                     // TN = kw
  constexpr int A_B = BM * 64;                      // [16 px][BM] fp32
  constexpr int B_B = 20 * CIS * 4;                 // [20 px][CIS] fp32 (18 used)
  constexpr int STAGE_B = A_B + B_B;
  constexpr int NA = A_B / 1024 / 4;                // A pieces per wave
  constexpr int NBT = B_B / 1024;                   // B pieces per chunk, dealt to the waves in order
  constexpr int RPA = 1024 / (BM * 4), RPB = 1024 / (CIS * 4);   // pixel rows per piece
  constexpr int EPI_B = 4 * 32 * 32 * 4;
  constexpr int NSLOT = PIPE ? 5 : 3;               // PIPE: two blocks per CU (registers), so the ring can be five chunks deep
  constexpr int LDS_B = NSLOT * STAGE_B > EPI_B ? NSLOT * STAGE_B : EPI_B;
  __shared__ __attribute__((aligned(1024))) char lds[LDS_B];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ncs = g.C / CIS;                        // channel slices of the input
  const int ntn = 3 * ncs;                          // N tiles: (kh, ci slice)
  const int ntm = (g.K + BM - 1) / BM;
  int tile_n, tile_m, split;                        // XCD-aware order, as in fast_wgrad_kernel
  {
    const int tps = ntm * ntn;
    int bid = bid0;
    if (g.nsplit % 8 == 0) {
      const int j = bid >> 3;
      split = (j / tps) * 8 + (bid & 7);
      bid = j % tps;
    } else {
      split = bid / tps;
      bid -= split * tps;
    }
    tile_n = bid % ntn;
    tile_m = bid / ntn;
  }
  const int kh = tile_n / ncs, cs = tile_n - kh * ncs;
  const int m0 = tile_m * BM, ci_base = cs * CIS;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

  const int q = g.Wo >> 4;                          // full segments per row (paired-tails mode)
  const int cpp = 2 * q + 1;                        // chunks per row pair
  const int rows_total = g.N * g.Ho;
  const int nchunks_total = tail_rem > 0 ? ((rows_total + 1) >> 1) * cpp : g.N * g.Ho * nseg;
  const int c_begin = split * chunks_per_split;
  const int c_end = min(c_begin + chunks_per_split, nchunks_total);
  const int nk = c_end - c_begin;

  // DMA lanes.  A (dy): RPA pixel rows per 1 KiB piece, NA pieces per wave.  B (x): RPB staged pixel rows per piece,
  // NBT pieces dealt to the waves in order (wave-uniform counts nb)
  const int a_col = lane % (BM / 4), a_rsub = lane / (BM / 4);
  const bool a_colok = (m0 + a_col * 4) < g.K;
  const int b_col = lane % (CIS / 4), b_rsub = lane / (CIS / 4);
  const int nb = NBT / 4 + (wave < NBT % 4 ? 1 : 0);
  const int b_first = wave * (NBT / 4) + (wave < NBT % 4 ? wave : NBT % 4);
  const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + wave * NA * 1024);
  const unsigned b_dst = __builtin_amdgcn_readfirstlane(lds_base + A_B + b_first * 1024);

  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, g.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, g.dy_bytes, 0x00020000);
  int i_n, i_ho, i_seg;                             // chunk the next issue() fetches (scalars); paired tails: (i_n, i_ho) = row A of the pair, i_seg = index inside the pair
  if (tail_rem > 0) {
    i_seg = c_begin % cpp;
    const int r0 = (c_begin / cpp) * 2;
    i_ho = r0 % g.Ho;
    i_n = r0 / g.Ho;
  } else {
    i_seg = c_begin % nseg;
    const int t = c_begin / nseg;
    i_ho = t % g.Ho;
    i_n = t / g.Ho;
  }
  auto issue_tail = [&](int slot) {                 // both rows' tails: pixel slots 0..7 <- row A, 8..15 <- row B
    const int nB = i_ho + 1 < g.Ho ? i_n : i_n + 1, hoB = i_ho + 1 < g.Ho ? i_ho + 1 : 0;
    const int wo0 = q * 16;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int j = (wave * NA + i) * RPA + a_rsub;
      const int half = j >> 3, jj = j & 7;
      const int n_ = half ? nB : i_n, ho_ = half ? hoB : i_ho;
      const bool ok = a_colok && jj < tail_rem && n_ < g.N;
      const unsigned off = ok ? (unsigned)((((long)n_ * g.Ho + ho_) * g.Wo + wo0 + jj) * g.ldy + m0 + a_col * 4) * 4u : F_OOB;
      lds_dma16_buf(off, rs_y, a_dst + slot * STAGE_B + i * 1024);
    }
#pragma unroll
    for (int i = 0; i < NBT / 4 + 1; ++i) {
      if (i < nb) {
        const int r = (b_first + i) * RPB + b_rsub;            // staged row 0..19: half = r >= 10
        const int half = r >= 10 ? 1 : 0, ss = r - 10 * half;
        const int n_ = half ? nB : i_n, ho_ = half ? hoB : i_ho;
        const int hi = ho_ - 1 + kh, wi = wo0 - 1 + ss;
        const bool ok = n_ < g.N && hi >= 0 && hi < g.H && ss < tail_rem + 2 && wi < g.W;
        lds_dma16_buf(ok ? (unsigned)((((long)n_ * g.H + hi) * g.W + wi) * g.ldx + ci_base + b_col * 4) * 4u : F_OOB, rs_x, b_dst + slot * STAGE_B + i * 1024);
      }
    }
  };
  auto issue = [&](int slot) {
    if (tail_rem > 0) {
      if (i_seg == 2 * q) {
        issue_tail(slot);
        i_seg = 0;
        i_ho += 2;
        if (i_ho >= g.Ho) {
          i_ho -= g.Ho;
          ++i_n;
        }
        return;
      }
    }
    const bool rowB = tail_rem > 0 && i_seg >= q;
    const int c_n = rowB ? (i_ho + 1 < g.Ho ? i_n : i_n + 1) : i_n;
    const int c_ho = rowB ? (i_ho + 1 < g.Ho ? i_ho + 1 : 0) : i_ho;
    const bool rowlive = c_n < g.N;                  // odd row count: the last pair has no row B
    const int wo0 = (rowB ? i_seg - q : i_seg) * 16;
    const long prow = ((long)c_n * g.Ho + c_ho) * g.Wo + wo0;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int j = (wave * NA + i) * RPA + a_rsub;            // output pixel slot 0..15
      const unsigned off = (a_colok && rowlive && wo0 + j < g.Wo) ? (unsigned)((prow + j) * g.ldy + m0 + a_col * 4) * 4u : F_OOB;
      lds_dma16_buf(off, rs_y, a_dst + slot * STAGE_B + i * 1024);   // out of range => the hardware writes zeros
    }
    const int hi = c_ho - 1 + kh;
    const bool rowok = rowlive && hi >= 0 && hi < g.H;
    const long xrow = ((long)c_n * g.H + hi) * g.W;
#pragma unroll
    for (int i = 0; i < NBT / 4 + 1; ++i) {
      if (i < nb) {
        const int r = (b_first + i) * RPB + b_rsub;            // staged pixel row: input column wo0 - 1 + r
        const int wi = wo0 - 1 + r;
        const bool ok = rowok && r < 18 && wi >= 0 && wi < g.W;
        lds_dma16_buf(ok ? (unsigned)((xrow + wi) * g.ldx + ci_base + b_col * 4) * 4u : F_OOB, rs_x, b_dst + slot * STAGE_B + i * 1024);
      }
    }
    if (tail_rem > 0) {
      ++i_seg;                                      // the tail chunk (i_seg == 2 q) closes the pair
    } else if (++i_seg == nseg) {
      i_seg = 0;
      if (++i_ho == g.Ho) {
        i_ho = 0;
        ++i_n;
      }
    }
  };
  // ADDR 1 (round 4): the same DMAs with the chunk's position as the instruction's SCALAR offset.  The form above rebuilds every
  // lane's byte offset per chunk (64-bit pixel arithmetic, selects lowered to exec-mask branches: ~90 VALU instructions and 16
  // branches per chunk and wave in front of the 18 MFMAs).  Here a lane keeps constant offsets relative to the chunk's first
  // pixel, the chunk's first pixel goes into the buffer instruction's soffset (tensors < 2 GiB, so offset + soffset cannot wrap
  // and a dead lane's 0x80000000 stays out of range whichever way the range check counts soffset), and validity is one
  // compare against a scalar limit.  The x descriptor starts one image row + one pixel BEFORE the tensor so that the halo
  // pixel (-1) of row -1 has offset 0; lanes that would read there are dead lanes.  Same bytes into the same LDS places.
  const unsigned x_shift = (unsigned)(g.W + 1) * (unsigned)g.ldx * 4u;
  __amdgpu_buffer_rsrc_t rs_xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) - (size_t)(g.W + 1) * g.ldx, 0, g.x_bytes + x_shift, 0x00020000);
  unsigned a_full[NA], a_tailv[NA], a_tailA[NA];     // dy: full chunk / tail chunk (both rows live) / tail chunk (row B dead)
  int a_j[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int j = (wave * NA + i) * RPA + a_rsub;
    a_j[i] = j;
    a_full[i] = a_colok ? (unsigned)(j * g.ldy + m0 + a_col * 4) * 4u : F_OOB;
    const int half = j >> 3, jj = j & 7;
    a_tailv[i] = (a_colok && jj < tail_rem) ? (unsigned)((half * g.Wo + jj) * g.ldy + m0 + a_col * 4) * 4u : F_OOB;
    a_tailA[i] = half ? F_OOB : a_tailv[i];
  }
  constexpr int NBI = NBT / 4 + 1;
  unsigned b_full[NBI], b_tailv[NBI];
  int b_r[NBI], b_half[NBI];
#pragma unroll
  for (int i = 0; i < NBI; ++i) {
    const int r = (b_first + i) * RPB + b_rsub;
    b_r[i] = r;
    b_full[i] = r < 18 ? (unsigned)(r * g.ldx + ci_base + b_col * 4) * 4u : F_OOB;
    const int half = r >= 10 ? 1 : 0, ss = r - 10 * half;
    b_half[i] = half;
    const int wo0t = (g.Wo >> 4) * 16;
    const bool okss = r < 20 && ss <= tail_rem && wo0t - 1 + ss >= 0;            // input column wo0 - 1 + ss inside the row
    b_tailv[i] = okss ? (unsigned)((half * g.W + ss) * g.ldx + ci_base + b_col * 4) * 4u : F_OOB;
  }
  int i_row = 0;                                    // flat output row (image * Ho + row) of the chunk the next issue fetches (row A of a pair)
  if (ADDR == 1) i_row = i_n * g.Ho + i_ho;
  const int rows_all = g.N * g.Ho;
  auto dma_s = [&](unsigned voff, unsigned soff, __amdgpu_buffer_rsrc_t r, unsigned dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(dst) : "memory");
  };
  auto issue_s = [&](int slot) {
    const unsigned adst = a_dst + slot * STAGE_B, bdst = b_dst + slot * STAGE_B;
    if (tail_rem > 0 && i_seg == 2 * q) {           // both rows' tails
      const int wo0 = q * 16;
      const bool liveB = i_row + 1 < rows_all;
      const int hoB = i_ho + 1 < g.Ho ? i_ho + 1 : 0;
      const int hiA = i_ho - 1 + kh, hiB = hoB - 1 + kh;
      const bool okA = hiA >= 0 && hiA < g.H, okB = liveB && hiB >= 0 && hiB < g.H;
      const unsigned sa = (unsigned)(i_row * g.Wo + wo0) * (unsigned)g.ldy * 4u;
      const unsigned sb = (unsigned)((i_row + kh) * g.W + wo0) * (unsigned)g.ldx * 4u;
#pragma unroll
      for (int i = 0; i < NA; ++i) dma_s(liveB ? a_tailv[i] : a_tailA[i], sa, rs_y, adst + i * 1024);
#pragma unroll
      for (int i = 0; i < NBI; ++i)
        if (i < nb) dma_s((b_half[i] ? okB : okA) ? b_tailv[i] : F_OOB, sb, rs_xs, bdst + i * 1024);
      i_seg = 0;
      i_row += 2;
      i_ho += 2;
      if (i_ho >= g.Ho) i_ho -= g.Ho;
      return;
    }
    const int rowB = (tail_rem > 0 && i_seg >= q) ? 1 : 0;
    const int c_row = i_row + rowB;
    int c_ho = i_ho + rowB;
    if (c_ho >= g.Ho) c_ho -= g.Ho;
    const bool rowlive = c_row < rows_all;
    const int wo0 = (rowB ? i_seg - q : i_seg) * 16;
    const int hi = c_ho - 1 + kh;
    const bool rowok = rowlive && hi >= 0 && hi < g.H;
    const unsigned sa = rowlive ? (unsigned)(c_row * g.Wo + wo0) * (unsigned)g.ldy * 4u : 0u;
    const unsigned sb = rowok ? (unsigned)((c_row + kh) * g.W + wo0) * (unsigned)g.ldx * 4u : 0u;
    const int lim_a = rowlive ? g.Wo - wo0 : 0;                       // pixel slots j < lim_a are inside the row
    const int lo_b = rowok ? 1 - wo0 : 64;                            // staged rows lo_b <= r < hi_b are inside the input row
    const int hi_b = g.W + 1 - wo0;
#pragma unroll
    for (int i = 0; i < NA; ++i) dma_s(a_j[i] < lim_a ? a_full[i] : F_OOB, sa, rs_y, adst + i * 1024);
#pragma unroll
    for (int i = 0; i < NBI; ++i)
      if (i < nb) dma_s((b_r[i] >= lo_b && b_r[i] < hi_b) ? b_full[i] : F_OOB, sb, rs_xs, bdst + i * 1024);
    if (tail_rem > 0) {
      ++i_seg;
    } else if (++i_seg == nseg) {
      i_seg = 0;
      ++i_row;
      if (++i_ho == g.Ho) i_ho = 0;
    }
  };
  auto wait_chunk = [&](bool more) {                // this wave's pieces of the oldest chunk in flight have landed
    if (!more) {
      wait_vmcnt<0>();
    } else if (nb == NBT / 4 + 1) {
      wait_vmcnt<NA + NBT / 4 + 1>();
    } else {
      wait_vmcnt<NA + NBT / 4>();
    }
  };

  const int wm = wave / WN, wn = wave - wm * WN;
  const int khalf = lane >> 5, l31 = lane & 31;
  // fragment gather offsets inside a stage (bytes): A value j of tile t = raw A[(8 khalf + j)][wm*64 + t*32 + l31];
  // B value j (0..9) = staged row 8 khalf + j, channel wn*32 + l31
  const int a_off = (8 * khalf) * BM * 4 + (wm * 64 + l31) * 4;
  const int b_off_full = A_B + (8 * khalf) * CIS * 4 + (wn * NCI + l31) * 4;
  const int b_off_tail = A_B + (10 * khalf) * CIS * 4 + (wn * NCI + l31) * 4;
  int c_sub = tail_rem > 0 ? c_begin % cpp : 0;     // compute side: position of the current chunk inside its row pair
  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  float bsum = 0.f;
  const bool want_bias = bias_partial != nullptr && tile_n == 0 && tid < BM;

  auto issue_any = [&](int slot) {
    if (ADDR == 1) issue_s(slot);
    else issue(slot);
  };
  struct Frags {
    bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
  };
  // raw fp32 stage -> this wave's split fragments of one chunk (+ the bias column sum, taken from the raw dy rows)
  auto convert = [&](int slot, Frags& f) {
    const char* sb = lds + slot * STAGE_B;
    const int b_off = (tail_rem > 0 && c_sub == 2 * q) ? b_off_tail : b_off_full;
    if (tail_rem > 0) c_sub = c_sub == 2 * q ? 0 : c_sub + 1;
    // A fragments: gather 8 pixels, split
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float*>(sb + a_off + j * BM * 4 + t * 128);
      split_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), f.ah[t], f.al[t]);
    }
    // B: 10 staged pixels -> packed hi/lo pairs P0..P4 -> the three kw fragments (kw 1 by a 16-bit funnel shift)
    unsigned ph[5], pl[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const float e0 = *reinterpret_cast<const float*>(sb + b_off + (2 * i) * CIS * 4);
      const float e1 = *reinterpret_cast<const float*>(sb + b_off + (2 * i + 1) * CIS * 4);
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      const bf16x2_t h = {(__bf16)e0, (__bf16)e1};
      ph[i] = __builtin_bit_cast(unsigned, h);
      const bf16x2_t l = {(__bf16)(e0 - __uint_as_float(ph[i] << 16)), (__bf16)(e1 - __uint_as_float(ph[i] & 0xffff0000u))};
      pl[i] = __builtin_bit_cast(unsigned, l);
    }
    {
      const u32x4 h0 = {ph[0], ph[1], ph[2], ph[3]}, l0 = {pl[0], pl[1], pl[2], pl[3]};
      const u32x4 h2 = {ph[1], ph[2], ph[3], ph[4]}, l2 = {pl[1], pl[2], pl[3], pl[4]};
      const u32x4 h1 = {__builtin_amdgcn_alignbit(ph[1], ph[0], 16), __builtin_amdgcn_alignbit(ph[2], ph[1], 16),
                        __builtin_amdgcn_alignbit(ph[3], ph[2], 16), __builtin_amdgcn_alignbit(ph[4], ph[3], 16)};
      const u32x4 l1 = {__builtin_amdgcn_alignbit(pl[1], pl[0], 16), __builtin_amdgcn_alignbit(pl[2], pl[1], 16),
                        __builtin_amdgcn_alignbit(pl[3], pl[2], 16), __builtin_amdgcn_alignbit(pl[4], pl[3], 16)};
      f.bh[0] = __builtin_bit_cast(bf16x8_t, h0); f.bl[0] = __builtin_bit_cast(bf16x8_t, l0);
      f.bh[1] = __builtin_bit_cast(bf16x8_t, h1); f.bl[1] = __builtin_bit_cast(bf16x8_t, l1);
      f.bh[2] = __builtin_bit_cast(bf16x8_t, h2); f.bl[2] = __builtin_bit_cast(bf16x8_t, l2);
    }
    if (want_bias) {
      const float* col = reinterpret_cast<const float*>(sb) + tid;
#pragma unroll
      for (int r = 0; r < 16; ++r) bsum += col[r * BM];
    }
  };
  auto mfma_all = [&](const Frags& f) {
#pragma unroll
    for (int i = 0; i < (SPLIT ? 3 : 1) * TM * TN; ++i) {
      const int grp = SPLIT ? i / (TM * TN) : 2, t = (i % (TM * TN)) / TN, u = i % TN;
      acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(grp == 0 ? f.al[t] : f.ah[t], grp == 1 ? f.bl[u] : f.bh[u], acc[t][u], 0, 0, 0);
    }
  };
  if constexpr (!PIPE) {
    if (nk > 0) {
      issue_any(0);
      if (nk > 1) issue_any(1);
      int stage = 0, nstage = 2;
      for (int kc = 0; kc < nk; ++kc) {
        wait_chunk(kc + 1 < nk);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kc + 2 < nk) issue_any(nstage);
        Frags f;
        convert(stage, f);
        mfma_all(f);
        stage = stage == 2 ? 0 : stage + 1;
        nstage = nstage == 2 ? 0 : nstage + 1;
      }
    }
  } else {
    // PIPE (round 4): the conversion of chunk k + 1 and the MFMAs of chunk k are independent instruction streams of one loop body.
    // In the form above a wave alternates a ~90-instruction gather / split phase with 18 back-to-back MFMAs, and the co-resident
    // waves of a SIMD (same code, started together, re-synchronised by every stall) do the same phases at the same time: the
    // ablation of round 2 found the parts ADDING UP (MFMAs 37, conversion 24, DMA 7, barrier 4.5 of 123 us).  Holding the next
    // chunk's fragments costs 40 registers: two waves per SIMD instead of three, which pays for a five-deep operand ring.
    constexpr int D = NSLOT - 1;                    // chunks in flight
    const int per = NA + nb;                        // DMAs of this wave per chunk
    auto wait_after = [&](int chunks) {             // all but the newest `chunks` chunks of this wave's DMAs have landed
      const int n = chunks * per;
      switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 12: wait_vmcnt<12>(); break;
        default: wait_vmcnt<0>(); break;
      }
    };
    static_assert(NA + NBT / 4 == 3, "wait_after's cases assume 3 or 4 DMAs per wave and chunk");
    if (nk > 0) {
      int issued = 0;
      for (; issued < D && issued < nk; ++issued) issue_any(issued);
      wait_after(issued - 1);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      int islot = D;
      if (issued < nk) {
        issue_any(islot);
        ++issued;
        islot = 0;
      }
      Frags f;
      convert(0, f);
      int cslot = 1;
      for (int kc = 0; kc + 1 < nk; ++kc) {
        wait_after(issued - (kc + 2));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (issued < nk) {
          issue_any(islot);
          ++issued;
          islot = islot == NSLOT - 1 ? 0 : islot + 1;
        }
        Frags gnext;
        convert(cslot, gnext);
        mfma_all(f);
        f = gnext;
        cslot = cslot == NSLOT - 1 ? 0 : cslot + 1;
      }
      mfma_all(f);
    }
  }

  // partial tile -> LDS (wave-private 32x32 region) -> row-contiguous 16-byte stores; sub-tile u is tap (kh, kw = u)
  float* out = partial + (size_t)split * g.K * g.Ktot;
  __syncthreads();
  float* wl = reinterpret_cast<float*>(lds) + wave * (32 * 32);
#pragma unroll
  for (int t = 0; t < TM; ++t) {
#pragma unroll
    for (int u = 0; u < TN; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * khalf) * 32 + l31] = acc[t][u][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int idx = i * 64 + lane;
        const int row = idx >> 3, cq = idx & 7;
        const float4 v = *reinterpret_cast<const float4*>(wl + row * 32 + cq * 4);
        const int m = m0 + wm * 64 + t * 32 + row;
        const int n = (kh * 3 + u) * g.C + ci_base + wn * NCI + cq * 4;
        if (m < g.K) *reinterpret_cast<float4*>(out + (size_t)m * g.Ktot + n) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (want_bias && m0 + tid < g.K) bias_partial[(size_t)split * g.K + m0 + tid] = bsum;
}

// partial[s][co][(tap,ci)] --sum over s--> dw[co][ci][kh][kw];  bias_partial[s][co] --> db[co]
// 64 outputs per block x SUB split lanes; each lane keeps 4 loads in flight, LDS combines the lanes in a fixed order.
// SUB = 16 for the one- and two-tile GEMMs (1x1 and 64 -> 64 convs: up to 768 splits of a 16 KB tile, only 65 blocks):
// with 4 lanes a thread walked 192 dependent-latency loads (30 us per call, 100 calls per step).
template <int SUB>
__global__ __launch_bounds__(64 * SUB) void fast_wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                     const float* __restrict__ bias_partial,
                                                                     float* __restrict__ dw, float* __restrict__ db,
                                                                     int nsplit, int cout, int cin, int khkw, int ktot,
                                                                     int accumulate) {
  __shared__ float red[64 * SUB];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
  const int total = cout * ktot;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < total) {
    int i = sub;
    for (; i + 3 * SUB < nsplit; i += 4 * SUB) {
      s0 += partial[(size_t)(i + 0 * SUB) * total + e];
      s1 += partial[(size_t)(i + 1 * SUB) * total + e];
      s2 += partial[(size_t)(i + 2 * SUB) * total + e];
      s3 += partial[(size_t)(i + 3 * SUB) * total + e];
    }
    for (; i < nsplit; i += SUB) s0 += partial[(size_t)i * total + e];
  } else if (db != nullptr && e < total + cout) {
    const int co = e - total;
    for (int i = sub; i < nsplit; i += SUB) s0 += bias_partial[(size_t)i * cout + co];
  }
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sub == 0) {
    const int t = threadIdx.x;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < SUB; j += 4) v += (red[t + 64 * j] + red[t + 64 * (j + 1)]) + (red[t + 64 * (j + 2)] + red[t + 64 * (j + 3)]);
    if (e < total) {
      const int co = e / ktot, kcol = e - co * ktot;
      const int tap = kcol / cin, ci = kcol - tap * cin;
      float* o = dw + ((size_t)co * cin + ci) * khkw + tap;
      *o = accumulate ? *o + v : v;
    } else if (db != nullptr && e < total + cout) {
      db[e - total] = accumulate ? db[e - total] + v : v;
    }
  }
}

// The same reduction, four consecutive outputs per thread (16-byte loads of the partial rows) and up to four problems of one
// shape per launch (blockIdx.y): round 4.  The summation order of every output element is EXACTLY the scalar kernel's (split lane
// sub sums i = sub, sub + SUB, ... round-robin into four accumulators, (s0 + s1) + (s2 + s3), then the SUB lanes in groups of
// four), so results are bit-identical to it; what changes is the access width (dword loads ran the 38 MB of a RAB conv's
// partials at 3.2 TB/s) and one launch per grouped weight gradient instead of one per convolution.
struct ReduceBatch {
  const float* partial[4];
  const float* bias_partial[4];
  float* dw[4];
  float* db[4];
};
__device__ __forceinline__ float4 add4(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
template <int SUB>
__global__ __launch_bounds__(64 * SUB) void fast_wgrad_reduce4_kernel(ReduceBatch rb, int nsplit, int cout, int cin, int khkw,
                                                                      int ktot, int accumulate, int nbias) {   // nbias: rows of bias_partial (= nsplit but for conv_wgrad_flat.hip's flat8 kernel)
  __shared__ float4 red[64 * SUB];
  const int prob = blockIdx.y;
  const float* __restrict__ partial = rb.partial[prob];
  const float* __restrict__ bias_partial = rb.bias_partial[prob];
  float* __restrict__ dw = rb.dw[prob];
  float* __restrict__ db = rb.db[prob];
  const int e = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, sub = threadIdx.x >> 6;
  const int total = cout * ktot;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 s0 = z, s1 = z, s2 = z, s3 = z;
  if (e < total) {
    int i = sub;
    for (; i + 3 * SUB < nsplit; i += 4 * SUB) {
      s0 = add4(s0, *reinterpret_cast<const float4*>(partial + (size_t)(i + 0 * SUB) * total + e));
      s1 = add4(s1, *reinterpret_cast<const float4*>(partial + (size_t)(i + 1 * SUB) * total + e));
      s2 = add4(s2, *reinterpret_cast<const float4*>(partial + (size_t)(i + 2 * SUB) * total + e));
      s3 = add4(s3, *reinterpret_cast<const float4*>(partial + (size_t)(i + 3 * SUB) * total + e));
    }
    for (; i < nsplit; i += SUB) s0 = add4(s0, *reinterpret_cast<const float4*>(partial + (size_t)i * total + e));
  } else if (db != nullptr && e < total + cout) {
    const int co = e - total;
    for (int i = sub; i < nbias; i += SUB) s0 = add4(s0, *reinterpret_cast<const float4*>(bias_partial + (size_t)i * cout + co));
  }
  red[threadIdx.x] = add4(add4(s0, s1), add4(s2, s3));
  __syncthreads();
  if (sub == 0) {
    const int t = threadIdx.x;
    float4 v = z;
#pragma unroll
    for (int j = 0; j < SUB; j += 4) v = add4(v, add4(add4(red[t + 64 * j], red[t + 64 * (j + 1)]), add4(red[t + 64 * (j + 2)], red[t + 64 * (j + 3)])));
    const float vv[4] = {v.x, v.y, v.z, v.w};
    if (e < total) {
      const int co = e / ktot, kcol = e - co * ktot;              // ktot % 4 == 0: the four outputs share co and the tap
      const int tap = kcol / cin, ci = kcol - tap * cin;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* o = dw + ((size_t)co * cin + ci + j) * khkw + tap;
        *o = accumulate ? *o + vv[j] : vv[j];
      }
    } else if (db != nullptr && e < total + cout) {
#pragma unroll
      for (int j = 0; j < 4; ++j) db[e - total + j] = accumulate ? db[e - total + j] + vv[j] : vv[j];
    }
  }
}
// one launch for nprob problems of one shape; false: the caller takes the scalar kernel (shapes / pointers the 16-byte form cannot serve)
static bool launch_reduce4(int nprob, const float* const* partial, const float* const* bias_partial, float* const* dw, float* const* db,
                           int nsplit, int cout, int cin, int khkw, int ktot, int accumulate, hipStream_t st, int nbias = -1) {
  if (nbias < 0) nbias = nsplit;
  extern int g_wgrad_cfg;
  if (g_wgrad_cfg == 8 || nprob < 1 || nprob > 4 || ktot % 4 != 0 || cout % 4 != 0 || cin % 4 != 0) return false;
  ReduceBatch rb;
  bool anydb = false;
  for (int k = 0; k < 4; ++k) {
    const int j = k < nprob ? k : 0;
    rb.partial[k] = partial[j];
    rb.bias_partial[k] = bias_partial ? bias_partial[j] : nullptr;
    rb.dw[k] = dw[j];
    rb.db[k] = db ? db[j] : nullptr;
    if (((uintptr_t)rb.partial[k] | (uintptr_t)rb.bias_partial[k]) & 15) return false;
    anydb = anydb || rb.db[k] != nullptr;
  }
  const long total = (long)cout * ktot + (anydb ? cout : 0);
  if (nsplit >= 64)
    hipLaunchKernelGGL(fast_wgrad_reduce4_kernel<16>, dim3(cdiv(total, 256), nprob), dim3(1024), 0, st, rb, nsplit, cout, cin, khkw, ktot, accumulate, nbias);
  else
    hipLaunchKernelGGL(fast_wgrad_reduce4_kernel<4>, dim3(cdiv(total, 256), nprob), dim3(256), 0, st, rb, nsplit, cout, cin, khkw, ktot, accumulate, nbias);
  return true;
}

bool launch_reduce4_shared(int nprob, const float* const* partial, const float* const* bias_partial, float* const* dw, float* const* db,
                           int nsplit, int cout, int cin, int khkw, int ktot, int accumulate, hipStream_t st, int nbias) {
  return launch_reduce4(nprob, partial, bias_partial, dw, db, nsplit, cout, cin, khkw, ktot, accumulate, st, nbias);   // conv_wgrad_flat.hip
}

// ================================================================================================ //
// Weight + bias gradient of the attention tail's 1x1 conv (64 -> 64) with its operand scales (round 4):
//     dWc[co][ci] = sum_p g[p][co] * (m[p] * s[b(p)][ci] * u[p][ci]),   dbc[co] = sum_p g[p][co]
// The generic register-staged kernel took 24.5 + 7.9 us at B = 32 for 48 MB of operands (48 launches per step).  Here the fp32
// MFMA 32x32x2 does the contraction over PIXELS directly: with K = 2 per instruction a lane holds ONE k value, so both operands
// are read as they lie in memory -- lanes 0-31 take pixel p, lanes 32-63 pixel p + 1, 32 consecutive channels each, no transpose
// and no LDS.  A block owns a pixel range of ONE image (s is factored out and applied once per block), a wave walks pixel pairs
// with four accumulator tiles (co halves x ci halves); the four waves' tiles are summed through LDS in a fixed order and leave as
// one split-K partial for fast_wgrad_reduce4_kernel.  Exact fp32 products like the kernel it replaces.
// ================================================================================================ //
__global__ __launch_bounds__(256) void wgrad_1x1_scaled_kernel(const float* __restrict__ u, const float* __restrict__ gy,
                                                               const float* __restrict__ m, const float* __restrict__ sc,
                                                               float* __restrict__ partial, float* __restrict__ bias_partial, int hw,
                                                               int per, int sp) {
  __shared__ float red[4][64 * 64];
  __shared__ float bred[4][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, kh = lane >> 5;
  const int b = blockIdx.x / sp, j = blockIdx.x - b * sp;
  const int p_begin = j * per, p_end = min(p_begin + per, hw);
  const size_t base = (size_t)b * hw;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;
  float bs0 = 0.f, bs1 = 0.f;
  constexpr int UN = 8;                               // pixel pairs per batch; the next batch's loads are issued before this one's MFMAs
  float g0[2][UN], g1[2][UN], x0[2][UN], x1[2][UN], mv[2][UN];
  auto fetch = [&](int set, int pp) {
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      const int p = pp + 8 * i + kh;
      const bool ok = p < p_end;
      const size_t o = (base + (ok ? p : p_begin)) * 64 + l31;
      g0[set][i] = ok ? gy[o] : 0.f;
      g1[set][i] = ok ? gy[o + 32] : 0.f;
      x0[set][i] = ok ? u[o] : 0.f;
      x1[set][i] = ok ? u[o + 32] : 0.f;
      mv[set][i] = ok ? m[base + p] : 0.f;
    }
  };
  auto compute = [&](int set) {
#pragma unroll
    for (int i = 0; i < UN; ++i) {
      bs0 += g0[set][i];
      bs1 += g1[set][i];
      const float a0 = g0[set][i] * mv[set][i], a1 = g1[set][i] * mv[set][i];
      acc[0][0] = mfma32f(a0, x0[set][i], acc[0][0]);
      acc[0][1] = mfma32f(a0, x1[set][i], acc[0][1]);
      acc[1][0] = mfma32f(a1, x0[set][i], acc[1][0]);
      acc[1][1] = mfma32f(a1, x1[set][i], acc[1][1]);
    }
  };
  int pp = p_begin + 2 * wave;
  if (pp < p_end) {
    fetch(0, pp);
    for (;;) {                                        // two batches per trip: the register sets are indexed at compile time
      const int pn = pp + 8 * UN;
      if (pn < p_end) fetch(1, pn);
      compute(0);
      if (pn >= p_end) break;
      const int pn2 = pn + 8 * UN;
      if (pn2 < p_end) fetch(0, pn2);
      compute(1);
      if (pn2 >= p_end) break;
      pp = pn2;
    }
  }
  // this wave's 64 x 64 tile -> LDS [co][ci]; C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave][(a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 64 + c * 32 + l31] = acc[a][c][r];
  bred[wave][kh][l31] = bs0;
  bred[wave][kh][32 + l31] = bs1;
  __syncthreads();
  float* out = partial + (size_t)blockIdx.x * 64 * 64;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = i * 256 + tid;                      // co = e / 64, ci = e % 64
    const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    out[e] = v * sc[b * 64 + (e & 63)];
  }
  if (bias_partial != nullptr && tid < 64)
    bias_partial[(size_t)blockIdx.x * 64 + tid] = ((bred[0][0][tid] + bred[0][1][tid]) + (bred[1][0][tid] + bred[1][1][tid])) +
                                                 ((bred[2][0][tid] + bred[2][1][tid]) + (bred[3][0][tid] + bred[3][1][tid]));
}

// OIHW -> n-major packed GEMM operand.
// mode 0 (fprop): P[co][(kh*KW+kw)*Cin + ci]  = w[co][ci][kh][kw]
// mode 1 (dgrad): P[ci][(kh*KW+kw)*Cout + co] = w[co][ci][kh][kw]
__global__ void fast_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int cout, int cin, int kh,
                                 int kw, int mode) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = cout * cin * kh * kw;
  if (idx >= total) return;
  const int khkw = kh * kw;
  if (mode == 0) {
    const int co = idx / (khkw * cin);
    const int rem = idx - co * khkw * cin;
    const int tap = rem / cin, ci = rem - tap * cin;
    fast_pack_store(packed, total, idx, w[((size_t)co * cin + ci) * khkw + tap], cout, cin, khkw);
  } else {
    const int ci = idx / (khkw * cout);
    const int rem = idx - ci * khkw * cout;
    const int tap = rem / cout, co = rem - tap * cout;
    fast_pack_store(packed, total, idx, w[((size_t)co * cin + ci) * khkw + tap], cin, cout, khkw);
  }
}

// ================================================================================================ //
// host side
// ================================================================================================ //
static bool shape_ok(int csrc, int kh, int kw) { return csrc % 16 == 0 && kh * kw <= 32 && kh == kw; }
bool fast_fwd_ok(int cin, int cout, int kh, int kw) { return shape_ok(cin, kh, kw); }
bool fast_dgrad_ok(int cin, int cout, int kh, int kw) { return shape_ok(cout, kh, kw); }
bool fast_wgrad_ok(int cin, int cout, int kh, int kw) { return cin % 16 == 0 && cout % 4 == 0 && kh == kw; }

int fast_pack_weight(const float* w, float* packed, int cout, int cin, int kh, int kw, int mode, hipStream_t st) {
  const long total = (long)cout * cin * kh * kw;
  hipLaunchKernelGGL(fast_pack_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, w, packed, cout, cin, kh, kw, mode);
  return check_launch("fast_pack_weight");
}

thread_local Dst2Request g_dst2_req;
thread_local ResRequest g_res_req;
thread_local PhaseRequest g_phase_req;
int g_conv_math = 0;     // SRHIP_MATH_*: 0 exact fp32 MFMA; 1 split-bf16 x3 MFMA; 2 one 16-bit product (fp16 activations / bf16 gradients)
int g_fast_dynlds = 0;   // experiment knob (srhip_debug_set(2, bytes)): extra dynamic LDS per block = occupancy limiter
template <int BM, int BN, int WM, int WN, int BK>
static int launch_fast(const float* src, const float* wt, const float* bias, const float* residual,
                       const float* rowscale, const float* chanscale, const float* actmask, float* dst,
                       const FastGeom& g, hipStream_t st) {
  const int nbm = cdiv(g.M, BM), nbn = cdiv(g.K, BN);
  hipLaunchKernelGGL((fast_conv_kernel<BM, BN, WM, WN, BK>), dim3(nbm * nbn), dim3(WM * WN * 64), g_fast_dynlds, st, src, wt, bias,
                     residual, rowscale, chanscale, actmask, dst, g, nbm, nbn);
  return check_launch("fast_conv");
}

// ================================================================================================ //
// ONE destination channel on a small grid (the discriminator's head conv 512 -> 1 at 14 x 14: 6272 output pixels, 4608
// multiply-adds each).  The MFMA kernels pad the channel to a 32-wide tile and have 49 blocks to offer, each walking 288
// K chunks serially: 230 us for 58 MFLOP, three times per step in the discriminator's serial chain.  Here a wave owns an
// output pixel: lanes stride the source channels with 16-byte loads (x row and weight row are both contiguous), fp32
// FMAs, one butterfly reduction.  Exact fp32 in every arithmetic mode.
// ================================================================================================ //
__global__ __launch_bounds__(256) void dot_conv_kernel(const float* __restrict__ src, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, float* __restrict__ dst, FastGeom g) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= g.M) return;
  const int ow = m % g.OW, t = m / g.OW;
  const int oh = t % g.OH, n = t / g.OH;
  float acc = 0.f;
  for (int th = 0; th < g.TH; ++th) {
    const int sh = oh * g.ss + g.dh0 + th * g.dhs;
    if (sh < 0 || sh >= g.Hs) continue;
    for (int tw = 0; tw < g.TW; ++tw) {
      const int sw = ow * g.ss + g.dw0 + tw * g.dws;
      if (sw < 0 || sw >= g.Ws) continue;
      const float* xp = src + ((size_t)(n * g.Hs + sh) * g.Ws + sw) * g.lds;
      const float* wp = wt + (size_t)((g.kh0 + th * g.khs) * g.KW + (g.kw0 + tw * g.kws)) * g.C;
      for (int c = lane * 4; c < g.C; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(xp + c);
        const float4 b = *reinterpret_cast<const float4*>(wp + c);
        acc = fmaf(a.x, b.x, acc);
        acc = fmaf(a.y, b.y, acc);
        acc = fmaf(a.z, b.z, acc);
        acc = fmaf(a.w, b.w, acc);
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (lane == 0) {
    float v = acc;
    if (g.flags & SRHIP_EPI_BIAS) v += bias[0];
    if (g.flags & SRHIP_EPI_LRELU) v = v > 0.f ? v : v * g.slope;
    dst[((size_t)(n * g.Hd + oh * g.dsd + g.ph) * g.Wd + ow * g.dsd + g.pw) * g.ldd] = v;
  }
}

// experiment knob (srhip_debug_set(0, cfg)): 0 = heuristic below
int g_patch_ks = 1;    // srhip_debug_set(10, v): 0 = conv_patch_kernel<64> instead of conv_patch_ks_kernel (bit-identical to the DMA kernel; A/B and pinned tests)
int g_fast_cfg = 0;
int g_phase_batch = 1;   // srhip_debug_set(17, v): 1 = the phases of a stride-2 data gradient as one launch of the LDS-DMA kernel

// Patch shape for conv_patch_kernel: PH x PW output pixels per block (<= 128), halo patch <= 192 rows = 12 DMA pieces; picks the
// shape that wastes the fewest of the 128 GEMM rows over the whole image (halo patch <= 192 rows = 12 DMA pieces).  Only stride-1 3x3 geometries.
static bool plan_patch(const FastGeom& g, PatchGeom* pg, double min_eff = 0.70) {
  if (g.TH != 3 || g.TW != 3 || g.ss != 1 || g.dsd != 1 || g.ph != 0 || g.pw != 0) return false;
  if ((g.dhs != 1 && g.dhs != -1) || (g.dws != 1 && g.dws != -1)) return false;
  if (g.Hd != g.OH || g.Wd != g.OW) return false;
  double best = 0.0;
  for (int pw = 4; pw <= 64 && pw <= g.OW + 3; ++pw) {
    int ph = 128 / pw;
    if (ph > g.OH) ph = g.OH;
    if (ph < 1 || (ph + 2) * (pw + 2) > 192) continue;
    const long tiles = (long)cdiv(g.OH, ph) * cdiv(g.OW, pw);
    // among equally efficient shapes prefer the one whose rows are mostly whole 16-pixel runs: those lane groups read
    // the patch without LDS bank conflicts (patch_pixel)
    const double eff = (double)g.OH * g.OW / ((double)tiles * 128.0) + 1e-6 * (double)(pw - pw % 16) / pw;
    if (eff > best + 1e-9) {
      best = eff;
      pg->PH = ph; pg->PW = pw;
    }
  }
  if (best < min_eff || best <= 0.0) return false;
  pg->tiles_h = cdiv(g.OH, pg->PH);
  pg->tiles_w = cdiv(g.OW, pg->PW);
  pg->PWP = pg->PW + 2;
  pg->PR = (pg->PH + 2) * pg->PWP;
  pg->npieces = cdiv(pg->PR, 16);
  pg->lo_h = g.dhs > 0 ? g.dh0 : g.dh0 + 2 * g.dhs;
  pg->lo_w = g.dws > 0 ? g.dw0 : g.dw0 + 2 * g.dws;
  // lane-row group order (patch_pixel): pairs of full 16-pixel runs whose first patch rows are congruent mod 16 first,
  // then the runs without a partner, then the left-over / dead groups
  const int a = pg->PW >> 4;
  const int nfull = pg->PH * a < 8 ? pg->PH * a : 8;
  auto base = [&](int q) { return ((q / a) * pg->PWP + (q % a) * 16) & 15; };
  int fin[8], nf = 0, singles[8], ns = 0;
  bool used[8] = {false, false, false, false, false, false, false, false};
  for (int q = 0; q < nfull; ++q) {
    if (used[q]) continue;
    used[q] = true;
    int partner = -1;
    for (int p2 = q + 1; p2 < nfull; ++p2)
      if (!used[p2] && base(p2) == base(q)) {
        partner = p2;
        break;
      }
    if (partner >= 0) {
      used[partner] = true;
      fin[nf++] = q;
      fin[nf++] = partner;
    } else {
      singles[ns++] = q;
    }
  }
  for (int i = 0; i < ns; ++i) fin[nf++] = singles[i];
  for (int q = nfull; q < 8; ++q) fin[nf++] = q;
  pg->gmap = 0;
  for (int i = 0; i < 8; ++i) pg->gmap |= (unsigned)(fin[i] & 15) << (4 * i);
  return true;
}

// Tiles of the persistent patch kernel's 128-wide walk over a stride-1 3x3 conv with an [n, h, w, cout] destination (the tile
// decomposition depends on nothing else): the size of a sign-word buffer (SignRequest), 2048 bytes per tile.
long pp_sign_tiles(int n, int h, int w, int cout) {
  if (cout < 128 || cout % 128 != 0) return 0;
  FastGeom g;
  g.TH = 3; g.TW = 3; g.ss = 1; g.dsd = 1; g.ph = 0; g.pw = 0; g.dhs = 1; g.dws = 1; g.dh0 = -1; g.dw0 = -1;
  g.Hd = g.OH = h; g.Wd = g.OW = w;
  PatchGeom pg;
  if (!plan_patch(g, &pg, 0.0)) return 0;
  return (long)n * pg.tiles_h * pg.tiles_w * (cout / 128);
}

static int run_fast(const float* src, const float* wt, const float* bias, const float* residual,
                    const float* rowscale, const float* chanscale, const float* actmask, float* dst, const FastGeom& g,
                    hipStream_t st) {
  if (g.M <= 0) return SRHIP_OK;
  if (g.src_pp || g.dst_pp) {                             // padded-plane operands: the persistent patch kernel or nothing
    PatchGeom pg;
    const int ef = g.flags & 0x3f;
    if (g_conv_math == 1 && g.K >= 64 && g.K % 8 == 0 && g.C % 32 == 0 && !(ef & (SRHIP_EPI_CHANSCALE | SRHIP_EPI_ROWSCALE)) && !g.accumulate &&
        plan_patch(g, &pg, 0.0)) {                        // (any patch efficiency: the planes have no other kernel)
      const int nbm = g.N * pg.tiles_h * pg.tiles_w;
      const bool wide = g.K >= 128;
      const int nbn = cdiv(g.K, wide ? 128 : 64);
      if (wide && !residual) {                            // >= 256 destination channels: the 8-wave two-group kernel (conv_patch8.hip)
        const int rc8 = launch_patch8(src, wt, bias, actmask, dst, g, pg, nbm, ef, st);
        if (rc8 >= 0) return rc8;
      }
      const int rc = launch_patch_pers(src, wt, bias, residual, actmask, dst, g, pg, nbm, nbn, wide, 0, ef, st);
      if (rc >= 0) return rc;
    }
    set_error("conv2d (padded planes): shape / arithmetic mode not served by the persistent patch kernel");
    return SRHIP_ERR_ARG;
  }
#define SRHIP_LF(BM_, BN_, WM_, WN_, BK_) \
  return launch_fast<BM_, BN_, WM_, WN_, BK_>(src, wt, bias, residual, rowscale, chanscale, actmask, dst, g, st)
  // <= 4 destination channels, stride-1 3x3, big image: exact-fp32 VALU kernel (both arithmetic modes; cfg 22 turns it off)
  if (g.K <= 4 && g_fast_cfg != 20 && g_fast_cfg != 22 && g.TH == 3 && g.TW == 3 && g.ss == 1 && g.dsd == 1 && g.ph == 0 &&
      g.pw == 0 && g.Hd == g.OH && g.Wd == g.OW && (g.dhs == 1 || g.dhs == -1) && (g.dws == 1 || g.dws == -1) &&
      !(g.flags & ~(SRHIP_EPI_BIAS | SRHIP_EPI_LRELU | SRHIP_EPI_GRADDATA)) && !g.accumulate && g.M >= 65536) {
    const int th = cdiv(g.OH, 16), tw = cdiv(g.OW, 16);
    const int lo_h = g.dhs > 0 ? g.dh0 : g.dh0 + 2 * g.dhs, lo_w = g.dws > 0 ? g.dw0 : g.dw0 + 2 * g.dws;
    if (g.K == 3)
      hipLaunchKernelGGL((narrow_conv_kernel<3>), dim3(g.N * th * tw), dim3(256), 0, st, src, wt, bias, dst, g, th, tw, lo_h, lo_w);
    else
      hipLaunchKernelGGL((narrow_conv_kernel<4>), dim3(g.N * th * tw), dim3(256), 0, st, src, wt, bias, dst, g, th, tw, lo_h, lo_w);
    return check_launch("narrow_conv");
  }
  // one destination channel, small grid: a wave per output pixel (cfg 22 turns it off with the other VALU kernel)
  if (g.K == 1 && g.C % 4 == 0 && g.lds % 4 == 0 && (((uintptr_t)src | (uintptr_t)wt) & 15) == 0 && g.M < 65536 && g_fast_cfg != 20 &&
      g_fast_cfg != 22 && !(g.flags & ~(SRHIP_EPI_BIAS | SRHIP_EPI_LRELU | SRHIP_EPI_GRADDATA)) && !g.accumulate) {
    hipLaunchKernelGGL(dot_conv_kernel, dim3(cdiv(g.M, 4)), dim3(256), 0, st, src, wt, bias, dst, g);
    return check_launch("dot_conv");
  }
  // 16-bit product arithmetic (see mma16): 0 split-bf16 (three products), 1 one bf16 product (SRHIP_MATH_HALF on gradient
  // data: every dgrad, and forward calls flagged GRADDATA), 2 one fp16 product (SRHIP_MATH_HALF on activations)
  const int prod = g_conv_math == 2 ? ((g.flags & SRHIP_EPI_GRADDATA) ? 1 : 2) : 0;
  const float* w16 = wt + (size_t)(g.w_bytes >> 2) * (prod == 2 ? 2 : 1);      // packed-weight section (fast_pack_store)
  if (g.K <= 32 && g_conv_math >= 1 && g_fast_cfg != 20 && !(g.flags & (SRHIP_EPI_CHANSCALE | 0x300))) {
    const int nbm = cdiv(g.M, 128), nbn = cdiv(g.K, 32);
    if (prod == 0)
      hipLaunchKernelGGL((fast_conv_kernel<128, 32, 4, 1, 16, 1>), dim3(nbm * nbn), dim3(256), g_fast_dynlds, st, src,
                         w16, bias, residual, rowscale, chanscale, actmask, dst, g, nbm, nbn);
    else if (prod == 1)
      hipLaunchKernelGGL((fast_conv_kernel<128, 32, 4, 1, 16, 2>), dim3(nbm * nbn), dim3(256), g_fast_dynlds, st, src,
                         w16, bias, residual, rowscale, chanscale, actmask, dst, g, nbm, nbn);
    else
      hipLaunchKernelGGL((fast_conv_kernel<128, 32, 4, 1, 16, 3>), dim3(nbm * nbn), dim3(256), g_fast_dynlds, st, src,
                         w16, bias, residual, rowscale, chanscale, actmask, dst, g, nbm, nbn);
    return check_launch("fast_conv");
  }
  if (g.K <= 32) SRHIP_LF(128, 32, 4, 1, 16);
  // LDS-DMA kernels (g_fast_cfg 20 forces them off): no A-operand scaling, ablation flags or accumulate variants needed
  const int eflags = g.flags & 0x3f;      // bit 0x40: GRADDATA; 0x100/0x200: ablations (reg kernel), 0x400: plain instead of non-temporal epilogue stores
  const bool al16 = g.K % 4 == 0 && g.ldd % 4 == 0 && ((uintptr_t)dst & 15) == 0 && (!residual || (g.ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0)) &&
                    (!actmask || ((uintptr_t)actmask & 15) == 0) && (!bias || ((uintptr_t)bias & 15) == 0);
  // stride-1 3x3 in split-bf16: the patch kernel (g_fast_cfg 21 turns it off, -2 forces it at any size)
  bool patch_to_dma = false;             // g_fast_cfg 23: every launch the patch kernel would take goes to the (bit-identical) DMA kernel
  if (g_conv_math >= 1 && g_fast_cfg != 20 && g_fast_cfg != 21 && (g_fast_cfg < 1 || g_fast_cfg == 23 || g_fast_cfg == 24) && !(g.flags & 0x300) && g.K >= 64 && al16 &&
      !(eflags & (SRHIP_EPI_CHANSCALE | SRHIP_EPI_ROWSCALE)) && !g.accumulate) {
    PatchGeom pg;
    if (plan_patch(g, &pg)) {
      const int nbm = g.N * pg.tiles_h * pg.tiles_w;
      const bool wide = g.K >= 128 && g_fast_cfg != 24;   // 24: experiment, 64-wide N tiles for every Cout (twice the blocks)
      const int nbn = cdiv(g.K, wide ? 128 : 64);
      if (g_fast_cfg == 23) {
        patch_to_dma = (long)nbm * nbn >= 256;
      } else if ((long)nbm * nbn >= 256 || g_fast_cfg == -2) {
        const float* wsplit = w16;
        if (!(g.flags & 0x400) && wide && prod == 0 && !residual && g_fast_cfg != 23) {   // >= 256 destination channels: the 8-wave two-group kernel
          const int rc8 = launch_patch8(src, wt, bias, actmask, dst, g, pg, nbm, eflags, st);
          if (rc8 >= 0) return rc8;
        }
        if (!(g.flags & 0x400)) {                         // persistent tile walk (srhip_debug_set(5, -1): never)
          const int rc = launch_patch_pers(src, wt, bias, residual, actmask, dst, g, pg, nbm, nbn, wide, prod, eflags, st);
          if (rc >= 0) return rc;
        }
        if (prod != 0) {                                  // SRHIP_MATH_HALF: run-time epilogue flags keep the variant count down
#define SRHIP_LPH(BN_, PROD_)                                                                                       \
  do {                                                                                                              \
    hipLaunchKernelGGL((conv_patch_kernel<BN_, -1, PROD_>), dim3(nbm * nbn), dim3(256), 0, st, src, wsplit, bias,   \
                       residual, actmask, dst, g, pg, nbm, nbn);                                                   \
    return check_launch("conv_patch");                                                                              \
  } while (0)
          if (wide && prod == 1) SRHIP_LPH(128, 1);
          if (wide) SRHIP_LPH(128, 2);
          if (prod == 1) SRHIP_LPH(64, 1);
          SRHIP_LPH(64, 2);
#undef SRHIP_LPH
        }
#define SRHIP_LP(BN_, EPI_)                                                                                         \
  do {                                                                                                              \
    hipLaunchKernelGGL((conv_patch_kernel<BN_, EPI_>), dim3(nbm * nbn), dim3(256), 0, st, src, wsplit, bias,       \
                       residual, actmask, dst, g, pg, nbm, nbn);                                                   \
    return check_launch("conv_patch");                                                                              \
  } while (0)
#define SRHIP_LPE(BN_)                                                  \
  do {                                                                  \
    if (eflags == 0) SRHIP_LP(BN_, 0);                                  \
    if (eflags == SRHIP_EPI_BIAS) SRHIP_LP(BN_, 1);                     \
    if (eflags == (SRHIP_EPI_BIAS | SRHIP_EPI_LRELU)) SRHIP_LP(BN_, 3); \
    if (eflags == SRHIP_EPI_ACTMASK) SRHIP_LP(BN_, 32);                 \
    if (eflags == SRHIP_EPI_RESIDUAL) SRHIP_LP(BN_, 4);                 \
    SRHIP_LP(BN_, -1);                                                  \
  } while (0)
        if (wide) SRHIP_LPE(128);
        if (g_patch_ks && g.C % 32 == 0 && g.C >= 32) {    // K-split form of the 64-wide tile (srhip_debug_set(10, 0): the 2 x 2 wave grid of rounds 1-3)
#define SRHIP_LK(EPI_)                                                                                              \
  do {                                                                                                              \
    hipLaunchKernelGGL((conv_patch_ks_kernel<EPI_>), dim3(nbm * nbn), dim3(256), 0, st, src, wsplit, bias, residual, \
                       actmask, dst, g, pg, nbm, nbn);                                                             \
    return check_launch("conv_patch_ks");                                                                           \
  } while (0)
          if (eflags == 0) SRHIP_LK(0);
          if (eflags == SRHIP_EPI_BIAS) SRHIP_LK(1);
          if (eflags == SRHIP_EPI_RESIDUAL) SRHIP_LK(4);
          SRHIP_LK(-1);
#undef SRHIP_LK
        }
        SRHIP_LPE(64);
#undef SRHIP_LPE
#undef SRHIP_LP
      }
    }
  }
  if (g_fast_cfg != 20 && (g_fast_cfg < 1 || g_fast_cfg == 23 || g_fast_cfg == 24) && !(g.flags & 0x300) && g.K >= 64 && al16 &&
      (!(eflags & SRHIP_EPI_CHANSCALE) || ((uintptr_t)chanscale & 15) == 0)) {
    const int nbm = cdiv(g.M, 128);
    const bool force = g_fast_cfg == -1 || patch_to_dma;   // tests: take the DMA kernels at any problem size
    // fp32: the register-staged kernel wins below ~2 tiles per CU; split-bf16: its fp32 MFMAs cost 5x more than
    // the DMA kernel's, so the DMA kernel is taken from half a wave of tiles on
    const long min_tiles = g_conv_math >= 1 ? 128 : 512;
    // (round 5, measured and dropped: 64-wide tiles for launches with fewer than 384 wide tiles -- D's 512 -> 512 stride-2 conv at 14 x 14,
    // 196 -> 392 blocks -- ran 154.5 us against 143.8)
    const bool wide = g.K >= 128 && ((long)nbm * cdiv(g.K, 128) >= min_tiles || force);
    const long b64 = (long)nbm * cdiv(g.K, 64);
    if (wide || b64 >= min_tiles || force) {
#define SRHIP_LD(BN_, EPI_)                                                                                        \
  do {                                                                                                             \
    const int nbn = cdiv(g.K, BN_);                                                                                \
    if (g.dst2_pp) g_dst2_req.served = 1;                                                                          \
    PhaseSet ps_;                                                                                                  \
    int grid_ = nbm * nbn;                                                                                         \
    if (g_phase_req.active) {                          /* all phases of a strided data gradient in this launch */  \
      ps_ = g_phase_req.ps;                                                                                        \
      grid_ = 0;                                                                                                   \
      for (int k_ = 0; k_ < ps_.n; ++k_) {                                                                         \
        ps_.first[k_] = grid_;                                                                                     \
        grid_ += ps_.p[k_].nblk_m * nbn;                                                                           \
      }                                                                                                            \
      ps_.first[ps_.n] = grid_;                                                                                    \
      g_phase_req.launched = 1;                                                                                    \
    }                                                                                                              \
    if (g_conv_math == 1) {                                                                                        \
      hipLaunchKernelGGL((fast_conv_dma_kernel<128, BN_, EPI_, 1>), dim3(grid_), dim3(256), g_fast_dynlds, st,     \
                         src, w16, bias, residual, rowscale, chanscale, actmask, dst, g, nbm,                      \
                         nbn, ps_);                                                                                \
    } else                                                                                                         \
      hipLaunchKernelGGL((fast_conv_dma_kernel<128, BN_, EPI_, 0>), dim3(grid_), dim3(256), g_fast_dynlds, st,     \
                         src, wt, bias, residual, rowscale, chanscale, actmask, dst, g, nbm, nbn, ps_);            \
    return check_launch("fast_conv_dma");                                                                          \
  } while (0)
#define SRHIP_LDE(BN_)                                                  \
  do {                                                                  \
    if (g.accumulate) SRHIP_LD(BN_, -1);                                \
    if (eflags == 0) SRHIP_LD(BN_, 0);                                  \
    if (eflags == SRHIP_EPI_BIAS) SRHIP_LD(BN_, 1);                     \
    if (eflags == (SRHIP_EPI_BIAS | SRHIP_EPI_LRELU)) SRHIP_LD(BN_, 3); \
    if (eflags == SRHIP_EPI_ACTMASK) SRHIP_LD(BN_, 32);                 \
    if (eflags == SRHIP_EPI_RESIDUAL) SRHIP_LD(BN_, 4);                 \
    if (eflags == 29) SRHIP_LD(BN_, 29); /* bias|residual|rowscale|chanscale: the attention tail */ \
    SRHIP_LD(BN_, -1);                                                  \
  } while (0)
      if (prod != 0) {                                    // SRHIP_MATH_HALF: run-time epilogue flags
#define SRHIP_LDH(BN_, MATH_)                                                                                       \
  do {                                                                                                              \
    const int nbn = cdiv(g.K, BN_);                                                                                 \
    hipLaunchKernelGGL((fast_conv_dma_kernel<128, BN_, -1, MATH_>), dim3(nbm * nbn), dim3(256), g_fast_dynlds, st,  \
                       src, w16, bias, residual, rowscale, chanscale, actmask, dst, g, nbm, nbn, PhaseSet());       \
    return check_launch("fast_conv_dma");                                                                           \
  } while (0)
        if (wide && prod == 1) SRHIP_LDH(128, 2);
        if (wide) SRHIP_LDH(128, 3);
        if (prod == 1) SRHIP_LDH(64, 2);
        SRHIP_LDH(64, 3);
#undef SRHIP_LDH
      }
      if (wide) SRHIP_LDE(128);
      SRHIP_LDE(64);
#undef SRHIP_LDE
#undef SRHIP_LD
    }
  }
  const bool k32 = g.C % 32 == 0;
  if (g_fast_cfg == 1 && g.K >= 128 && k32) SRHIP_LF(128, 128, 2, 2, 32);
  if (g_fast_cfg == 2 && g.K >= 128) SRHIP_LF(64, 128, 1, 4, 16);
  if (g_fast_cfg == 3 && g.K >= 128 && k32) SRHIP_LF(64, 128, 1, 4, 32);
  if (g_fast_cfg == 4 && g.K >= 128) SRHIP_LF(256, 128, 4, 2, 16);
  if (g_fast_cfg == 5 && g.K >= 128) SRHIP_LF(128, 128, 2, 2, 16);
  if (g_fast_cfg == 6 && k32) SRHIP_LF(128, 64, 2, 2, 32);
  if (g_fast_cfg == 7) SRHIP_LF(64, 64, 2, 2, 16);
  if (g_fast_cfg == 8 && k32) SRHIP_LF(128, 64, 4, 1, 32);
  const long b128 = (long)cdiv(g.M, 128) * cdiv(g.K, 128);
  if (g.K >= 128 && b128 >= 512) SRHIP_LF(128, 128, 2, 2, 16);
  const long b64 = (long)cdiv(g.M, 128) * cdiv(g.K, 64);
  if (b64 >= 512) SRHIP_LF(128, 64, 2, 2, 16);
  SRHIP_LF(64, 64, 2, 2, 16);
#undef SRHIP_LF
}

static bool bytes_ok(long pixels, int ld, int c, unsigned* out) {
  const long b = pixels > 0 ? ((pixels - 1) * (long)ld + c) * 4L : 0;
  if (b >= (1L << 31)) return false;
  *out = (unsigned)b;
  return true;
}

int fast_conv2d_fwd(const float* x, const float* packed, const float* bias, const float* residual,
                    const float* rowscale, const float* chanscale, float* y, int n, int h, int w, int cin, int cout, int kh, int kw,
                    int stride, int pad, int ldx, int ldy, int ldr, float slope, int flags, hipStream_t st) {
  SRHIP_REQUIRE(ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)packed) & 15) == 0, "conv2d_fwd: x/packed must be 16-byte aligned with ldx % 4 == 0");
  FastGeom g;
  g.N = n; g.Hs = h; g.Ws = w; g.C = cin; g.lds = ldx;
  g.OH = (h + 2 * pad - kh) / stride + 1;
  g.OW = (w + 2 * pad - kw) / stride + 1;
  SRHIP_REQUIRE(g.OH > 0 && g.OW > 0, "conv2d_fwd: empty output");
  const long M = (long)n * g.OH * g.OW;
  SRHIP_REQUIRE(M < (1L << 31), "conv2d_fwd: pixel count overflows int32");
  g.M = (int)M; g.ss = stride;
  g.TH = kh; g.TW = kw; g.dh0 = -pad; g.dhs = 1; g.dw0 = -pad; g.dws = 1;
  g.kh0 = 0; g.khs = 1; g.kw0 = 0; g.kws = 1; g.KW = kw;
  g.Hd = g.OH; g.Wd = g.OW; g.dsd = 1; g.ph = 0; g.pw = 0; g.ldd = ldy; g.K = cout;
  g.ldw = kh * kw * cin; g.ldr = ldr; g.slope = slope; g.flags = flags | g_fast_ablate; g.accumulate = 0; g.dst_identity = 1;
  SRHIP_REQUIRE(bytes_ok((long)n * h * w, ldx, cin, &g.src_bytes), "conv2d_fwd: source tensor >= 2 GiB");
  g.w_bytes = (unsigned)((long)cout * g.ldw * 4);
  if (g_dst2_req.pp != nullptr && stride == 1 && cout % 8 == 0 && g.OH == g.Hd && g.OW == g.Wd) {   // srhip_conv2d_fwd_dual
    g.dst2_pp = g_dst2_req.pp;
    g.dst2_guard = pp_guard(g.Wd);
  }
  return run_fast(x, packed, bias, residual, rowscale, chanscale, nullptr, y, g, st);
}

// 3x3 stride-1 pad-1 forward / data gradient with padded-plane operands (src_pp / dst_pp; with dst_pp an activation mask is pp too)
int fast_conv2d_fwd_pp(const void* x, int x_pp, const float* packed, const float* bias, void* y, int y_pp, int n, int h, int w, int cin,
                       int cout, int ldx, int ldy, float slope, int flags, hipStream_t st) {
  FastGeom g;
  g.N = n; g.Hs = h; g.Ws = w; g.C = cin; g.lds = x_pp ? cin : ldx;
  g.OH = h; g.OW = w;
  const long M = (long)n * h * w;
  SRHIP_REQUIRE(M < (1L << 31), "conv2d_fwd_pp: pixel count overflows int32");
  g.M = (int)M; g.ss = 1;
  g.TH = 3; g.TW = 3; g.dh0 = -1; g.dhs = 1; g.dw0 = -1; g.dws = 1;
  g.kh0 = 0; g.khs = 1; g.kw0 = 0; g.kws = 1; g.KW = 3;
  g.Hd = h; g.Wd = w; g.dsd = 1; g.ph = 0; g.pw = 0; g.ldd = y_pp ? cout : ldy; g.K = cout;
  g.ldw = 9 * cin; g.ldr = 0; g.slope = slope; g.flags = flags; g.accumulate = 0; g.dst_identity = 1;
  const long ppx = pp_plane_pixels(n, h, w);
  g.src_pp = x_pp; g.dst_pp = y_pp;
  g.src_guard = g.dst_guard = pp_guard(w);
  if (x_pp) {
    SRHIP_REQUIRE(ppx * cin * 4L < (1L << 31), "conv2d_fwd_pp: source planes >= 2 GiB");
    g.src_plane_bytes = (unsigned)(ppx * cin * 2L);
    g.src_bytes = 2u * g.src_plane_bytes;
  } else {
    SRHIP_REQUIRE(bytes_ok(M, ldx, cin, &g.src_bytes), "conv2d_fwd_pp: source tensor >= 2 GiB");
  }
  if (y_pp) {
    SRHIP_REQUIRE(ppx * cout * 4L < (1L << 31), "conv2d_fwd_pp: destination planes >= 2 GiB");
    g.dst_plane_bytes = (unsigned)(ppx * cout * 2L);
  }
  g.w_bytes = (unsigned)((long)cout * g.ldw * 4);
  return run_fast(static_cast<const float*>(x), packed, bias, nullptr, nullptr, nullptr, nullptr, static_cast<float*>(y), g, st);
}

int fast_conv2d_dgrad_pp(const void* dy, int dy_pp, const float* packed, void* dx, int dx_pp, const float* residual, const void* actmask,
                         float slope, int n, int h, int w, int cin, int cout, int ldy, int ldx, int ldr, hipStream_t st) {
  FastGeom g;
  g.N = n; g.Hs = h; g.Ws = w; g.C = cout; g.lds = dy_pp ? cout : ldy;
  g.KW = 3; g.Hd = h; g.Wd = w; g.dsd = 1; g.ldd = dx_pp ? cin : ldx; g.K = cin;
  g.ldw = 9 * cout; g.ldr = ldr; g.slope = slope; g.accumulate = 0;
  g.flags = (residual ? SRHIP_EPI_RESIDUAL : 0) | (actmask ? SRHIP_EPI_ACTMASK : 0) | SRHIP_EPI_GRADDATA;
  g.ss = 1; g.dhs = -1; g.dws = -1; g.khs = 1; g.kws = 1;
  g.dst_identity = 1;
  // stride 1, pad 1, 3 x 3: ONE phase (fast_conv2d_dgrad's general code with stride = 1, pad = 1): tap th reads dy row hh + 1 - th
  g.ph = 0; g.pw = 0; g.OH = h; g.OW = w; g.kh0 = 0; g.kw0 = 0;
  g.TH = 3; g.TW = 3;
  g.dh0 = 1; g.dw0 = 1;
  const long M = (long)n * h * w;
  SRHIP_REQUIRE(M < (1L << 31), "conv2d_dgrad_pp: pixel count overflows int32");
  g.M = (int)M;
  const long ppx = pp_plane_pixels(n, h, w);
  g.src_pp = dy_pp; g.dst_pp = dx_pp;
  g.src_guard = g.dst_guard = pp_guard(w);
  if (dy_pp) {
    SRHIP_REQUIRE(ppx * cout * 4L < (1L << 31), "conv2d_dgrad_pp: dy planes >= 2 GiB");
    g.src_plane_bytes = (unsigned)(ppx * cout * 2L);
    g.src_bytes = 2u * g.src_plane_bytes;
  } else {
    SRHIP_REQUIRE(bytes_ok(M, ldy, cout, &g.src_bytes), "conv2d_dgrad_pp: dy tensor >= 2 GiB");
  }
  if (dx_pp) {
    SRHIP_REQUIRE(ppx * cin * 4L < (1L << 31), "conv2d_dgrad_pp: dx planes >= 2 GiB");
    g.dst_plane_bytes = (unsigned)(ppx * cin * 2L);
  }
  g.w_bytes = (unsigned)((long)cin * g.ldw * 4);
  if (residual && !dx_pp) {                           // srhip_conv2d_dgrad_pp_res3: the kernel that takes them marks the request served
    g.res2 = g_res_req.r2;
    g.res3 = g_res_req.r3;
  }
  return run_fast(static_cast<const float*>(dy), packed, nullptr, residual, nullptr, nullptr, static_cast<const float*>(actmask),
                  static_cast<float*>(dx), g, st);
}

int fast_conv2d_dgrad(const float* dy, const float* packed, float* dx, const float* residual, const float* actmask,
                      float slope, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int ldy,
                      int ldx, int ldr, int accumulate, hipStream_t st) {
  SRHIP_REQUIRE(ldy % 4 == 0 && (((uintptr_t)dy | (uintptr_t)packed) & 15) == 0, "conv2d_dgrad: dy/packed must be 16-byte aligned with ldy % 4 == 0");
  const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
  SRHIP_REQUIRE(ho > 0 && wo > 0, "conv2d_dgrad: empty dy");
  FastGeom g;
  g.N = n; g.Hs = ho; g.Ws = wo; g.C = cout; g.lds = ldy;
  g.KW = kw; g.Hd = h; g.Wd = w; g.dsd = stride; g.ldd = ldx; g.K = cin;
  g.ldw = kh * kw * cout; g.ldr = ldr; g.slope = slope; g.accumulate = accumulate ? 1 : 0;
  g.flags = (residual ? SRHIP_EPI_RESIDUAL : 0) | (actmask ? SRHIP_EPI_ACTMASK : 0) | SRHIP_EPI_GRADDATA;
  g.ss = 1; g.dhs = -1; g.dws = -1; g.khs = stride; g.kws = stride;
  g.dst_identity = stride == 1 ? 1 : 0;
  SRHIP_REQUIRE(bytes_ok((long)n * ho * wo, ldy, cout, &g.src_bytes), "conv2d_dgrad: dy tensor >= 2 GiB");
  g.w_bytes = (unsigned)((long)cin * g.ldw * 4);
  if (residual && stride == 1 && !accumulate) {       // srhip_conv2d_dgrad_res3: the kernel that takes them marks the request served
    g.res2 = g_res_req.r2;
    g.res3 = g_res_req.r3;
  }
  // dx[hh] gathers dy[(hh + pad - kh)/stride] for kh == (hh + pad) mod stride: one dense GEMM per phase
  // Round 5: where the LDS-DMA kernel takes the phases (split-bf16 / fp32 arithmetic), ALL of them go out as ONE launch (PhaseSet):
  // the request rides beside the first phase's run_fast call, which launches every phase when it reaches that kernel.
  PhaseRequest& pq = g_phase_req;
  pq = PhaseRequest();
  if (stride == 2 && g_conv_math != 2 && g_phase_batch) {          // (srhip_debug_set(17, 0): one launch per phase, rounds 1-4)
    bool all = true;
    int k = 0;
    for (int ph = 0; ph < stride && all; ++ph)
      for (int pw = 0; pw < stride && all; ++pw) {
        PhaseSet::P& q = pq.ps.p[k++];
        q.ph = ph; q.pw = pw;
        q.OH = (h - ph + stride - 1) / stride;
        q.OW = (w - pw + stride - 1) / stride;
        q.kh0 = (ph + pad) % stride; q.kw0 = (pw + pad) % stride;
        q.TH = q.kh0 < kh ? (kh - q.kh0 + stride - 1) / stride : 0;
        q.TW = q.kw0 < kw ? (kw - q.kw0 + stride - 1) / stride : 0;
        q.dh0 = (ph + pad - q.kh0) / stride; q.dw0 = (pw + pad - q.kw0) / stride;
        const long M = (long)n * q.OH * q.OW;
        all = q.OH > 0 && q.OW > 0 && q.TH > 0 && q.TW > 0 && M < (1L << 31);
        q.M = (int)M;
        q.nblk_m = cdiv(M, 128);
      }
    if (all) {
      pq.ps.n = stride * stride;
      pq.active = 1;
    }
  }
  for (int ph = 0; ph < stride; ++ph) {
    for (int pw = 0; pw < stride; ++pw) {
      g.ph = ph; g.pw = pw;
      g.OH = (h - ph + stride - 1) / stride;
      g.OW = (w - pw + stride - 1) / stride;
      if (g.OH <= 0 || g.OW <= 0) continue;
      g.kh0 = (ph + pad) % stride; g.kw0 = (pw + pad) % stride;
      g.TH = g.kh0 < kh ? (kh - g.kh0 + stride - 1) / stride : 0;
      g.TW = g.kw0 < kw ? (kw - g.kw0 + stride - 1) / stride : 0;
      if (g.TH == 0 || g.TW == 0) { g.TH = 0; g.TW = 0; }
      g.dh0 = (ph + pad - g.kh0) / stride; g.dw0 = (pw + pad - g.kw0) / stride;
      const long M = (long)n * g.OH * g.OW;
      SRHIP_REQUIRE(M < (1L << 31), "conv2d_dgrad: pixel count overflows int32");
      g.M = (int)M;
      int rc = run_fast(dy, packed, nullptr, residual, nullptr, nullptr, actmask, dx, g, st);
      const bool all_out = pq.active && pq.launched;
      pq = PhaseRequest();                              // (a request only ever rides beside the first phase)
      if (rc) return rc;
      if (all_out) return SRHIP_OK;
    }
  }
  return SRHIP_OK;
}

// ---- wgrad ------------------------------------------------------------------------------------- //
struct FastWgradPlan {
  int bm, bn, bk, nsplit, chunks_per_split;
};
int g_rowtap_pipe = 0; // srhip_debug_set(9, v): 1 = software-pipelined row-tap kernel (conversion of chunk k+1 beside the MFMAs of chunk k, 2 blocks / CU)
int g_rowtap_addr = 1; // srhip_debug_set(8, v): 0 = per-lane DMA offsets of rounds 1-3 in wgrad_rowtap_kernel (A/B), 1 = scalar chunk offsets
int g_wgrad_cfg = 0;   // experiment knob (srhip_debug_set(1, cfg)): 0 heuristic, 1: bn=64, 2: bn=128, 8: scalar split-K reduce kernels (rounds 1-3), +10: register-staged kernel
// row-tap kernel (wgrad_rowtap_kernel): split-bf16, 3x3 stride 1 pad 1, Cin % 64 == 0, Cout % 4 == 0 (g_wgrad_cfg 7 turns it off)
static int rowtap_ok(int cin, int cout, int kh, int kw, int stride, int pad) {   // 0: no, 1: 128 x (kh, 64 ci), 2: 64 x (kh, 128 ci)
  if (!(g_conv_math >= 1 && g_wgrad_cfg != 7 && (g_wgrad_cfg < 10 || g_wgrad_cfg >= 100) && kh == 3 && kw == 3 && stride == 1 && pad == 1 && cout % 4 == 0))
    return 0;
  if (cout >= 128 && cin % 64 == 0) return 1;
  if (cout == 64 && cin % 128 == 0) return 2;
  return 0;
}

static FastWgradPlan plan_fast_wgrad(long P, int cout, int ktot, int rowtap = 0) {
  FastWgradPlan p;
  p.bm = cout > 64 ? 128 : 64;
  p.bn = (ktot % 128 == 0) ? 128 : 64;            // Ktot = 9*64 tiles exactly by 64, not by 128
  if (g_wgrad_cfg % 10 == 1) p.bn = 64;
  if (g_wgrad_cfg % 10 == 2) p.bn = 128;
  p.bk = (g_wgrad_cfg == 3 || g_wgrad_cfg == 4) ? 32 : FBK;
  if (g_wgrad_cfg == 4) p.bn = 64;
  if (g_wgrad_cfg == 5) {                              // 256-wide tiles: 32 MFMAs per wave per chunk for Cout=256 / Cout=64
    if (cout % 256 == 0 && ktot % 64 == 0) { p.bm = 256; p.bn = 64; }
    else if (cout == 64 && ktot % 256 == 0) { p.bm = 64; p.bn = 256; }
  }
  // split-bf16 wgrad is VALU-issue bound (every wave splits the fragments it reads): where Ktot only tiles by 64
  // (Cin = 64), a 256 x 64 tile doubles the MFMAs per split fragment (measured -10 % on 64->256 convs)
  if (g_conv_math >= 1 && g_wgrad_cfg == 0 && p.bn == 64 && cout % 256 == 0) p.bm = 256;
  if (rowtap == 1) { p.bm = 128; p.bn = 192; }
  if (rowtap == 2) { p.bm = 64; p.bn = 384; }
  const long tiles = (long)cdiv(cout, p.bm) * cdiv(ktot, p.bn);
  const int nchunks = cdiv(P, p.bk);
  // blocks aimed at: ~2.5 per CU; the row-tap kernel runs 3 per CU and is fastest with exactly one full wave of
  // blocks (768: measured 0.161 -> 0.128 ms on RAB conv1 against 640; 512 and 1024 are both slower)
  const long target = rowtap ? (g_wgrad_cfg >= 100 ? g_wgrad_cfg : 768) : 640;
  long ns = (target + tiles - 1) / tiles;
  const long maxsplit = (nchunks + 15) / 16;          // at least 16 chunks (256 pixels) per split
  if (ns > maxsplit) ns = maxsplit;
  if (ns > (tiles <= 2 ? 768 : 256)) ns = tiles <= 2 ? 768 : 256;   // one- and two-tile GEMMs (1x1, 64 -> 64) still want a full wave of blocks
  if (ns < 1) ns = 1;
  if (ns >= 8) ns = (ns + 7) / 8 * 8;                  // multiples of 8: one split per XCD lane (see kernels)
  p.chunks_per_split = (int)((nchunks + ns - 1) / ns);
  p.nsplit = cdiv(nchunks, p.chunks_per_split);
  if (p.nsplit >= 8 && p.nsplit % 8 != 0) p.nsplit = (p.nsplit + 7) / 8 * 8;   // empty tail splits write zeros
  return p;
}

// splits per image of wgrad_1x1_scaled_kernel (the attention tail's 64 -> 64 1x1 conv): ~one block per CU, >= 64 pixels per block
static int tail1x1_splits_per_image(int n, int hw) {
  int sp = cdiv(256, n);                               // one block per CU: the split-K reduce of a 64 x 64 tile is latency-bound on the split count
  if (sp > cdiv(hw, 64)) sp = cdiv(hw, 64);
  return sp < 1 ? 1 : sp;
}
size_t fast_conv2d_wgrad_workspace(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  const int ho = (h + 2 * pad - kh) / stride + 1, wo = (w + 2 * pad - kw) / stride + 1;
  const long P = (long)n * ho * wo;
  if (P <= 0) return 0;
  FastWgradPlan p = plan_fast_wgrad(P, cout, kh * kw * cin, rowtap_ok(cin, cout, kh, kw, stride, pad));
  long ns = p.nsplit;
  if (kh == 1 && kw == 1 && cin == 64 && cout == 64 && stride == 1 && pad == 0) {
    const long t = (long)n * tail1x1_splits_per_image(n, h * w);
    if (t > ns) ns = t;
  }
  return (size_t)ns * ((size_t)cout * kh * kw * cin + cout) * sizeof(float);
}

int fast_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, const float* xrow, const float* xchan,
                      int accumulate, void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int ldx, int ldy,
                      hipStream_t st) {
  SRHIP_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0,
                "conv2d_wgrad: x/dy must be 16-byte aligned with row strides % 4 == 0");
  WgradGeom g;
  g.N = n; g.H = h; g.W = w; g.C = cin; g.ldx = ldx;
  g.Ho = (h + 2 * pad - kh) / stride + 1;
  g.Wo = (w + 2 * pad - kw) / stride + 1;
  SRHIP_REQUIRE(g.Ho > 0 && g.Wo > 0, "conv2d_wgrad: empty output");
  g.K = cout; g.ldy = ldy; g.KH = kh; g.KW = kw; g.stride = stride; g.pad = pad;
  const long P = (long)n * g.Ho * g.Wo;
  SRHIP_REQUIRE(P < (1L << 31) - 64, "conv2d_wgrad: pixel count overflows int32");
  g.P = (int)P; g.Ktot = kh * kw * cin;
  SRHIP_REQUIRE(bytes_ok((long)n * h * w, ldx, cin, &g.x_bytes) && bytes_ok(P, ldy, cout, &g.dy_bytes),
                "conv2d_wgrad: tensor >= 2 GiB");
  // (an operand-scaled 3x3 would plan without row-tap and may then need more workspace than the query reported)
  const int rowtap = (!xrow && !xchan) ? rowtap_ok(cin, cout, kh, kw, stride, pad) : 0;
  FastWgradPlan p = plan_fast_wgrad(P, cout, g.Ktot, rowtap);
  g.nsplit = p.nsplit; g.chunks_per_split = p.chunks_per_split;
  const size_t need = (size_t)p.nsplit * ((size_t)cout * g.Ktot + cout) * sizeof(float);
  if (!workspace || workspace_bytes < need) {
    set_error("conv2d_wgrad: workspace %zu bytes < required %zu", workspace_bytes, need);
    return SRHIP_ERR_WORKSPACE;
  }
  float* partial = static_cast<float*>(workspace);
  float* bias_partial = partial + (size_t)p.nsplit * cout * g.Ktot;
  // the attention tail's 1x1 conv with both operand scales: pixel-contraction kernel (wgrad_1x1_scaled_kernel); g_wgrad_cfg 8 / 6
  // keep the generic kernel for A/B runs
  if (xrow && xchan && kh == 1 && kw == 1 && cin == 64 && cout == 64 && stride == 1 && pad == 0 && ldx == 64 && ldy == 64 &&
      g_wgrad_cfg != 8 && g_wgrad_cfg != 6 && n >= 1 &&
      workspace_bytes >= (size_t)n * tail1x1_splits_per_image(n, h * w) * ((size_t)cout * g.Ktot + cout) * sizeof(float)) {
    const int hw = h * w;
    const int sp = tail1x1_splits_per_image(n, hw);
    const int per = (cdiv(hw, sp) + 1) & ~1;           // even: a pixel pair never straddles two blocks
    const int ns = n * sp;
    float* bp = db ? partial + (size_t)ns * cout * g.Ktot : nullptr;
    hipLaunchKernelGGL(wgrad_1x1_scaled_kernel, dim3(ns), dim3(256), 0, st, x, dy, xrow, xchan, partial, bp, hw, per, sp);
    int rc1 = check_launch("wgrad_1x1_scaled");
    if (rc1) return rc1;
    const float* pp[1] = {partial};
    const float* bpp[1] = {bp};
    float* dwp[1] = {dw};
    float* dbp[1] = {db};
    if (launch_reduce4(1, pp, bpp, dwp, dbp, ns, cout, cin, 1, g.Ktot, accumulate, st)) return check_launch("fast_wgrad_reduce4");
    hipLaunchKernelGGL(fast_wgrad_reduce_kernel<4>, dim3(cdiv((long)cout * g.Ktot + (db ? cout : 0), 64)), dim3(256), 0, st, partial, bp, dw, db, ns,
                       cout, cin, 1, g.Ktot, accumulate);
    return check_launch("fast_wgrad_reduce");
  }
  const int blocks = cdiv(cout, p.bm) * cdiv(g.Ktot, p.bn) * p.nsplit;
#define SRHIP_LW(BM_, BN_, WM_, WN_)                                                                              \
  do {                                                                                                            \
    if (!xrow && !xchan && g_wgrad_cfg < 10 && g_conv_math == 2)                                           \
      hipLaunchKernelGGL((fast_wgrad_dma_kernel<BM_, BN_, WM_, WN_, 16, 2>), dim3(blocks), dim3(256), 0, st, x, dy, \
                         partial, db ? bias_partial : nullptr, g);                                               \
    else if (!xrow && !xchan && g_wgrad_cfg < 10 && g_conv_math == 1)                                           \
      hipLaunchKernelGGL((fast_wgrad_dma_kernel<BM_, BN_, WM_, WN_, 16, 1>), dim3(blocks), dim3(256), 0, st, x, dy, \
                         partial, db ? bias_partial : nullptr, g);                                               \
    else if (!xrow && !xchan && g_wgrad_cfg < 10 && p.bk == 32)                                                        \
      hipLaunchKernelGGL((fast_wgrad_dma_kernel<BM_, BN_, WM_, WN_, 32, 0>), dim3(blocks), dim3(256), 0, st, x, dy, \
                         partial, db ? bias_partial : nullptr, g);                                               \
    else if (!xrow && !xchan && g_wgrad_cfg < 10)                                                                 \
      hipLaunchKernelGGL((fast_wgrad_dma_kernel<BM_, BN_, WM_, WN_, 16, 0>), dim3(blocks), dim3(256), 0, st, x, dy, \
                         partial, db ? bias_partial : nullptr, g);                                               \
    else                                                                                                          \
      hipLaunchKernelGGL((fast_wgrad_kernel<BM_, BN_, WM_, WN_>), dim3(blocks), dim3(256), 0, st, x, dy, partial, \
                         db ? bias_partial : nullptr, xrow, xchan, g);                                           \
  } while (0)
  if (rowtap) {
    const int nseg = cdiv(g.Wo, 16);
    // paired tails (see the kernel): rows of 16 q + rem pixels, 0 < rem <= 8, at least two rows per image
    const int rem = g.Wo & 15;
    const int tail_rem = (rem > 0 && rem <= 8 && g.Wo >= 16 && g.Ho >= 2 && g_wgrad_cfg != 9) ? rem : 0;
    const int nchunks_rt = tail_rem ? ((g.N * g.Ho + 1) / 2) * (2 * (g.Wo / 16) + 1) : g.N * g.Ho * nseg;
    const int cps = cdiv(nchunks_rt, p.nsplit);
#define SRHIP_RT(BM_, CIS_, SP_)                                                                                         \
  do {                                                                                                                 \
    if (g_rowtap_pipe)                                                                                                 \
      hipLaunchKernelGGL((wgrad_rowtap_kernel<BM_, CIS_, SP_, 1, true>), dim3(blocks), dim3(256), 0, st, x, dy, partial, \
                         db ? bias_partial : nullptr, g, nseg, cps, tail_rem, WgradBatch{});                           \
    else if (g_rowtap_addr)                                                                                            \
      hipLaunchKernelGGL((wgrad_rowtap_kernel<BM_, CIS_, SP_, 1>), dim3(blocks), dim3(256), 0, st, x, dy, partial,     \
                         db ? bias_partial : nullptr, g, nseg, cps, tail_rem, WgradBatch{});                           \
    else                                                                                                               \
      hipLaunchKernelGGL((wgrad_rowtap_kernel<BM_, CIS_, SP_, 0>), dim3(blocks), dim3(256), 0, st, x, dy, partial,     \
                         db ? bias_partial : nullptr, g, nseg, cps, tail_rem, WgradBatch{});                           \
  } while (0)
    if (rowtap == 1 && g_conv_math == 2) SRHIP_RT(128, 64, false);
    else if (rowtap == 1) SRHIP_RT(128, 64, true);
    else if (g_conv_math == 2) SRHIP_RT(64, 128, false);
    else SRHIP_RT(64, 128, true);
#undef SRHIP_RT
  } else if (p.bm == 256 && p.bn == 64)
    SRHIP_LW(256, 64, 4, 1);
  else if (p.bm == 64 && p.bn == 256)
    SRHIP_LW(64, 256, 1, 4);
  else if (p.bm == 128 && p.bn == 128)
    SRHIP_LW(128, 128, 2, 2);
  else if (p.bm == 128)
    SRHIP_LW(128, 64, 2, 2);
  else if (p.bn == 128)
    SRHIP_LW(64, 128, 1, 4);
  else
    SRHIP_LW(64, 64, 2, 2);
#undef SRHIP_LW
  int rc = check_launch("fast_wgrad");
  if (rc) return rc;
  const long total = (long)cout * g.Ktot + (db ? cout : 0);
  {
    const float* pp[1] = {partial};
    const float* bp[1] = {bias_partial};
    float* dwp[1] = {dw};
    float* dbp[1] = {db};
    if (g_wgrad_cfg != 6 && launch_reduce4(1, pp, bp, dwp, dbp, p.nsplit, cout, cin, kh * kw, g.Ktot, accumulate, st))
      return check_launch("fast_wgrad_reduce4");
  }
  if (p.nsplit >= (g_wgrad_cfg == 6 ? 256 : 64))
    hipLaunchKernelGGL(fast_wgrad_reduce_kernel<16>, dim3(cdiv(total, 64)), dim3(1024), 0, st, partial, bias_partial, dw,
                       db, p.nsplit, cout, cin, kh * kw, g.Ktot, accumulate);
  else
    hipLaunchKernelGGL(fast_wgrad_reduce_kernel<4>, dim3(cdiv(total, 64)), dim3(256), 0, st, partial, bias_partial, dw,
                       db, p.nsplit, cout, cin, kh * kw, g.Ktot, accumulate);
  return check_launch("fast_wgrad_reduce");
}

// ---- grouped row-tap weight gradient: nprob (2..4) convolutions of one shape, one main launch + nprob reduces ----
int fast_wgrad_multi_ok(int cin, int cout, int kh, int kw, int stride, int pad) {
  return g_conv_math >= 1 ? rowtap_ok(cin, cout, kh, kw, stride, pad) : 0;
}
static int multi_nsplit(const FastWgradPlan& p, int nprob) {
  int ns = p.nsplit / nprob;
  ns = ns / 8 * 8;
  return ns < 8 ? 0 : ns;
}
// largest group size (2..4) a problem of this geometry can share a launch with, 0: none (shape, arithmetic mode or too few chunks)
int fast_wgrad_multi_max(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  const int rowtap = fast_wgrad_multi_ok(cin, cout, kh, kw, stride, pad);
  if (!rowtap) return 0;
  FastWgradPlan p = plan_fast_wgrad((long)n * h * w, cout, kh * kw * cin, rowtap);
  for (int k = 4; k >= 2; --k)
    if (multi_nsplit(p, k) > 0) return k;
  return 0;
}
size_t fast_conv2d_wgrad_multi_workspace(int nprob, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad) {
  return fast_conv2d_wgrad_workspace(n, h, w, cin, cout, kh, kw, stride, pad);     // nprob * (nsplit / nprob) partial sets
}
int fast_conv2d_wgrad_multi(int nprob, const float* const* x, const float* const* dy, float* const* dw, float* const* db,
                            int accumulate, void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout,
                            int kh, int kw, int stride, int pad, int ldx, int ldy, hipStream_t st) {
  SRHIP_REQUIRE(nprob >= 2 && nprob <= 4, "conv2d_wgrad_multi: 2..4 problems per launch");
  const int rowtap = fast_wgrad_multi_ok(cin, cout, kh, kw, stride, pad);
  SRHIP_REQUIRE(rowtap != 0, "conv2d_wgrad_multi: shape / arithmetic mode not served by the row-tap kernel");
  WgradGeom g;
  g.N = n; g.H = h; g.W = w; g.C = cin; g.ldx = ldx;
  g.Ho = h; g.Wo = w;                                     // stride 1, pad 1, 3 x 3
  g.K = cout; g.ldy = ldy; g.KH = kh; g.KW = kw; g.stride = stride; g.pad = pad;
  const long P = (long)n * g.Ho * g.Wo;
  SRHIP_REQUIRE(P < (1L << 31) - 64, "conv2d_wgrad_multi: pixel count overflows int32");
  g.P = (int)P; g.Ktot = kh * kw * cin;
  SRHIP_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "conv2d_wgrad_multi: row strides % 4 == 0");
  SRHIP_REQUIRE(bytes_ok((long)n * h * w, ldx, cin, &g.x_bytes) && bytes_ok(P, ldy, cout, &g.dy_bytes), "conv2d_wgrad_multi: tensor >= 2 GiB");
  FastWgradPlan p = plan_fast_wgrad(P, cout, g.Ktot, rowtap);
  const int ns = multi_nsplit(p, nprob);
  SRHIP_REQUIRE(ns > 0, "conv2d_wgrad_multi: problem too small to share a launch (use srhip_conv2d_wgrad)");
  const int nseg = cdiv(g.Wo, 16);
  const int rem = g.Wo & 15;
  const int tail_rem = (rem > 0 && rem <= 8 && g.Wo >= 16 && g.Ho >= 2 && g_wgrad_cfg != 9) ? rem : 0;
  const int nchunks_rt = tail_rem ? ((g.N * g.Ho + 1) / 2) * (2 * (g.Wo / 16) + 1) : g.N * g.Ho * nseg;
  const int cps = cdiv(nchunks_rt, ns);
  g.nsplit = ns; g.chunks_per_split = cps;
  const size_t per = (size_t)ns * ((size_t)cout * g.Ktot + cout);
  SRHIP_REQUIRE(workspace && workspace_bytes >= per * nprob * sizeof(float), "conv2d_wgrad_multi: workspace too small");
  WgradBatch bt;
  bt.nprob = nprob;
  const int tiles = cdiv(cout, p.bm) * cdiv(g.Ktot, p.bn);
  bt.bpp = tiles * ns;
  bool any_bias = false;
  for (int i = 0; i < 4; ++i) {
    const int k = i < nprob ? i : 0;
    SRHIP_REQUIRE(x[k] && dy[k] && dw[k] && ((((uintptr_t)x[k]) | ((uintptr_t)dy[k])) & 15) == 0, "conv2d_wgrad_multi: null / unaligned tensor");
    float* part = static_cast<float*>(workspace) + per * k;
    bt.x[i] = x[k];
    bt.dy[i] = dy[k];
    bt.partial[i] = part;
    bt.bias_partial[i] = (db && db[k]) ? part + (size_t)ns * cout * g.Ktot : nullptr;
    any_bias = any_bias || bt.bias_partial[i] != nullptr;
  }
  const int blocks = bt.bpp * nprob;
#define SRHIP_RTM(BM_, CIS_, SP_)                                                                                        \
  do {                                                                                                                 \
    if (g_rowtap_pipe)                                                                                                 \
      hipLaunchKernelGGL((wgrad_rowtap_kernel<BM_, CIS_, SP_, 1, true>), dim3(blocks), dim3(256), 0, st, bt.x[0], bt.dy[0], \
                         bt.partial[0], bt.bias_partial[0], g, nseg, cps, tail_rem, bt);                               \
    else if (g_rowtap_addr)                                                                                            \
      hipLaunchKernelGGL((wgrad_rowtap_kernel<BM_, CIS_, SP_, 1>), dim3(blocks), dim3(256), 0, st, bt.x[0], bt.dy[0],  \
                         bt.partial[0], bt.bias_partial[0], g, nseg, cps, tail_rem, bt);                               \
    else                                                                                                               \
      hipLaunchKernelGGL((wgrad_rowtap_kernel<BM_, CIS_, SP_, 0>), dim3(blocks), dim3(256), 0, st, bt.x[0], bt.dy[0],  \
                         bt.partial[0], bt.bias_partial[0], g, nseg, cps, tail_rem, bt);                               \
  } while (0)
  if (rowtap == 1 && g_conv_math == 2) SRHIP_RTM(128, 64, false);
  else if (rowtap == 1) SRHIP_RTM(128, 64, true);
  else if (g_conv_math == 2) SRHIP_RTM(64, 128, false);
  else SRHIP_RTM(64, 128, true);
#undef SRHIP_RTM
  int rc = check_launch("fast_wgrad_multi");
  if (rc) return rc;
  if (launch_reduce4(nprob, bt.partial, bt.bias_partial, dw, db, ns, cout, cin, kh * kw, g.Ktot, accumulate, st))
    return check_launch("fast_wgrad_multi_reduce4");
  for (int k = 0; k < nprob; ++k) {
    float* dbk = db ? db[k] : nullptr;
    const long total = (long)cout * g.Ktot + (dbk ? cout : 0);
    if (ns >= 64)
      hipLaunchKernelGGL(fast_wgrad_reduce_kernel<16>, dim3(cdiv(total, 64)), dim3(1024), 0, st, bt.partial[k], bt.bias_partial[k], dw[k], dbk, ns, cout, cin,
                         kh * kw, g.Ktot, accumulate);
    else
      hipLaunchKernelGGL(fast_wgrad_reduce_kernel<4>, dim3(cdiv(total, 64)), dim3(256), 0, st, bt.partial[k], bt.bias_partial[k], dw[k], dbk, ns, cout, cin,
                         kh * kw, g.Ktot, accumulate);
  }
  return check_launch("fast_wgrad_multi_reduce");
}

}  // namespace srhip
