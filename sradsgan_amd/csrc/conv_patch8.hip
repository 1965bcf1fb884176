// Stride-1 3x3 convolution, split-bf16, >= 256 destination channels: the patch kernel as ONE 8-wave block per CU whose two wave
// groups run one barrier apart (round 5; same idea as wgrad_flat8_kernel, the guide's 256^2 8-phase GEMM template).
//
// conv_patch_pers_kernel (conv_patch_pers.hip) puts three independent 4-wave blocks on a CU; every wave alternates, per tap, DMA issue /
// in-place split / 8 fragment reads / 12 MFMAs behind one barrier.  Co-resident waves run the same phases at the same time, and the
// matrix pipe is busy 54 % of the kernel (profiles/r04_roofline_pmc_summary.txt).  Here:
//   * a block owns 128 output pixels x 256 destination channels: group g (waves 4 g .. 4 g + 3) multiplies the SAME input patch with
//     its own 128 channels of weights -- the patch is fetched and split once for both (half the A traffic and conversions per MFMA);
//   * work is cut into PHASES of three taps (one filter row kh of one 16-channel chunk): a group's LOAD part (issue the next phase's
//     weight tiles and a piece of the next chunk's patch, split the pieces that landed, read the three taps' patch fragments and the
//     first tap's weight fragments) and its MULTIPLY part (36 MFMAs per wave, the next tap's weight fragments fetched under the
//     current tap's MFMAs) are separated by
//     block-wide barriers, and group 1 runs one barrier behind group 0: while one group multiplies, the other loads;
//   * two barriers per 36 MFMAs instead of one per 12; the weight ring needs two slots per group (the slot a phase fills was read
//     two barriers earlier), 96 KiB, + two patch buffers + a private epilogue staging area per wave: 152 KiB, one block per CU.
// Product order, chunk order and arithmetic are those of the 4-wave kernels: results are bit-identical to them.
// Destination: padded split-bf16 planes (DSTPP: the RAB's conv1 / conv2-dgrad) or fp32 NHWC; epilogue flags at run time.
#include "conv_dev.h"

namespace srhip {

template <int DSTPP>
__global__ __launch_bounds__(512) void conv_patch8_kernel(const float* __restrict__ src, const float* __restrict__ wt,
                                                          const float* __restrict__ bias, const float* __restrict__ actmask,
                                                          float* __restrict__ dst, FastGeom g, PatchGeom pg, int nblk_m, int nblk_n2,
                                                          unsigned dst_bytes, int ndst16, int abl) {   // abl: timing-only ablation bits (srhip_debug_set(16, bits)): 1 no MFMAs, 2 no DMAs, 4 no split, 8 no epilogue, 16 no fragment reads
  constexpr int BK = 16, WTM = 64, WTN = 64, TM = 2, TN = 2;
  constexpr int PATCH_B = 12 * 1024;                 // 192 patch rows of 64 bytes
  constexpr int BTAP_B = 128 * 64;                   // one tap's weight tile of a group
  constexpr int BSLOT_B = 3 * BTAP_B;                // a phase = three taps
  constexpr int RING0 = 2 * PATCH_B;                 // [group][slot]
  constexpr int STG0 = RING0 + 4 * BSLOT_B;          // per-wave epilogue staging, 16 rows x 64 floats
  constexpr int STG_B = 16 * WTN * 4;
  constexpr int LDS_B = STG0 + 8 * STG_B;
  __shared__ __attribute__((aligned(1024))) char lds[LDS_B];
  __shared__ int pix_tab[128];
  __shared__ unsigned rel_tab[128];
  __shared__ __attribute__((aligned(16))) float bias_s[1024];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  const int ntiles = nblk_m * nblk_n2;
  const int tpi = pg.tiles_h * pg.tiles_w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int flags = g.flags & 0x3f;
  if (tid < 128) {
    int pr_, pc_;
    patch_pixel(tid, pg.PH, pg.PW, pg.gmap, pr_, pc_);
    pix_tab[tid] = (pr_ << 16) | pc_;
    rel_tab[tid] = pr_ < pg.PH ? (DSTPP ? (unsigned)((pr_ * (g.Wd + 1) + pc_) * g.K) * 4u : (unsigned)((pr_ * g.Wd + pc_) * g.ldd) * 4u) : F_OOB;
  }
  if (flags & SRHIP_EPI_BIAS)
    for (int i = tid; i < g.K; i += 512) bias_s[i] = bias[i];
  __syncthreads();

  // ---- DMA addressing
  const int swz = (lane >> 4) & 3;
  const int aq = (lane & 3) ^ swz;                  // 16-byte quad of the 64-byte patch row this lane's slot holds
  int pij[2];                                       // patch coordinates of the rows this wave feeds: pieces wave (all) and 8 + wave (waves 0-3)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int row = (k * 8 + wave) * 16 + (lane >> 2);
    pij[k] = -1;
    if ((k == 0 || wave < 4) && row < pg.PR) {
      const int pi = row / pg.PWP;
      pij[k] = (pi << 16) | (row - pi * pg.PWP);
    }
  }
  const int CC = g.C / BK;
  const int wchunk = 9 * ndst16 * 64;               // bytes of one 16-channel chunk of the tiled weight image ([tap][n] rows of 64 bytes)
  int wtap[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wtap[t] = ((g.kh0 + (t / 3) * g.khs) * g.KW + (g.kw0 + (t % 3) * g.kws)) * ndst16 * 64;
  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, g.src_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, g.w_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(dst, 0, dst_bytes, 0x00020000);

  struct TileAt {
    int img, oh0, ow0, n0;                          // n0: first destination channel of THIS GROUP's 128 (-1: no tile)
  };
  auto decode = [&](int v) {
    const int tile = xcd_tile(v, ntiles);
    const int tile_n = tile % nblk_n2, pid = tile / nblk_n2;
    TileAt t;
    t.n0 = tile_n * 256 + grp * 128;
    t.img = pid / tpi;
    const int prem = pid - t.img * tpi;
    const int ty = prem / pg.tiles_w, tx = prem - ty * pg.tiles_w;
    t.oh0 = ty * pg.PH;
    t.ow0 = tx * pg.PW;
    return t;
  };
  unsigned aoffb[2], boffb[2];
  auto set_a = [&](const TileAt& t) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pi = pij[k] >> 16, pj = pij[k] & 0xffff;
      const int sh = t.oh0 + pg.lo_h + pi, sw = t.ow0 + pg.lo_w + pj;
      const bool ok = t.n0 >= 0 && pij[k] >= 0 && sh >= 0 && sh < g.Hs && sw >= 0 && sw < g.Ws;
      aoffb[k] = ok ? (unsigned)(((t.img * g.Hs + sh) * g.Ws + sw) * g.lds + aq * 4) * 4u : F_OOB;
    }
  };
  auto set_b = [&](const TileAt& t) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = t.n0 + w4 * 32 + 16 * j + (lane >> 2);
      boffb[j] = (t.n0 >= 0 && n < g.K) ? (unsigned)(n * 64 + (lane & 3) * 16) : F_OOB;   // rows are stored pre-swizzled
    }
  };
  const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);                     // + buffer * PATCH_B + k * 8192
  const unsigned b_dst = __builtin_amdgcn_readfirstlane(lds_base + RING0 + grp * 2 * BSLOT_B + w4 * 2048);   // + slot * BSLOT_B + tap * BTAP_B + j * 1024
  auto issue_a = [&](int buf, int k, unsigned coff) {
    if (k == 0 || wave < 4) lds_dma16_buf((abl & 2) ? F_OOB : aoffb[k] + coff, rs_a, a_dst + buf * PATCH_B + k * 8192);
  };
  auto issue_b3 = [&](int slot, int r, int cc) {    // the three weight tiles of phase (chunk cc, filter row r)
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
      const unsigned wk = (unsigned)(wtap[0] + cc * wchunk);
      const unsigned wo = r == 0 ? (unsigned)(wtap[tt] - wtap[0]) : r == 1 ? (unsigned)(wtap[3 + tt] - wtap[0]) : (unsigned)(wtap[6 + tt] - wtap[0]);
#pragma unroll
      for (int j = 0; j < 2; ++j) lds_dma16_buf((abl & 2) ? F_OOB : boffb[j] + wk + wo, rs_b, b_dst + slot * BSLOT_B + tt * BTAP_B + j * 1024);
    }
  };
  // fp32 -> split bf16 in place for one patch piece this wave fetched (conv_patch_pers_kernel's convert_piece: same roundings)
  auto convert_piece = [&](int buf, int k) {
    if (!(k == 0 || wave < 4) || (abl & 4)) return;
    float4* slot = reinterpret_cast<float4*>(lds + buf * PATCH_B + (k * 8 + wave) * 1024 + lane * 16);
    const float4 own = *slot;
    const bool odd = aq & 1;
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t h01 = {(__bf16)own.x, (__bf16)own.y}, h23 = {(__bf16)own.z, (__bf16)own.w};
    const unsigned uh01 = __builtin_bit_cast(unsigned, h01), uh23 = __builtin_bit_cast(unsigned, h23);
    const bf16x2_t l01 = {(__bf16)(own.x - __uint_as_float(uh01 << 16)), (__bf16)(own.y - __uint_as_float(uh01 & 0xffff0000u))};
    const bf16x2_t l23 = {(__bf16)(own.z - __uint_as_float(uh23 << 16)), (__bf16)(own.w - __uint_as_float(uh23 & 0xffff0000u))};
    const unsigned ul01 = __builtin_bit_cast(unsigned, l01), ul23 = __builtin_bit_cast(unsigned, l23);
    const unsigned s0 = odd ? uh01 : ul01, s1 = odd ? uh23 : ul23;
    const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xF, 0xF, true);
    const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xF, 0xF, true);
    u32x4 o;
    o.x = odd ? r0 : uh01;
    o.y = odd ? r1 : uh23;
    o.z = odd ? ul01 : r0;
    o.w = odd ? ul23 : r1;
    *reinterpret_cast<u32x4*>(slot) = o;
  };

  // ---- fragment addressing
  const int wm = w4 >> 1, wn = w4 & 1;
  const int khalf = lane >> 5, l31 = lane & 31;
  int arow[TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int pt = pix_tab[wm * WTM + t * 32 + l31];
    const int orow = pt >> 16, ocol = pt & 0xffff;
    arow[t] = orow < pg.PH ? orow * pg.PWP + ocol : 0;
  }
  int tapsh[9];                                     // patch-row shift of a tap (scalars)
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
    tapsh[tap] = ((g.dh0 + (tap / 3) * g.dhs) - pg.lo_h) * pg.PWP + ((g.dw0 + (tap % 3) * g.dws) - pg.lo_w);
  int boff[TN];
#pragma unroll
  for (int u = 0; u < TN; ++u) {
    const int row = wn * WTN + u * 32 + l31;
    boff[u] = RING0 + grp * 2 * BSLOT_B + row * 64 + (((2 * khalf) ^ ((row >> 2) & 3)) << 4);
  }
  struct AFrags {
    bf16x8_t ah[TM], al[TM];
  };
  struct BFrags {
    bf16x8_t bh[TN], bl[TN];
  };
  // Patch fragments are read in the LOAD part only: the other group, one barrier ahead, may already be fetching the next chunk's
  // patch into the buffer this group's previous phase used -- a patch read in the multiply part would race with it.  Weight
  // fragments come from the group's own ring and may be read at any time between the group's barriers.
  auto load_a = [&](AFrags& f, int pbuf, int tap) {
    if (abl & 16) return;
    const char* pb = lds + pbuf * PATCH_B;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int pr = arow[t] + tapsh[tap];
      const int ao = pr * 64 + (((2 * khalf) ^ ((pr >> 2) & 3)) << 4);
      f.ah[t] = *reinterpret_cast<const bf16x8_t*>(pb + ao);
      f.al[t] = *reinterpret_cast<const bf16x8_t*>(pb + (ao ^ 16));
    }
  };
  auto load_b = [&](BFrags& f, int slot, int tt) {
    if (abl & 16) return;
    const char* sb = lds + slot * BSLOT_B + tt * BTAP_B;
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      f.bh[u] = *reinterpret_cast<const bf16x8_t*>(sb + boff[u]);
      f.bl[u] = *reinterpret_cast<const bf16x8_t*>(sb + (boff[u] ^ 16));
    }
  };
  f32x16 acc[TM][TN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
      for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  };
  zero_acc();
  auto mfma12 = [&](const AFrags& a, const BFrags& b) {   // product order of conv_patch_kernel: al*bh, ah*bl, ah*bh over the four tiles
    if (abl & 1) return;
#pragma unroll
    for (int i = 0; i < 3 * TM * TN; ++i) {
      const int pr = i / (TM * TN), t = (i % (TM * TN)) / TN, u = i % TN;
      acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pr == 0 ? a.al[t] : a.ah[t], pr == 1 ? b.bl[u] : b.bh[u], acc[t][u], 0, 0, 0);
    }
  };

  // ---- epilogue of one tile (no barrier: the staging area is this wave's own)
  auto epilogue = [&](const TileAt& t) {
    float* wl = reinterpret_cast<float*>(lds + STG0 + wave * STG_B);
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int l31e = ln & 31, khe = ln >> 5;
    constexpr int OPR = WTN / 8;                    // 8-channel groups per staged row
    constexpr int NRP = 16 * OPR / 64;              // rows per lane per pass
    const int oq = ln & (OPR - 1), rsub = ln / OPR;
    const int n = t.n0 + wn * WTN + oq * 8;
    const bool nok = t.n0 >= 0 && n < g.K;
    const int ns = nok ? n : 0;
    const unsigned tile_base = DSTPP ? (unsigned)((g.dst_guard + (t.img * (g.Hd + 1) + t.oh0) * (g.Wd + 1) + t.ow0) * g.K + n) * 4u
                                     : (unsigned)(((t.img * g.Hd + t.oh0) * g.Wd + t.ow0) * g.ldd + n) * 4u;
    const bool interior = t.oh0 + pg.PH <= g.OH && t.ow0 + pg.PW <= g.OW;
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (flags & SRHIP_EPI_BIAS) {
      b0 = *reinterpret_cast<const float4*>(bias_s + ns);
      b1 = *reinterpret_cast<const float4*>(bias_s + ns + 4);
    }
    auto pass_offsets = [&](int p, unsigned (&doff)[NRP], bool (&okv)[NRP]) {
#pragma unroll
      for (int i = 0; i < NRP; ++i) {
        const int rr = wm * WTM + (p >> 1) * 32 + (p & 1) * 16 + i * (64 / OPR) + rsub;
        const unsigned rel = rel_tab[rr];
        bool ok = nok && rel < F_OOB;
        if (!interior) {
          const int pt = pix_tab[rr];
          ok = ok && t.oh0 + (pt >> 16) < g.OH && t.ow0 + (pt & 0xffff) < g.OW;
        }
        okv[i] = ok;
        doff[i] = ok ? tile_base + rel : 0u;
      }
    };
    // activation mask (dgrad of a conv whose producer's LeakyReLU output is kept): planes -> the 8 hi halves; fp32 -> two float4
    u32x4 am[2][NRP][2];
    unsigned doffs[2][NRP];
    bool oks[2][NRP];
    auto load_mask = [&](int b) {
#pragma unroll
      for (int i = 0; i < NRP; ++i) {
        am[b][i][0] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(actmask) + doffs[b][i]);
        if (!DSTPP) am[b][i][1] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(actmask) + doffs[b][i] + 16);
      }
    };
    pass_offsets(0, doffs[0], oks[0]);
    if (flags & SRHIP_EPI_ACTMASK) load_mask(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int tt = p >> 1, rb = (p & 1) * 8, cur = p & 1, nx = cur ^ 1;
#pragma unroll
      for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) wl[((r8 & 3) + 8 * (r8 >> 2) + 4 * khe) * WTN + u * 32 + l31e] = acc[tt][u][rb + r8];
      if (p < 3) {
        pass_offsets(p + 1, doffs[nx], oks[nx]);
        if (flags & SRHIP_EPI_ACTMASK) load_mask(nx);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NRP; ++i) {
        const int row = i * (64 / OPR) + rsub;
        const float4 v0 = *reinterpret_cast<const float4*>(wl + row * WTN + oq * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(wl + row * WTN + oq * 8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (flags & SRHIP_EPI_BIAS) {
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (flags & SRHIP_EPI_LRELU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * g.slope;
        }
        if (flags & SRHIP_EPI_ACTMASK) {
          if (DSTPP) {
            const unsigned mv[4] = {am[cur][i][0].x, am[cur][i][0].y, am[cur][i][0].z, am[cur][i][0].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float a = __uint_as_float((j & 1) ? (mv[j >> 1] & 0xffff0000u) : (mv[j >> 1] << 16));
              v[j] = a > 0.f ? v[j] : v[j] * g.slope;
            }
          } else {
            const unsigned mv[8] = {am[cur][i][0].x, am[cur][i][0].y, am[cur][i][0].z, am[cur][i][0].w,
                                    am[cur][i][1].x, am[cur][i][1].y, am[cur][i][1].z, am[cur][i][1].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __uint_as_float(mv[j]) > 0.f ? v[j] : v[j] * g.slope;
          }
        }
        const unsigned doff = doffs[cur][i];
        const bool ok = oks[cur][i];
        const unsigned dead = F_OOB + 32u * (unsigned)(p * NRP + i);
        if (DSTPP) {
          bf16x8_t hi, lo;
          split_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, hi), rs_d, ok ? doff : dead, 0, 2);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, lo), rs_d, ok ? doff + 16u : dead + 16u, 0, 2);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(v[0], v[1], v[2], v[3])), rs_d, ok ? doff : dead, 0, 2);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, make_float4(v[4], v[5], v[6], v[7])), rs_d, ok ? doff + 16u : dead + 16u, 0, 2);
        }
      }
      if (p < 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    zero_acc();
  };

  // ---- the walk.  Global phase counter ph: ring slot = ph & 1; global chunk counter gc: patch buffer = gc & 1.
  TileAt cur = decode(blockIdx.x), nxt = cur;
  set_a(cur);
  set_b(cur);
  issue_a(0, 0, 0u);
  issue_a(0, 1, 0u);
  issue_b3(0, 0, 0);
  wait_vmcnt<0>();
  convert_piece(0, 0);
  convert_piece(0, 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) {                                   // group 1 runs one barrier behind
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  int ph = 0, gc = 0;
  auto barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // one phase: filter row R of chunk cc.  nb_*: the phase whose weights are fetched now (the next one); na_*: the chunk whose patch is
  // being fetched / split now (the next one).
  auto phase = [&](auto rc, int cc, int nb_r_is0_cc, unsigned na_coff) {
    constexpr int R = decltype(rc)::value;
    const int slot = ph & 1, pbuf = gc & 1;
    barrier();                                                          // A: the phase's operands are in LDS, the slot / buffer filled below are free
    // ---- load part
    if (R < 2) issue_b3(slot ^ 1, R + 1, cc);
    else issue_b3(slot ^ 1, 0, nb_r_is0_cc);
    if (R < 2) issue_a(pbuf ^ 1, R, na_coff);
    if (R == 2) {
      convert_piece(pbuf ^ 1, 0);
      convert_piece(pbuf ^ 1, 1);
    }
    AFrags a0 = {}, a1 = {}, a2 = {};
    BFrags b0 = {}, b1 = {};
    load_a(a0, pbuf, 3 * R);
    load_b(b0, slot, 0);
    load_a(a1, pbuf, 3 * R + 1);
    load_a(a2, pbuf, 3 * R + 2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    barrier();                                                          // B
    // ---- multiply part: the next tap's weight fragments are fetched under the current tap's MFMAs
    __builtin_amdgcn_s_setprio(1);
    load_b(b1, slot, 1);
    mfma12(a0, b0);
    load_b(b0, slot, 2);
    mfma12(a1, b1);
    mfma12(a2, b0);
    __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wait_vmcnt<0>();                                                    // this wave's DMAs for the next phase have landed
    ++ph;
  };
  for (int vt = blockIdx.x; vt < ntiles; vt += (int)gridDim.x) {
    const int vn = vt + (int)gridDim.x;
    if (vn < ntiles) nxt = decode(vn);
    else nxt.n0 = -1;
    for (int cc = 0; cc < CC; ++cc) {
      const bool last = cc + 1 == CC;
      if (last) {                                   // the next chunk is the next tile's first: its patch and (from row 2 on) its weights
        set_a(nxt);
      }
      const unsigned na_coff = last ? 0u : (unsigned)((cc + 1) * BK * 4);
      phase(IC<0>(), cc, 0, na_coff);
      phase(IC<1>(), cc, 0, na_coff);
      if (last) set_b(nxt);
      phase(IC<2>(), cc, last ? 0 : cc + 1, na_coff);
      ++gc;
    }
    if (!(abl & 8)) epilogue(cur);
    cur = nxt;
  }
  if (grp == 0) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  wait_vmcnt<0>();
}

int g_patch8_abl = 0;      // srhip_debug_set(16, bits)
// NOT the default (round 5 measurement, tools/ablate_patch8.py, RAB conv1 fprop B = 32): 87-94 us against the 4-wave kernel's 74-90 on the
// same box (0.38 against 0.44 of the split-bf16 ceiling).  What the ablations say: the MFMAs alone 51 us, the load part without MFMAs
// 63 us -- a wave issues 6 weight DMAs + 1 patch DMA per 36 MFMAs, ~100 cycles each, plus 16 fragment reads and the split: longer than
// the partner's multiply part, so the groups wait for each other's LOAD parts --, and the epilogue 26 us: 256 blocks reach their tile
// ends together, 32 MB leave the chip in one burst three times per launch and every wave's next vmcnt(0) waits for its stores.  The
// weight-gradient kernel gains from the same structure (18 MFMAs per 3 DMAs and no per-tile epilogue); here the 4-wave walk with
// three free-running blocks per CU hides the same costs better.  Kept for the record and as a test subject (bit-identical).
int g_patch8 = 0;          // srhip_debug_set(15, v): 1 = take this kernel where it applies (>= 256 destination channels, >= one tile per CU)
static int num_cu8() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}

// -1: not applicable (the caller takes conv_patch_pers_kernel)
int launch_patch8(const float* src, const float* wt, const float* bias, const float* actmask, float* dst, const FastGeom& g_,
                  const PatchGeom& pg, int nbm, int eflags, hipStream_t st) {
  FastGeom g = g_;
  if (!g_patch8 || g_sign_req.mode != 0 || g.src_pp || g.K < 256 || g.K % 8 != 0 || g.C % 16 != 0 || g.K > 1024) return -1;
  if (eflags & ~(SRHIP_EPI_BIAS | SRHIP_EPI_LRELU | SRHIP_EPI_ACTMASK)) return -1;
  if ((eflags & SRHIP_EPI_ACTMASK) && !actmask) return -1;
  const size_t total = (size_t)(g.w_bytes >> 2);
  const int ndst16 = (g.K + 15) / 16 * 16;
  const long tb = (long)9 * g.C * ndst16 * 4L;
  if (tb >= (1L << 31)) return -1;
  const float* wsplit = wt + total * 3;             // the tiled split-bf16 section of the packed weight (conv_internal.h)
  g.w_bytes = (unsigned)tb;
  const long dbytes = g.dst_pp ? 2L * g.dst_plane_bytes : ((long)g.N * g.Hd * g.Wd - 1) * (long)g.ldd * 4L + (long)g.K * 4L;
  if (dbytes >= (1L << 31)) return -1;
  const int nbn2 = (g.K + 255) / 256;
  const long ntiles = (long)nbm * nbn2;
  if (ntiles < num_cu8()) return -1;                // fewer tiles than CUs: the 4-wave kernels fill the chip better
  int grid = num_cu8();
  grid -= grid % 8;
  if (g.dst_pp)
    hipLaunchKernelGGL((conv_patch8_kernel<1>), dim3(grid), dim3(512), 0, st, src, wsplit, bias, actmask, dst, g, pg, nbm, nbn2, (unsigned)dbytes, ndst16, g_patch8_abl);
  else
    hipLaunchKernelGGL((conv_patch8_kernel<0>), dim3(grid), dim3(512), 0, st, src, wsplit, bias, actmask, dst, g, pg, nbm, nbn2, (unsigned)dbytes, ndst16, g_patch8_abl);
  return check_launch("conv_patch8");
}

}  // namespace srhip
