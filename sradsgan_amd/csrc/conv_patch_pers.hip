// Persistent form of the stride-1 3x3 patch kernel (conv_fast.hip: conv_patch_kernel) for gfx950.
//
// conv_patch_kernel runs one 128-pixel x BN tile per block: at the bench shape 1536 blocks on 768 slots, i.e. two
// lock-stepped rounds, each paying its own cold prologue (descriptor / address set-up, the patch and the first weight
// tiles fetched with nothing to hide them) and its own epilogue (every block of the chip stores its 64 KB at the same
// time while the matrix pipes idle).  Here a block stays resident and walks tiles v = blockIdx.x, + gridDim.x, ...:
//   * the tap stream never stops at a tile boundary: during the LAST chunk of a tile the A patch of the next tile's first
//     chunk and its first two weight tiles are already being fetched (taps 0..2 / 7..8 of the unrolled schedule), so the
//     next tile's first MFMA follows the epilogue directly;
//   * the epilogue stores are buffer stores with an out-of-range offset for dead rows, so every wave issues exactly NS
//     stores per tile: they stay in flight under the next tile's taps and the counted vmcnt waits of taps 0 and 1 simply
//     allow NS more operations (vmcnt completes in order on gfx9-class memory pipelines);
//   * the accumulators are staged through LDS in four 16-row passes inside the patch buffer / ring slot that the last tap
//     has just released (the next tile's prefetch occupies the others), so the block still needs 50.5 KB: 3 blocks per CU.
//     (Tried and kept as ablation 64: MFMA operand roles swapped -- weights = rows, pixels = columns -- so that a lane holds 4
//     consecutive channels of one pixel and stores them itself, no LDS transpose, a third of the address arithmetic.  Same
//     time in split-bf16 and TWICE the HBM write traffic: a wave instruction then writes 32 B per pixel and the memory side
//     counts 212 MB for 95.6 MB of output (profiles/r04_roofline_pmc_direct_epilogue.txt); the store-bound `half` kernel ran
//     102 us instead of 55.)
// Arithmetic, product order and chunk order are those of conv_patch_kernel: results are bit-identical to it (and so to
// fast_conv_dma_kernel<.., MATH 1>).
#include "conv_dev.h"

namespace srhip {

extern int g_fast_ablate;

// ablation helpers: keep a value alive / make it opaque without an instruction (vector-register constraints only exist in the device pass)
#if defined(__HIP_DEVICE_COMPILE__)
#define KEEP_IN(x) asm volatile("" ::"v"(x))
#define KEEP_OUT(x) asm volatile("" : "=v"(x))
#else
#define KEEP_IN(x) (void)(x)
#define KEEP_OUT(x) (void)(x)
#endif

// ABL: timing-only ablations (wrong results), srhip_debug_set(6, bits) on the <128, bias+lrelu> fprop: 1 stores dropped (out-of-range
// offsets: issued, never written), 2 no in-place conversion, 4 no MFMAs, 8 no fragment reads, 16 no B DMA in the loop, 32 no epilogue
// POOL (round 4, BN = 64 with K <= 64 only): the epilogue also reduces the tile's final outputs per channel -- sum, NaN-propagating
// maximum, first arg-max pixel -- and writes one partial per (image, tile, wave row): the CLAM pooling partials of the RAB tail
// (clam_pool_partial_kernel's job: a 24 MB read and a launch per RAB) as three more counted stores of the producing conv.
// pool_out: [3 sections: sum | max | arg][image][2 * tiles per image][64], section stride pool_sec bytes.
// SRCPP / DSTPP (round 5): the source / the destination are padded split-bf16 planes (conv_wgrad_flat.hip).  A pp source arrives as
// the very LDS image the in-place split would have left (a lane's 16-byte quad = 8 hi or 8 lo halves of one pixel), so the split and
// its LDS round trip are skipped; a pp destination takes the epilogue's fp32 rows as 8-channel hi | lo stores (the same bytes as the
// fp32 row), and its activation mask (the dgrad of a conv whose producer's LeakyReLU output is kept as planes) is read from the
// mask tensor's hi plane at the same offsets.
// SIGNS (round 5, BN = 128 with a pp destination): the LeakyReLU mask as SIGN WORDS instead -- one 64-bit word per (tile, wave, lane):
// the signs of the 8 items x 8 channels that lane converts in the epilogue (bit (p * NRP + i) * 8 + j).  1 = the bias + LeakyReLU
// forward WRITES them (one more counted store per tile; `actmask` is the word buffer), 2 = the masked data gradient READS them (one
// 8-byte load per lane and tile instead of eight 16-byte loads of the producer's hi plane: 3 MB instead of 48 at the bench shape).
// Producer and consumer must walk the same tiles: same image size, same channel count, both BN = 128 (srhip_conv2d_pp_sign_bytes).
template <int BN, int EPI, int PROD = 0, int ABL = 0, int POOL = 0, int SRCPP = 0, int DSTPP = 0, int SIGNS = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void conv_patch_pers_kernel(
    const float* __restrict__ src, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ residual, const float* __restrict__ actmask, float* __restrict__ dst, FastGeom g,
    PatchGeom pg, int nblk_m, int nblk_n, unsigned dst_bytes, int ndst16, float* __restrict__ pool_out, unsigned pool_sec) {
  static_assert(POOL == 0 || (BN == 64 && PROD == 0 && ABL == 0), "pooling epilogue: 64-wide tile, split-bf16");
  static_assert((SRCPP == 0 && DSTPP == 0) || (PROD == 0 && ABL == 0), "padded planes: split-bf16 only");
  static_assert(DSTPP == 0 || POOL == 0, "a pp destination has no pooling epilogue");
  static_assert(SIGNS == 0 || (BN == 128 && DSTPP == 1 && EPI == (SIGNS == 1 ? 3 : 32)), "sign words: 128-wide tile onto planes, bias + LeakyReLU forward / masked data gradient");
  constexpr bool TILED = PROD == 0;                 // B tiles from the tiled section of the packed weight (conv_internal.h)
  constexpr int XB = (ABL & 128) ? 2 : 1, XA = (ABL & 256) ? 2 : 1;   // ablations 128 / 256: every B / A DMA issued twice (marginal cost of the streams)
  constexpr bool DIRECT = (ABL & 64) != 0;          // ablation 64: epilogue straight from the accumulators with the MFMA operand roles swapped (see the header)
  constexpr int NW = 4, BK = 16;
  constexpr int WTM = 64, WTN = BN / 2;
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int BPW = BN / 64;                      // B DMA pieces per wave per tap
  constexpr int MAXP = 3;                           // A patch pieces per wave: 12 pieces = 192 rows per patch
  constexpr int PATCH_B = 12 * 1024;
  constexpr int BSTAGE_B = BN * 64;
  constexpr int RING0 = 2 * PATCH_B;
  constexpr int LDS_B = 2 * PATCH_B + 3 * BSTAGE_B;
  constexpr int QPRW = WTN / 4;                     // float4s per staged row
  constexpr int NRD = 16 * QPRW / 64;               // float4s per lane per 16-row pass
  constexpr int NS = 4 * NRD + (POOL ? 3 : 0) + (SIGNS == 1 ? 1 : 0);   // epilogue stores per wave per tile (always issued); POOL: + sum, max, arg partials; SIGNS 1: + the sign word
  constexpr int STG_B = 16 * WTN * 4;               // staging bytes per wave
  static_assert(3 * STG_B <= PATCH_B && STG_B <= BSTAGE_B, "epilogue staging must fit the released buffers");
  __shared__ __attribute__((aligned(1024))) char lds[LDS_B];
  __shared__ int pix_tab[128];                      // lane-row -> (orow << 16 | ocol)
  __shared__ unsigned rel_tab[128];                 // lane-row -> byte offset of its pixel inside the destination, relative to the tile's
                                                    // first pixel (dead rows: out of range): the epilogue adds one per-tile base
  __shared__ __attribute__((aligned(16))) float bias_s[512];   // the bias vector: the epilogue must not issue global loads of its own
                                                    // (hipcc would wait vmcnt(0) for them and serialise the stores behind each other)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = nblk_m * nblk_n;
  const int tpi = pg.tiles_h * pg.tiles_w;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  if (tid < 128) {
    int pr_, pc_;
    patch_pixel(tid, pg.PH, pg.PW, pg.gmap, pr_, pc_);
    pix_tab[tid] = (pr_ << 16) | pc_;
    rel_tab[tid] = pr_ < pg.PH ? (DSTPP ? (unsigned)((pr_ * (g.Wd + 1) + pc_) * g.K) * 4u : (unsigned)((pr_ * g.Wd + pc_) * g.ldd) * 4u) : F_OOB;
  }
  if ((EPI >= 0 ? EPI : g.flags) & SRHIP_EPI_BIAS)
    for (int i = tid; i < g.K; i += 256) bias_s[i] = bias[i];
  __syncthreads();

  // ---- tile-invariant part of the DMA addressing ----
  const int swz = (lane >> 4) & 3;
  const int aq = (lane & 3) ^ swz;                  // global 16-byte quad held by this lane's slot
  int pij[MAXP];                                    // patch coordinates (pi << 16 | pj) of the row this lane feeds, -1: past the patch
#pragma unroll
  for (int k = 0; k < MAXP; ++k) {
    const int row = (k * NW + wave) * 16 + (lane >> 2);
    pij[k] = -1;
    if (row < pg.PR) {
      const int pi = row / pg.PWP;
      pij[k] = (pi << 16) | (row - pi * pg.PWP);
    }
  }
  const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + wave * 1024);
  const unsigned b_dst = __builtin_amdgcn_readfirstlane(lds_base + RING0 + wave * BPW * 1024);
  const int CC = g.C / BK;
  int wtap[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int tapidx = (g.kh0 + (t / 3) * g.khs) * g.KW + (g.kw0 + (t % 3) * g.kws);
    wtap[t] = TILED ? tapidx * ndst16 * 64 : tapidx * g.C;   // TILED: byte offset of the tap's rows inside a chunk; else packed column
  }
  const int wchunk = 9 * ndst16 * 64;               // TILED: bytes of one 16-channel chunk ([tap][n] rows of 64 bytes)
  __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, g.src_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, g.w_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(dst, 0, dst_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(POOL ? pool_out : dst, 0, POOL ? 3u * pool_sec : 0u, 0x00020000);   // pooling partials (POOL)
  __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc(SIGNS ? const_cast<float*>(actmask) : dst, 0, SIGNS ? (unsigned)ntiles * 2048u : 0u, 0x00020000);   // sign words (SIGNS)

  // ---- per-tile addressing ----
  struct TileAt {
    int img, oh0, ow0, n0, id;
  };
  auto decode = [&](int v) {
    const int tile = xcd_tile(v, ntiles);
    const int tile_n = tile % nblk_n, pid = tile / nblk_n;
    TileAt t;
    t.id = tile;
    t.n0 = tile_n * BN;
    t.img = pid / tpi;
    const int prem = pid - t.img * tpi;
    const int ty = prem / pg.tiles_w, tx = prem - ty * pg.tiles_w;
    t.oh0 = ty * pg.PH;
    t.ow0 = tx * pg.PW;
    return t;
  };
  unsigned aoffb[MAXP], boffb[BPW];
  auto set_a = [&](const TileAt& t) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
      const int pi = pij[k] >> 16, pj = pij[k] & 0xffff;
      const int sh = t.oh0 + pg.lo_h + pi, sw = t.ow0 + pg.lo_w + pj;
      if (SRCPP) {       // the zero pad row / column / guard ARE the halo: only positions beyond them are out of range
        const bool ok = t.n0 >= 0 && pij[k] >= 0 && sh >= -1 && sh <= g.Hs && sw >= -1 && sw <= g.Ws;
        aoffb[k] = ok ? (unsigned)((g.src_guard + (t.img * (g.Hs + 1) + sh) * (g.Ws + 1) + sw) * g.C + aq * 4) * 4u : F_OOB;   // quad aq = 8 hi or 8 lo halves
        continue;
      }
      const bool ok = t.n0 >= 0 && pij[k] >= 0 && sh >= 0 && sh < g.Hs && sw >= 0 && sw < g.Ws;
      aoffb[k] = ok ? (unsigned)(((t.img * g.Hs + sh) * g.Ws + sw) * g.lds + aq * 4) * 4u : F_OOB;
    }
  };
  auto set_b = [&](const TileAt& t) {
#pragma unroll
    for (int j = 0; j < BPW; ++j) {
      const int n = t.n0 + wave * 16 * BPW + 16 * j + (lane >> 2);
      if (TILED) boffb[j] = (t.n0 >= 0 && n < g.K) ? (unsigned)(n * 64 + (lane & 3) * 16) : F_OOB;   // rows are stored pre-swizzled
      else boffb[j] = (t.n0 >= 0 && n < g.K) ? (unsigned)(n * g.ldw + aq * 4) * 4u : F_OOB;
    }
  };
  auto issue_a = [&](int buf, int k, unsigned coff) {   // one 1 KiB piece of a patch; coff = byte offset of the chunk's channels
    lds_dma16_buf(aoffb[k] + coff, rs_a, a_dst + buf * PATCH_B + k * (NW * 1024));   // (a chunk = 16 channels = 64 bytes, fp32 or padded planes)
  };
  auto issue_b = [&](int stage, int tap, int cc) {  // the B tile of (chunk cc, tap)
    const unsigned wk = TILED ? (unsigned)(wtap[tap] + cc * wchunk) : (unsigned)((wtap[tap] + cc * BK) * 4);
#pragma unroll
    for (int j = 0; j < BPW; ++j) lds_dma16_buf(boffb[j] + wk, rs_b, b_dst + stage * BSTAGE_B + j * 1024);
  };
  // fp32 -> split bf16 in place for one piece this wave fetched.  Lanes 2i, 2i+1 hold the two quads (8 consecutive channels)
  // of a half row; afterwards the even quad's slot holds the 8 hi halves, the odd one's the 8 lo halves (as conv_patch_kernel
  // leaves them).  Each lane splits only ITS four values (conv_patch_kernel: both lanes split all eight), then the pair swaps
  // what the other needs with one DPP quad permutation per register: 20 VALU per piece instead of 52, same roundings.
  auto convert_piece = [&](int buf, int k) {
    float4* slot = reinterpret_cast<float4*>(lds + buf * PATCH_B + (k * NW + wave) * 1024 + lane * 16);
    const float4 own = *slot;
    const bool odd = aq & 1;
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    if (PROD == 0) {
      const bf16x2_t h01 = {(__bf16)own.x, (__bf16)own.y}, h23 = {(__bf16)own.z, (__bf16)own.w};
      const unsigned uh01 = __builtin_bit_cast(unsigned, h01), uh23 = __builtin_bit_cast(unsigned, h23);
      const bf16x2_t l01 = {(__bf16)(own.x - __uint_as_float(uh01 << 16)), (__bf16)(own.y - __uint_as_float(uh01 & 0xffff0000u))};
      const bf16x2_t l23 = {(__bf16)(own.z - __uint_as_float(uh23 << 16)), (__bf16)(own.w - __uint_as_float(uh23 & 0xffff0000u))};
      const unsigned ul01 = __builtin_bit_cast(unsigned, l01), ul23 = __builtin_bit_cast(unsigned, l23);
      // the even lane keeps hi and needs the partner's hi; the odd lane keeps lo and needs the partner's lo
      const unsigned s0 = odd ? uh01 : ul01, s1 = odd ? uh23 : ul23;
      const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xF, 0xF, true);
      const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xF, 0xF, true);
      u32x4 o;
      o.x = odd ? r0 : uh01;
      o.y = odd ? r1 : uh23;
      o.z = odd ? ul01 : r0;
      o.w = odd ? ul23 : r1;
      *reinterpret_cast<u32x4*>(slot) = o;
    } else {
      float4 oth;
      oth.x = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.x), 0xB1, 0xF, 0xF, true));
      oth.y = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.y), 0xB1, 0xF, 0xF, true));
      oth.z = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.z), 0xB1, 0xF, 0xF, true));
      oth.w = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, own.w), 0xB1, 0xF, 0xF, true));
      if (!odd) *reinterpret_cast<bf16x8_t*>(slot) = round16x8<PROD>(own, oth);
    }
  };

  // ---- fragment addressing (tile-invariant) ----
  const int wm = wave >> 1, wn = wave & 1;
  const int khalf = lane >> 5, l31 = lane & 31;
  int aoff[9][TM];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int r = wm * WTM + t * 32 + l31;
    const int pt = pix_tab[r];
    const int orow = pt >> 16, ocol = pt & 0xffff;
    const int arow = orow < pg.PH ? orow * pg.PWP + ocol : 0;
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
    boff[u] = RING0 + row * 64 + (((2 * khalf) ^ ((row >> 2) & 3)) << 4);
  }
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

  // One tap of the endless tap stream.  PAR = patch buffer of this chunk (CC is even: chunk parity), FIRST = first chunk
  // of a tile: NS epilogue stores (or the prologue's NS dummies) sit between the two prefetched B tiles and this tap's DMAs.
  // Every chunk has a successor -- the next chunk of the tile, or the first chunk of the block's next tile (the caller has
  // pointed aoffb / a_coff / b_ncc at it; after the block's last tile these are out-of-range offsets: the DMAs deliver
  // zeros into free buffers and the block drains them before it ends) -- so there is ONE schedule and no special cases.
  auto do_tap = [&](auto tapc, auto firstc, auto parc, int cc, unsigned a_coff, int b_ncc) {
    constexpr int TAP = decltype(tapc)::value;
    constexpr bool FIRST = decltype(firstc)::value != 0;
    constexpr int pbuf = decltype(parc)::value;
    constexpr int NEWER = XB * BPW + ((TAP >= 1 && TAP <= MAXP) ? XA : 0) + ((FIRST && TAP < 2) ? NS : 0);
    wait_vmcnt<NEWER>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (TAP < MAXP) issue_a(pbuf ^ 1, TAP, a_coff);
    if (XA == 2 && TAP < MAXP) issue_a(pbuf ^ 1, TAP, a_coff);
    if (XB == 2) {
      if (TAP + 2 < 9) issue_b((TAP + 2) % 3, TAP + 2, cc);
      else issue_b((TAP + 2) % 3, TAP + 2 - 9, b_ncc);
    }
    if (ABL & 16) {
#pragma unroll
      for (int j = 0; j < BPW; ++j) lds_dma16_buf(F_OOB, rs_b, b_dst + ((TAP + 2) % 3) * BSTAGE_B + j * 1024);
    } else if (TAP + 2 < 9) issue_b((TAP + 2) % 3, TAP + 2, cc);
    else issue_b((TAP + 2) % 3, TAP + 2 - 9, b_ncc);
    if (!SRCPP && !(ABL & 2) && TAP >= 2 && TAP - 2 < MAXP) convert_piece(pbuf ^ 1, TAP - 2);
    const char* pb = lds + pbuf * PATCH_B;
    const char* sb = lds + (TAP % 3) * BSTAGE_B;
    bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
    int x16 = 16;                                       // opaque to the optimiser: keeps the 18 "lo" addresses out of registers
    asm volatile("" : "+s"(x16));
    if (ABL & 8) {
#pragma unroll
      for (int t = 0; t < TM; ++t) { KEEP_OUT(ah[t]); KEEP_OUT(al[t]); }
#pragma unroll
      for (int u = 0; u < TN; ++u) { KEEP_OUT(bh[u]); KEEP_OUT(bl[u]); }
    } else {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      ah[t] = *reinterpret_cast<const bf16x8_t*>(pb + aoff[TAP][t]);
      al[t] = PROD == 0 ? *reinterpret_cast<const bf16x8_t*>(pb + (aoff[TAP][t] ^ x16)) : ah[t];
    }
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      bh[u] = *reinterpret_cast<const bf16x8_t*>(sb + boff[u]);
      bl[u] = PROD == 0 ? *reinterpret_cast<const bf16x8_t*>(sb + (boff[u] ^ 16)) : bh[u];
    }
    }
    if (ABL & 4) {
#pragma unroll
      for (int t = 0; t < TM; ++t) { KEEP_IN(ah[t]); KEEP_IN(al[t]); }
#pragma unroll
      for (int u = 0; u < TN; ++u) { KEEP_IN(bh[u]); KEEP_IN(bl[u]); }
    } else
#pragma unroll
    for (int i = 0; i < nprod<PROD>() * TM * TN; ++i) {   // same product order as conv_patch_kernel
      const int grp = PROD == 0 ? i / (TM * TN) : 2, t = (i % (TM * TN)) / TN, u = i % TN;
      // DIRECT: weights are the MFMA's row operand, pixels its columns: a lane ends up with 4 consecutive channels of ONE pixel
      // per register quad and stores them itself (the same products summed over the same k order: bit-identical)
      if (DIRECT) acc[t][u] = mma16<PROD>(grp == 1 ? bl[u] : bh[u], grp == 0 ? al[t] : ah[t], acc[t][u]);
      else acc[t][u] = mma16<PROD>(grp == 0 ? al[t] : ah[t], grp == 1 ? bl[u] : bh[u], acc[t][u]);
    }
  };
  // swapb: the B tiles fetched from tap 7 on belong to the next tile (nt; n0 < 0: there is none)
  auto do_chunk = [&](auto firstc, auto parc, int cc, unsigned a_coff, int b_ncc, bool swapb, const TileAt& nt) {
    do_tap(IC<0>(), firstc, parc, cc, a_coff, b_ncc);
    do_tap(IC<1>(), firstc, parc, cc, a_coff, b_ncc);
    do_tap(IC<2>(), firstc, parc, cc, a_coff, b_ncc);
    do_tap(IC<3>(), firstc, parc, cc, a_coff, b_ncc);
    do_tap(IC<4>(), firstc, parc, cc, a_coff, b_ncc);
    do_tap(IC<5>(), firstc, parc, cc, a_coff, b_ncc);
    do_tap(IC<6>(), firstc, parc, cc, a_coff, b_ncc);
    if (swapb) set_b(nt);
    do_tap(IC<7>(), firstc, parc, cc, a_coff, b_ncc);
    do_tap(IC<8>(), firstc, parc, cc, a_coff, b_ncc);
  };

  // ---- epilogue of one tile: accumulators -> wave-private LDS (16 rows at a time) -> row-contiguous buffer stores ----
  const int flags = EPI >= 0 ? EPI : (g.flags & 0x3f);
  auto epi_math = [&](float4 v, int dpix, int ns) {
    if (flags & SRHIP_EPI_BIAS) {
      const float4 bb = *reinterpret_cast<const float4*>(bias_s + ns);
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    }
    if (flags & SRHIP_EPI_LRELU) {
      v.x = v.x > 0.f ? v.x : v.x * g.slope;
      v.y = v.y > 0.f ? v.y : v.y * g.slope;
      v.z = v.z > 0.f ? v.z : v.z * g.slope;
      v.w = v.w > 0.f ? v.w : v.w * g.slope;
    }
    if (flags & SRHIP_EPI_ACTMASK) {
      const float4 a4 = *reinterpret_cast<const float4*>(actmask + (size_t)dpix * g.ldd + ns);
      v.x = a4.x > 0.f ? v.x : v.x * g.slope;
      v.y = a4.y > 0.f ? v.y : v.y * g.slope;
      v.z = a4.z > 0.f ? v.z : v.z * g.slope;
      v.w = a4.w > 0.f ? v.w : v.w * g.slope;
    }
    if (flags & SRHIP_EPI_RESIDUAL) {
      const float4 r4 = *reinterpret_cast<const float4*>(residual + (size_t)dpix * g.ldr + ns);
      v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
    }
    return v;
  };
  auto epilogue_direct = [&](const TileAt& t) {
    if (ABL & 32) {
#pragma unroll
      for (int tt = 0; tt < TM; ++tt)
#pragma unroll
        for (int u = 0; u < TN; ++u) KEEP_IN(acc[tt][u]);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(z, rs_d, F_OOB + 16u * i, 0, 0);
      }
      zero_acc();
      return;
    }
    int ln = lane;                                      // opaque: per-tile addresses are recomputed, not kept in registers
    asm volatile("" : "+v"(ln));
    const int l31e = ln & 31, khe = ln >> 5;
#pragma unroll
    for (int tt = 0; tt < TM; ++tt) {
      const int pt = pix_tab[wm * WTM + tt * 32 + l31e];
      const int orow = pt >> 16, ocol = pt & 0xffff;
      const int oh = t.oh0 + orow, ow = t.ow0 + ocol;
      const bool okp = orow < pg.PH && oh < g.OH && ow < g.OW;
      const int dpix = okp ? (t.img * g.Hd + oh) * g.Wd + ow : 0;
      const int nb = t.n0 + wn * WTN + 4 * khe;
#pragma unroll
      for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) {                  // registers 4q .. 4q+3: channels u*32 + 8q + 4*khalf + 0..3
          const int n = nb + u * 32 + 8 * q;
          const bool ok = okp && n < g.K;
          const int ns = ok ? n : 0;
          float4 v = make_float4(acc[tt][u][4 * q], acc[tt][u][4 * q + 1], acc[tt][u][4 * q + 2], acc[tt][u][4 * q + 3]);
          v = epi_math(v, dpix, ns);
          const unsigned eoff = (ok && !(ABL & 1)) ? (unsigned)(dpix * g.ldd + n) * 4u : F_OOB + ((ABL & 1) ? 16u * (unsigned)((tt * TN + u) * 4 + q) : 0u);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_d, eoff, 0, 2);   // aux 2 = nt
        }
    }
    zero_acc();
  };
  auto epilogue = [&](const TileAt& t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every wave's fragment reads of the last tap are done:
    __builtin_amdgcn_s_barrier();                        // its patch buffer and ring slot 2 are free
    asm volatile("" ::: "memory");
    if (ABL & 32) {
#pragma unroll
      for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u) KEEP_IN(acc[t][u]);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        __builtin_amdgcn_raw_buffer_store_b128(z, rs_d, F_OOB + 16u * i, 0, 0);
      }
      zero_acc();
      return;
    }
    float* wl = reinterpret_cast<float*>(wave < 3 ? lds + PATCH_B + wave * STG_B : lds + RING0 + 2 * BSTAGE_B);   // the last chunk's buffer is 1
    int ln = lane;                                      // opaque: the per-pass addresses are recomputed per tile, not kept in registers
    asm volatile("" : "+v"(ln));
    const int l31e = ln & 31, khe = ln >> 5;
    // per tile and lane: this lane always serves the same four channels (cq) and the row (i * 64 + lane) / QPRW of a pass
    const int cq = ln & (QPRW - 1), rsub = ln / QPRW;
    const int n = t.n0 + wn * WTN + cq * 4;
    const bool nok = n < g.K;
    const int ns = nok ? n : 0;
    const unsigned tile_base = (unsigned)(((t.img * g.Hd + t.oh0) * g.Wd + t.ow0) * g.ldd + n) * 4u;
    const bool interior = t.oh0 + pg.PH <= g.OH && t.ow0 + pg.PW <= g.OW;      // block-uniform: no per-row bound checks
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (flags & SRHIP_EPI_BIAS) bb = *reinterpret_cast<const float4*>(bias_s + ns);
    // destination offsets of one pass (and the validity of each store)
    auto pass_offsets = [&](int p, unsigned (&doff)[NRD], bool (&okv)[NRD]) {
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        const int rr = wm * WTM + (p >> 1) * 32 + (p & 1) * 16 + i * (64 / QPRW) + rsub;
        const unsigned rel = rel_tab[rr];
        bool ok = nok && rel < F_OOB;
        if (!interior) {
          const int pt = pix_tab[rr];
          ok = ok && t.oh0 + (pt >> 16) < g.OH && t.ow0 + (pt & 0xffff) < g.OW;
        }
        okv[i] = ok;
        doff[i] = ok ? tile_base + rel : 0u;            // byte offset of (pixel, n) in dst (and in actmask: same geometry)
      }
    };
    // The activation mask of pass p + 1 is fetched BEFORE the stores of pass p are issued: the memory pipeline retires loads and
    // stores in order, so a load issued behind a pass's stores cannot return before they have completed -- inside the training
    // step, with three streams on the chip, that chained four store latencies per tile (dgrad 64 -> 256: 135 us in the step).
    float4 am[2][NRD];
    unsigned doffs[2][NRD];
    bool oks[2][NRD];
    float ps[4] = {0.f, 0.f, 0.f, 0.f}, pm[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};   // POOL: this lane's 4 channels over its 8 rows
    int pa[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    pass_offsets(0, doffs[0], oks[0]);
    if (flags & SRHIP_EPI_ACTMASK) {
#pragma unroll
      for (int i = 0; i < NRD; ++i) am[0][i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(actmask) + doffs[0][i]);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int tt = p >> 1, rb = (p & 1) * 8, cur = p & 1, nx = cur ^ 1;
#pragma unroll
      for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) wl[((r8 & 3) + 8 * (r8 >> 2) + 4 * khe) * WTN + u * 32 + l31e] = acc[tt][u][rb + r8];
      if (p < 3) {
        pass_offsets(p + 1, doffs[nx], oks[nx]);
        if (flags & SRHIP_EPI_ACTMASK) {
#pragma unroll
          for (int i = 0; i < NRD; ++i) am[nx][i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(actmask) + doffs[nx][i]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        const int row = i * (64 / QPRW) + rsub;
        float4 v = *reinterpret_cast<const float4*>(wl + row * WTN + cq * 4);
        const unsigned doff = doffs[cur][i];
        const bool ok = oks[cur][i];
        if (flags & SRHIP_EPI_BIAS) {
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (flags & SRHIP_EPI_LRELU) {
          v.x = v.x > 0.f ? v.x : v.x * g.slope;
          v.y = v.y > 0.f ? v.y : v.y * g.slope;
          v.z = v.z > 0.f ? v.z : v.z * g.slope;
          v.w = v.w > 0.f ? v.w : v.w * g.slope;
        }
        if (flags & SRHIP_EPI_ACTMASK) {
          const float4 a4 = am[cur][i];
          v.x = a4.x > 0.f ? v.x : v.x * g.slope;
          v.y = a4.y > 0.f ? v.y : v.y * g.slope;
          v.z = a4.z > 0.f ? v.z : v.z * g.slope;
          v.w = a4.w > 0.f ? v.w : v.w * g.slope;
        }
        if (flags & SRHIP_EPI_RESIDUAL) {
          const unsigned dpix = (doff >> 2) / (unsigned)g.ldd;      // (rare epilogue: the residual has its own row stride)
          const float4 r4 = *reinterpret_cast<const float4*>(residual + (size_t)dpix * g.ldr + ns);
          v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
          if (g.res2 != nullptr) {                       // the block input's other gradients (FastGeom::res2 / res3), in this order
            const float4 e4 = *reinterpret_cast<const float4*>(g.res2 + (size_t)dpix * g.ldr + ns);
            v.x += e4.x; v.y += e4.y; v.z += e4.z; v.w += e4.w;
          }
          if (g.res3 != nullptr) {
            const float4 e4 = *reinterpret_cast<const float4*>(g.res3 + (size_t)dpix * g.ldr + ns);
            v.x += e4.x; v.y += e4.y; v.z += e4.z; v.w += e4.w;
          }
        }
        const unsigned eoff = (ok && !(ABL & 1)) ? doff : F_OOB + ((ABL & 1) ? 16u * (unsigned)(p * NRD + i) : 0u);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_d, eoff, 0, 2);   // aux 2 = nt
        if (POOL && ok) {                               // pixel index inside the image: the arg-max the backward scatters to
          const int pt = pix_tab[wm * WTM + (p >> 1) * 32 + (p & 1) * 16 + i * (64 / QPRW) + rsub];
          const int pidx = (t.oh0 + (pt >> 16)) * g.OW + t.ow0 + (pt & 0xffff);
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            ps[c] += vv[c];
            if (pool_merge_takes(vv[c], pidx, pm[c], pa[c])) {
              pm[c] = vv[c];
              pa[c] = pidx;
            }
          }
        }
      }
      if (p < 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (POOL) {
      // the 8 lanes that serve the same channel quad (lane = rsub * 8 + cq) hold the wave's 64 rows between them: butterfly over
      // lane bits 3, 4, 5 (a + b == b + a and the (value, index) merge is symmetric: every lane ends with the same result)
#pragma unroll
      for (int step = 0; step < 3; ++step) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float os, om;
          int oa;
          if (step == 0) {
            os = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ps[c]), 0x128, 0xF, 0xF, false));   // row_ror:8
            om = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, pm[c]), 0x128, 0xF, 0xF, false));
            oa = __builtin_amdgcn_update_dpp(0, pa[c], 0x128, 0xF, 0xF, false);
          } else {
            const int m = step == 1 ? 16 : 32;
            os = __shfl_xor(ps[c], m, 64);
            om = __shfl_xor(pm[c], m, 64);
            oa = __shfl_xor(pa[c], m, 64);
          }
          ps[c] += os;
          if (pool_merge_takes(om, oa, pm[c], pa[c])) {
            pm[c] = om;
            pa[c] = oa;
          }
        }
      }
      const int tin = (t.oh0 / pg.PH) * pg.tiles_w + t.ow0 / pg.PW;            // tile inside the image
      const unsigned poff = (ln < QPRW && nok) ? (unsigned)(((t.img * tpi + tin) * 2 + wm) * 64 + n) * 4u : F_OOB;
      const float4 s4 = make_float4(ps[0], ps[1], ps[2], ps[3]), m4 = make_float4(pm[0], pm[1], pm[2], pm[3]);
      const u32x4 a4 = {(unsigned)pa[0], (unsigned)pa[1], (unsigned)pa[2], (unsigned)pa[3]};
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s4), rs_p, poff, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, m4), rs_p, poff < F_OOB ? poff + pool_sec : F_OOB + 16u, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(a4, rs_p, poff < F_OOB ? poff + 2u * pool_sec : F_OOB + 32u, 0, 0);
    }
    zero_acc();
  };

  // ---- epilogue onto padded planes: the same four 16-row passes.  A plane row holds the bytes of the fp32 row (per 8 channels: 8 hi | 8 lo
  // halves = 32 bytes), so the STORES are those of the fp32 form -- lane = one 16-byte piece, 64 / QPRW rows per instruction, every
  // instruction writes whole 256-byte (BN = 128) / 128-byte (BN = 64) runs.  Between staging and the stores each lane converts 8
  // channels of 64 / OPR rows IN PLACE inside its wave's staging buffer (two float4 in, bias / LeakyReLU / mask, split, 8 hi + 8 lo out).
  // (First form, until round 5's last day: the converting lane stored its own hi and lo pieces, i.e. every store instruction wrote
  // 16 bytes of each 32: the memory side counted 131.9 MB for 95.6 MB of output and the launch ran 79.5 us against 74.5 for an fp32
  // destination, profiles/r05_roofline_pmc_summary.txt.)
  auto epilogue_pp = [&](const TileAt& t) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    constexpr int OPR = WTN / 8;                        // 8-channel groups per staged row
    constexpr int NRP = 16 * OPR / 64;                  // (row, group) items per lane per pass
    float* wl = reinterpret_cast<float*>(wave < 3 ? lds + PATCH_B + wave * STG_B : lds + RING0 + 2 * BSTAGE_B);
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int l31e = ln & 31, khe = ln >> 5;
    // conversion mapping: lane = group oq of rows i * (64 / OPR) + rsub
    const int oq = ln & (OPR - 1), rsub = ln / OPR;
    const int n = t.n0 + wn * WTN + oq * 8;
    const bool nok = n < g.K;
    const int ns = nok ? n : 0;
    const unsigned row0 = (unsigned)((g.dst_guard + (t.img * (g.Hd + 1) + t.oh0) * (g.Wd + 1) + t.ow0) * g.K) * 4u;   // byte offset of the tile's first pixel row
    const unsigned tile_base = row0 + (unsigned)n * 4u;                                                                 // 8 channels = 32 bytes: [8 hi | 8 lo]
    // store mapping: lane = 16-byte piece cq of rows i * (64 / QPRW) + rsub2 (the fp32 form's)
    const int cq = ln & (QPRW - 1), rsub2 = ln / QPRW;
    const int n2 = t.n0 + wn * WTN + cq * 4;
    const bool nok2 = n2 < g.K;
    const unsigned tile_base2 = row0 + (unsigned)n2 * 4u;
    const bool interior = t.oh0 + pg.PH <= g.OH && t.ow0 + pg.PW <= g.OW;
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (flags & SRHIP_EPI_BIAS) {
      b0 = *reinterpret_cast<const float4*>(bias_s + ns);
      b1 = *reinterpret_cast<const float4*>(bias_s + ns + 4);
    }
    auto row_ok = [&](int rr, unsigned& rel) {
      rel = rel_tab[rr];
      bool ok = rel < F_OOB;
      if (!interior) {
        const int pt = pix_tab[rr];
        ok = ok && t.oh0 + (pt >> 16) < g.OH && t.ow0 + (pt & 0xffff) < g.OW;
      }
      return ok;
    };
    auto mask_offsets = [&](int p, unsigned (&doff)[NRP]) {           // where the conversion mapping finds its mask pieces
#pragma unroll
      for (int i = 0; i < NRP; ++i) {
        unsigned rel;
        const bool ok = row_ok(wm * WTM + (p >> 1) * 32 + (p & 1) * 16 + i * (64 / OPR) + rsub, rel) && nok;
        doff[i] = ok ? tile_base + rel : 0u;
      }
    };
    u32x4 am[2][NRP];                                   // mask = the hi halves of the producer's activation output (same geometry)
    unsigned moff[NRP];
    const unsigned soff = (unsigned)((t.id * NW + wave) * 64 + ln) * 8u;     // SIGNS: this lane's word of this tile
    unsigned long long sw = 0ull;
    if (SIGNS == 2) {
      const u32x2 w2 = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_s, soff, 0, 0));
      sw = (unsigned long long)w2.x | ((unsigned long long)w2.y << 32);
    }
    if (SIGNS != 2 && (flags & SRHIP_EPI_ACTMASK)) {
      mask_offsets(0, moff);
#pragma unroll
      for (int i = 0; i < NRP; ++i) am[0][i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(actmask) + moff[i]);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int tt = p >> 1, rb = (p & 1) * 8, cur = p & 1, nx = cur ^ 1;
#pragma unroll
      for (int u = 0; u < TN; ++u)
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) wl[((r8 & 3) + 8 * (r8 >> 2) + 4 * khe) * WTN + u * 32 + l31e] = acc[tt][u][rb + r8];
      if (SIGNS != 2 && p < 3 && (flags & SRHIP_EPI_ACTMASK)) {       // (before this pass's stores are issued: see the fp32 form)
        mask_offsets(p + 1, moff);
#pragma unroll
        for (int i = 0; i < NRP; ++i) am[nx][i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(actmask) + moff[i]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NRP; ++i) {                   // in-place conversion of this lane's 32-byte items
        float* item = wl + (i * (64 / OPR) + rsub) * WTN + oq * 8;
        const float4 v0 = *reinterpret_cast<const float4*>(item);
        const float4 v1 = *reinterpret_cast<const float4*>(item + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (flags & SRHIP_EPI_BIAS) {
          v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w;
          v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (flags & SRHIP_EPI_LRELU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * g.slope;
        }
        if (SIGNS == 1) {
          unsigned byte = 0u;
#pragma unroll
          for (int j = 0; j < 8; ++j) byte |= (v[j] > 0.f ? 1u : 0u) << j;
          sw |= (unsigned long long)byte << (8 * (p * NRP + i));
        }
        if (SIGNS == 2) {
          const unsigned byte = (unsigned)(sw >> (8 * (p * NRP + i))) & 0xffu;
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = ((byte >> j) & 1u) ? v[j] : v[j] * g.slope;
        } else if (flags & SRHIP_EPI_ACTMASK) {
          const unsigned mv[4] = {am[cur][i].x, am[cur][i].y, am[cur][i].z, am[cur][i].w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a = __uint_as_float((j & 1) ? (mv[j >> 1] & 0xffff0000u) : (mv[j >> 1] << 16));
            v[j] = a > 0.f ? v[j] : v[j] * g.slope;
          }
        }
        bf16x8_t hi, lo;
        split_bf16x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
        *reinterpret_cast<u32x4*>(item) = __builtin_bit_cast(u32x4, hi);
        *reinterpret_cast<u32x4*>(item + 4) = __builtin_bit_cast(u32x4, lo);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave, one staging buffer: the wave's own LDS writes are all it waits for)
#pragma unroll
      for (int i = 0; i < NRD; ++i) {
        const int row = i * (64 / QPRW) + rsub2;
        unsigned rel;
        const bool ok = row_ok(wm * WTM + (p >> 1) * 32 + (p & 1) * 16 + row, rel) && nok2;
        const u32x4 piece = *reinterpret_cast<const u32x4*>(wl + row * WTN + cq * 4);
        __builtin_amdgcn_raw_buffer_store_b128(piece, rs_d, ok ? tile_base2 + rel : F_OOB + 16u * (unsigned)(p * NRD + i), 0, 2);
      }
      if (p < 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (SIGNS == 1) {                                   // 512 contiguous bytes per wave
      const u32x2 w2 = {(unsigned)sw, (unsigned)(sw >> 32)};
      __builtin_amdgcn_raw_buffer_store_b64(w2, rs_s, soff, 0, 0);
    }
    zero_acc();
  };

  // ---- the tile walk ----
  TileAt cur = decode(blockIdx.x), nxt = cur;
  set_a(cur);
  set_b(cur);
#pragma unroll
  for (int k = 0; k < MAXP; ++k) issue_a(0, k, 0u);
  issue_b(0, 0, 0);
  if (XB == 2) issue_b(0, 0, 0);
  issue_b(1, 1, 0);
  if (XB == 2) issue_b(1, 1, 0);
#pragma unroll
  for (int i = 0; i < NS; ++i) {                    // NS dropped stores: the first tile's taps 0 and 1 count like every other tile's
    const u32x4 z = {0u, 0u, 0u, 0u};
    __builtin_amdgcn_raw_buffer_store_b128(z, rs_d, F_OOB + 16u * i, 0, 0);   // distinct offsets: identical stores would be merged
  }
  wait_vmcnt<XB * BPW + NS>();
  if (!SRCPP) {
#pragma unroll
    for (int k = 0; k < MAXP; ++k) convert_piece(0, k);
  }
  for (int vt = blockIdx.x; vt < ntiles; vt += (int)gridDim.x) {
    do_chunk(IC<1>(), IC<0>(), 0, (unsigned)(BK * 4), 1, false, cur);
    for (int cc = 1; cc + 2 < CC; cc += 2) {        // CC is even: odd chunks live in patch buffer 1, even ones in 0
      do_chunk(IC<0>(), IC<1>(), cc, (unsigned)((cc + 1) * BK * 4), cc + 1, false, cur);
      do_chunk(IC<0>(), IC<0>(), cc + 1, (unsigned)((cc + 2) * BK * 4), cc + 2, false, cur);
    }
    const int vn = vt + (int)gridDim.x;
    if (vn < ntiles) nxt = decode(vn);
    else nxt.n0 = -1;
    set_a(nxt);                                     // this tile's patches are all fetched: taps 0..2 fetch the next tile's first
    do_chunk(IC<0>(), IC<1>(), CC - 1, 0u, 0, true, nxt);
    if (DSTPP) epilogue_pp(cur);
    else if (DIRECT) epilogue_direct(cur);
    else epilogue(cur);
    cur = nxt;
  }
  wait_vmcnt<0>();                                  // the zero-fill DMAs of the tile that does not exist
}

thread_local PoolRequest g_pool_req;
thread_local SignRequest g_sign_req;
int g_pool_epi_any = 1; // srhip_debug_set(19, 0): the CLAM pooling epilogue only while at most two blocks share a CU (rounds 4-5).  Round 6: at any launch size --
                        // alone the epilogue's reduction costs the B = 32 conv what the stand-alone pooling pass takes (+5.8 against 6.0 us), in the generator's
                        // forward that pass is a launch of its own in a serial chain on the one stream that runs: +0.35 % of the step (profiles/r06_step_ab.txt)
int g_pers_small = 1;   // srhip_debug_set(11, v): 0 = launches with fewer tiles than block slots keep the one-tile kernels
int g_pers_grid = 0;      // srhip_debug_set(5, n)
int g_pers_abl = 0;       // srhip_debug_set(6, bits): timing-only ablations of conv_patch_pers_kernel<128, bias+lrelu>
static int g_num_cu = 0;
static int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    g_num_cu = n;
  }
  return g_num_cu;
}

// Launches the persistent patch kernel when it applies; returns -1 when the caller should take conv_patch_kernel instead.
int launch_patch_pers(const float* src, const float* wt, const float* bias, const float* residual, const float* actmask,
                      float* dst, const FastGeom& g_, const PatchGeom& pg, int nbm, int nbn, bool wide, int prod, int eflags,
                      hipStream_t st) {
  FastGeom g = g_;
  // section of the packed weight this arithmetic reads (conv_internal.h): split-bf16 -> the tiled image (its own byte count in
  // the descriptor), one bf16 product -> the row-major split section, fp16 -> the fp16 section
  const size_t total = (size_t)(g.w_bytes >> 2);
  const int ndst16 = (g.K + 15) / 16 * 16;
  const float* wsplit = wt + total * (prod == 0 ? 3 : prod == 2 ? 2 : 1);
  if (prod == 0) {
    const long tb = (long)9 * g.C * ndst16 * 4L;
    if (tb >= (1L << 31)) return -1;
    g.w_bytes = (unsigned)tb;
  }
  if (g.C % 32 != 0 || ((eflags & SRHIP_EPI_BIAS) && g.K > 512)) return -1;                      // an even number of 16-channel chunks: the patch-buffer parity is compile-time
  const long dbytes = ((long)g.N * g.Hd * g.Wd - 1) * (long)g.ldd * 4L + (long)g.K * 4L;
  if (dbytes >= (1L << 31)) return -1;
  const long ntiles = (long)nbm * nbn;
  int slots = 3 * num_cu();
  slots -= slots % 8;                               // keeps a block's XCD (blockIdx % 8) fixed over its tiles
  if (g.src_pp || g.dst_pp) {
    // padded-plane operands (round 5): always this kernel -- min(tiles, slots) blocks --, split-bf16 only.  Instantiated: what the RAB
    // uses (conv1 fprop: fp32 -> planes with bias + LeakyReLU; conv2 dgrad: fp32 -> planes with the activation mask; conv2 fprop:
    // planes -> fp32, plain / bias, with or without the pooling partials; conv1 dgrad: planes -> fp32 + residual) and run-time-flag forms.
    if (prod != 0) return -1;
    const long pdb = g.dst_pp ? 2L * g.dst_plane_bytes : dbytes;
    if (pdb >= (1L << 31)) return -1;
    const unsigned pdbu = (unsigned)pdb;
    const int pgrid = g_pers_grid > 0 ? (int)(g_pers_grid < ntiles ? g_pers_grid : ntiles) : (int)(ntiles < slots ? ntiles : slots);
#define SRHIP_PPX(BN_, EPI_, POOL_, SRC_, DST_, PO_, PS_)                                                                        \
  do {                                                                                                                           \
    if ((g.res2 || g.res3) && !(DST_)) g_res_req.served = 1;                                                                     \
    hipLaunchKernelGGL((conv_patch_pers_kernel<BN_, EPI_, 0, 0, POOL_, SRC_, DST_>), dim3(pgrid), dim3(256), 0, st, src, wsplit, bias, \
                       residual, actmask, dst, g, pg, nbm, nbn, pdbu, ndst16, PO_, PS_);                                         \
    return check_launch("conv_patch_pers_pp");                                                                                   \
  } while (0)
    if (g_sign_req.mode != 0) {                       // sign words of the LeakyReLU mask: the two 128-wide plane-writing forms of the RAB or nothing
      const int want = g_sign_req.mode == 1 ? (SRHIP_EPI_BIAS | SRHIP_EPI_LRELU) : SRHIP_EPI_ACTMASK;
      if (!wide || !g.dst_pp || g.K % 128 != 0 || eflags != want || g_sign_req.bytes < (size_t)ntiles * 2048) return -1;
      const float* words = static_cast<const float*>(g_sign_req.words);
      g_sign_req.served = 1;
#define SRHIP_PPS(EPI_, SRC_, SG_)                                                                                              \
  do {                                                                                                                         \
    hipLaunchKernelGGL((conv_patch_pers_kernel<128, EPI_, 0, 0, 0, SRC_, 1, SG_>), dim3(pgrid), dim3(256), 0, st, src, wsplit, bias, \
                       residual, words, dst, g, pg, nbm, nbn, pdbu, ndst16, nullptr, 0u);                                      \
    return check_launch("conv_patch_pers_pp_signs");                                                                           \
  } while (0)
      if (g_sign_req.mode == 1 && g.src_pp) SRHIP_PPS(3, 1, 1);
      if (g_sign_req.mode == 1) SRHIP_PPS(3, 0, 1);
      if (g.src_pp) SRHIP_PPS(32, 1, 2);
      SRHIP_PPS(32, 0, 2);
#undef SRHIP_PPS
    }
    if (g.src_pp && !g.dst_pp && !wide) {
      if (g_pool_req.out != nullptr && nbn == 1 && g.K == 64 && (eflags == 0 || eflags == SRHIP_EPI_BIAS) &&
          2 * pg.tiles_h * pg.tiles_w <= POOL_MAXSEG && g.Hd == g.OH && g.Wd == g.OW && (ntiles <= 2 * (slots / 3) || g_pool_epi_any)) {
        const int nseg = 2 * pg.tiles_h * pg.tiles_w;
        if ((size_t)g.N * nseg * 64 * 4 <= (size_t)g_pool_req.sec_bytes) {
          float* po = g_pool_req.out;
          const unsigned ps = g_pool_req.sec_bytes;
          g_pool_req.served_nseg = nseg;
          if (eflags == 0) SRHIP_PPX(64, 0, 1, 1, 0, po, ps);
          SRHIP_PPX(64, 1, 1, 1, 0, po, ps);
        }
      }
      if (eflags == 0) SRHIP_PPX(64, 0, 0, 1, 0, nullptr, 0u);
      if (eflags == SRHIP_EPI_BIAS) SRHIP_PPX(64, 1, 0, 1, 0, nullptr, 0u);
      if (eflags == SRHIP_EPI_RESIDUAL) SRHIP_PPX(64, 4, 0, 1, 0, nullptr, 0u);
      SRHIP_PPX(64, -1, 0, 1, 0, nullptr, 0u);
    }
    if (g.src_pp && !g.dst_pp) SRHIP_PPX(128, -1, 0, 1, 0, nullptr, 0u);
    if (!g.src_pp && wide) {
      if (eflags == (SRHIP_EPI_BIAS | SRHIP_EPI_LRELU)) SRHIP_PPX(128, 3, 0, 0, 1, nullptr, 0u);
      if (eflags == SRHIP_EPI_ACTMASK) SRHIP_PPX(128, 32, 0, 0, 1, nullptr, 0u);
      SRHIP_PPX(128, -1, 0, 0, 1, nullptr, 0u);
    }
    if (!g.src_pp) SRHIP_PPX(64, -1, 0, 0, 1, nullptr, 0u);
    if (wide && eflags == SRHIP_EPI_ACTMASK) SRHIP_PPX(128, 32, 0, 1, 1, nullptr, 0u);     // conv2 dgrad from the tail's du planes
    if (wide && eflags == (SRHIP_EPI_BIAS | SRHIP_EPI_LRELU)) SRHIP_PPX(128, 3, 0, 1, 1, nullptr, 0u);   // conv1 fprop from x planes
    if (wide) SRHIP_PPX(128, -1, 0, 1, 1, nullptr, 0u);
    SRHIP_PPX(64, -1, 0, 1, 1, nullptr, 0u);
#undef SRHIP_PPX
  }
  if (g_pers_grid < 0) return -1;
  // Fewer tiles than block slots (round 4, srhip_debug_set(11, 0) = the one-tile kernels of rounds 1-3 instead): this kernel still
  // wins -- bias in LDS, staged non-temporal epilogue, counted waits -- with ONE tile per block for the 64-wide tile (768 tiles at
  // B = 32: 74.5 against 77.8 us for the K-split one-tile kernel, 384 tiles at B = 16: 45.6 against 48.0), and for the 128-wide
  // tile with two blocks per CU walking <= 2 tiles each (729 tiles at B = 16: 39.2 us against 42.3 one-tile and 43.9 with 729
  // blocks: two resident blocks that pipeline beat three that start and end together).  `profiles/r04_patch_pers_small_grids.txt`
  int grid;
  if (g_pers_grid > 0) {
    grid = (int)(g_pers_grid < ntiles ? g_pers_grid : ntiles);
  } else if (ntiles > slots) {
    grid = slots;
  } else if (!g_pers_small) {
    return -1;
  } else if (!wide) {
    grid = (int)ntiles;
  } else if (ntiles > 2 * (slots / 3)) {
    grid = 2 * (slots / 3);
    grid -= grid % 8;
  } else {
    return -1;
  }
  const unsigned db = (unsigned)dbytes;
#define SRHIP_PP(BN_, EPI_, PROD_)                                                                                      \
  do {                                                                                                                  \
    if (g.res2 || g.res3) g_res_req.served = 1;                                                                         \
    hipLaunchKernelGGL((conv_patch_pers_kernel<BN_, EPI_, PROD_>), dim3(grid), dim3(256), 0, st, src, wsplit, bias,     \
                       residual, actmask, dst, g, pg, nbm, nbn, db, ndst16, nullptr, 0u);                                                    \
    return check_launch("conv_patch_pers");                                                                             \
  } while (0)
  if (prod != 0) {
    if (wide && prod == 1) SRHIP_PP(128, -1, 1);
    if (wide) SRHIP_PP(128, -1, 2);
    if (prod == 1) SRHIP_PP(64, -1, 1);
    SRHIP_PP(64, -1, 2);
  }
  // CLAM pooling partials from the epilogue (srhip_conv2d_fwd_pool): 64 destination channels in one N tile, plain or bias epilogue
  if (g_pool_req.out != nullptr && !wide && prod == 0 && nbn == 1 && g.K == 64 && (eflags == 0 || eflags == SRHIP_EPI_BIAS) &&
      2 * pg.tiles_h * pg.tiles_w <= POOL_MAXSEG && g.Hd == g.OH && g.Wd == g.OW &&
      (ntiles <= 2 * (slots / 3) || g_pers_grid > 0 || g_pool_epi_any)) {
    // (the reduction is ~600 VALU instructions per lane and tile at the exposed end of every block -- with 768 one-tile blocks (B = 32) it
    // adds 5.8 us to the conv, as much as the 24 MB pooling pass it replaces takes alone; with 384 (B = 16) 2.8 us against 4.3-5.9:
    // `tools/time_pool_epi.py`.  Rounds 4-5 therefore left larger launches to the pooling pass; in the step the pass costs more than alone.)
    const int nseg = 2 * pg.tiles_h * pg.tiles_w;
    if ((size_t)g.N * nseg * 64 * 4 <= (size_t)g_pool_req.sec_bytes) {
      float* po = g_pool_req.out;
      const unsigned ps = g_pool_req.sec_bytes;
      g_pool_req.served_nseg = nseg;
      if (eflags == 0)
        hipLaunchKernelGGL((conv_patch_pers_kernel<64, 0, 0, 0, 1>), dim3(grid), dim3(256), 0, st, src, wsplit, bias, residual, actmask, dst, g,
                           pg, nbm, nbn, db, ndst16, po, ps);
      else
        hipLaunchKernelGGL((conv_patch_pers_kernel<64, 1, 0, 0, 1>), dim3(grid), dim3(256), 0, st, src, wsplit, bias, residual, actmask, dst, g,
                           pg, nbm, nbn, db, ndst16, po, ps);
      return check_launch("conv_patch_pers_pool");
    }
  }
  if (g_pers_abl != 0 && wide && prod == 0 && eflags == (SRHIP_EPI_BIAS | SRHIP_EPI_LRELU)) {
#define SRHIP_PA(ABL_)                                                                                                  \
  if (g_pers_abl == ABL_) {                                                                                             \
    hipLaunchKernelGGL((conv_patch_pers_kernel<128, 3, 0, ABL_>), dim3(grid), dim3(256), 0, st, src, wsplit, bias,      \
                       residual, actmask, dst, g, pg, nbm, nbn, db, ndst16, nullptr, 0u);                                                    \
    return check_launch("conv_patch_pers");                                                                             \
  }
    SRHIP_PA(1) SRHIP_PA(2) SRHIP_PA(4) SRHIP_PA(12) SRHIP_PA(16) SRHIP_PA(32) SRHIP_PA(30) SRHIP_PA(26) SRHIP_PA(63) SRHIP_PA(64) SRHIP_PA(128) SRHIP_PA(256)
#undef SRHIP_PA
  }
#define SRHIP_PPE(BN_)                                                     \
  do {                                                                     \
    if (eflags == 0) SRHIP_PP(BN_, 0, 0);                                  \
    if (eflags == SRHIP_EPI_BIAS) SRHIP_PP(BN_, 1, 0);                     \
    if (eflags == (SRHIP_EPI_BIAS | SRHIP_EPI_LRELU)) SRHIP_PP(BN_, 3, 0); \
    if (eflags == SRHIP_EPI_ACTMASK) SRHIP_PP(BN_, 32, 0);                 \
    if (eflags == SRHIP_EPI_RESIDUAL) SRHIP_PP(BN_, 4, 0);                 \
    SRHIP_PP(BN_, -1, 0);                                                  \
  } while (0)
  if (wide) SRHIP_PPE(128);
  SRHIP_PPE(64);
#undef SRHIP_PPE
#undef SRHIP_PP
}

}  // namespace srhip
