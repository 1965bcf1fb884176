// Weight gradient of stride-1 pad-1 3x3 convolutions in split-bf16 as ONE FLAT GEMM over padded pixels (round 5), and the
// "padded planes" activation format it is built on.
//
// Padded planes (pp) of an NHWC tensor [N,H,W,C]: the split-bf16 form of every value -- hi = bf16(v), lo = bf16(v - hi), the split
// the bf16x3 kernels form anyway -- as [guard + N*(H+1)*(W+1) + tail] pixel rows of C * 4 bytes, a row = for every 8 channels the 8 hi
// halves followed by the 8 lo halves (32 bytes: exactly the LDS row image the patch kernels' in-place split leaves, so a 16-channel
// chunk of a pixel is one 64-byte run like in the fp32 tensor).  Every image row carries ONE zero pixel behind it, every image ONE
// zero row, `guard` zero pixels lie in front of pixel (0,0,0) and a zero tail behind the last.  With that padding the
// nine taps of a 3x3 window are FLAT shifts of the pixel index:
//     dW[co][ci][kh][kw] = sum over flat p of dy[p][co] * x[p + (kh-1)*(W+1) + (kw-1)][ci]
// (a shift that leaves the image lands on a zero pixel), so the K loop of the weight gradient needs no row / column / image
// bookkeeping at all: a chunk is 16 consecutive flat pixels, its operands are 16 (dy) and 18 (x, three kw taps) consecutive rows
// of the planes, and the position travels in the buffer instructions' scalar offset.  wgrad_rowtap_kernel (conv_fast.hip) spent
// ~190 scalar and ~160 vector instructions per 18 MFMAs on exactly that bookkeeping and on splitting / transposing fp32
// fragments (profiles/r04_sq_wait_buckets_roofline_kernels.txt: a wave spent 39 % of its life issuing them).
//
// Who writes planes: the RAB's own convs (conv_patch_pers.hip, round 5): conv1's epilogue leaves t = LeakyReLU(conv1(x)) and conv2's
// dgrad epilogue leaves dt as planes -- the same bytes as fp32, and every consumer (conv2 fprop, conv1 dgrad, both weight
// gradients) only ever multiplied the hi|lo split of them.  The 64-channel operand of each weight gradient (the block input x,
// the tail's gradient du) stays fp32 NHWC -- it is the residual stream -- and is split here, once per element, in place, by the
// wave that fetched it.
//
// Fragments: the planes are channel-contiguous (what fprop / dgrad want), the weight gradient contracts over PIXELS: the
// transpose is the LDS read itself (ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels] block, lane i gets
// channel i of the four pixels).  No VALU between LDS and the MFMA operands.
//
// Tile = wgrad_rowtap_kernel's: a block owns [BM output channels] x [one kh, CIS input channels, three kw] and one split of the
// pixel range; a wave 64 co x (3 kw x 32 ci) = 6 accumulator tiles, 18 MFMAs per chunk.  Same split-K partial layout, reduce
// kernels and XCD mapping, so launches can be grouped the same way.
//   CFG 1: BM 128, CIS  64: dy = planes (256-channel gradient), x = fp32        (RAB conv1, 64 -> 256)
//   CFG 2: BM  64, CIS 128: dy = fp32,  x = planes (256-channel activation)     (RAB conv2, 256 -> 64)
//
// LDS stage (13 KiB, 52 rows of 256 B; 4 stages, 3 blocks per CU):
//   CFG 1: rows 0-15 dy hi | 16-31 dy lo | 32-51 x fp32 (18 used)      CFG 2: rows 0-15 dy fp32 | 16-33 x hi | 34-51 x lo
// A plane row holds 128 channels; its four 64-byte segments are XOR-swizzled with (row & 3) so that the four pixel rows one
// transposing read touches fall on the four quarter-banks.  An fp32 row holds 64 channels; after the in-place split a 16-byte
// granule is [hi x4 | lo x4] (odd rows: [lo | hi]) and the row's two 128-byte halves are swapped when (row >> 1) & 1 -- the same
// argument for 8-byte reads at a 16-byte pitch.
#include "conv_dev.h"

namespace srhip {

struct FlatGeom {
  int N, H, W, Hp, Wp;             // image grid; Hp = H + 1, Wp = W + 1
  int C, K;                        // input / output channels
  int ldf;                         // row stride (elements) of the fp32 operand (x in CFG 1, dy in CFG 2)
  int guard;                       // zero pixels in front of the planes' pixel 0
  unsigned plane_bytes;            // byte distance hi -> lo plane
  unsigned pp_bytes, f32_bytes;    // extents for the buffer descriptors (4-wave form: the plane operand / the fp32 operand)
  unsigned x_plane_bytes, x_pp_bytes;   // 8-wave form: plane_bytes / pp_bytes describe dy, these x
  int nchunks, nsplit, cps;        // 16-pixel chunks over the padded range, split-K
  int Ktot;
};
struct FlatBatch {
  const void* x[4];
  const void* dy[4];
  float* partial[4];
  float* bias_partial[4];
  int nprob, bpp;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;

template <int ABL = 0>
__device__ __forceinline__ bf16x8_t tr_frag(unsigned addr) {
  if (ABL & 4) {
    bf16x8_t z;
    asm volatile("" : "=v"(z) : "v"(addr));
    return z;
  }      // 8 K values (two transposing reads, 4 pixel rows each) of this lane's channel
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(addr));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(addr + 1024));     // + 4 rows of 256 B
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// ABL: timing-only ablations (wrong results; srhip_debug_set(13, bits)): 1 no MFMAs, 2 no DMAs, 4 no fragment reads, 8 no partial stores, 16 no in-place split
template <int CFG, int ABL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void wgrad_flat_kernel(FlatGeom g, FlatBatch bt) {
  static_assert(CFG == 1 || CFG == 2, "two tile shapes");
  constexpr int BM = CFG == 1 ? 128 : 64, CIS = CFG == 1 ? 64 : 128;
  constexpr int WM = BM / 64, WN = 4 / WM;
  constexpr int TM = 2, TN = 3;
  constexpr int STAGE_B = 13 * 1024, NSTAGE = 4;
  constexpr int A_OFF = 0;                                   // byte offset of the dy image(s) inside a stage
  constexpr int B_OFF = (CFG == 1 ? 32 : 16) * 256;          // x image(s)
  constexpr int A_LO = 16 * 256, B_LO = 18 * 256;            // hi -> lo plane image (plane operands)
  __shared__ __attribute__((aligned(1024))) char lds[NSTAGE * STAGE_B];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const void* xp = bt.x[0];
  const void* dyp = bt.dy[0];
  float* partial = bt.partial[0];
  float* bias_partial = bt.bias_partial[0];
  int bid0 = blockIdx.x;
  if (bt.nprob > 1) {
    const int prob = __builtin_amdgcn_readfirstlane((int)blockIdx.x / bt.bpp);
    bid0 = (int)blockIdx.x - prob * bt.bpp;
    xp = bt.x[prob];
    dyp = bt.dy[prob];
    partial = bt.partial[prob];
    bias_partial = bt.bias_partial[prob];
  }
  const int ncs = g.C / CIS, ntn = 3 * ncs, ntm = (g.K + BM - 1) / BM;
  int tile_n, tile_m, split;                                 // XCD-aware order (as wgrad_rowtap_kernel): a split's tiles share an XCD's L2
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
  const int c_begin = split * g.cps;
  const int c_end = min(c_begin + g.cps, g.nchunks);
  const int nk = c_end - c_begin;
  const int shiftB = (kh - 1) * g.Wp - 1;                    // flat pixel of x row 0 of a chunk, relative to the chunk's first pixel

  // ---- DMA set-up.  Pieces (1 KiB = 4 stage rows) of a stage: CFG 1: 0-7 dy planes, 8-12 x fp32; CFG 2: 0-3 dy fp32, 4-12 x planes.
  // Every wave fetches two plane pieces and one fp32 piece; wave 0 also piece 12 (fp32 in CFG 1, planes in CFG 2).
  constexpr bool APL = CFG == 1;
  const void* pl_ptr = APL ? dyp : xp;                       // the plane operand / the fp32 operand
  const void* f_ptr = APL ? xp : dyp;
  const int plC = APL ? g.K : g.C, fC = APL ? g.C : g.K;     // their channel counts
  __amdgpu_buffer_rsrc_t rs_pl = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(pl_ptr), 0, g.pp_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(f_ptr), 0, g.f32_bytes, 0x00020000);
  const int lrow = lane >> 4, pg = lane & 15;
  constexpr int NPL = CFG == 1 ? 2 : 3, NF = CFG == 1 ? 2 : 1;      // piece slots per wave (the last one of the larger count: wave 0 only)
  unsigned pl_voff[NPL];
  unsigned pl_dst[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
    int piece, plane, r;
    if (CFG == 1) {
      piece = 2 * wave + k;                                  // 0..7: hi rows 0-15, lo rows 0-15
      plane = piece >> 2;
      r = 4 * (piece & 3) + lrow;
    } else {
      piece = k < 2 ? 4 + 2 * wave + k : 12;                 // 4..12: 36 rows = hi 0-17, lo 0-17
      const int rr = 4 * (piece - 4) + lrow;
      plane = rr >= 18 ? 1 : 0;
      r = rr - 18 * plane;
    }
    const int lg = (((pg >> 2) ^ (r & 3)) << 2) | (pg & 3);  // logical 8-channel granule this lane's slot holds
    const int ch = (APL ? m0 : ci_base) + lg * 8;
    const bool live = ch < plC && (k < 2 || wave == 0);
    pl_voff[k] = live ? (unsigned)(r * plC + ch) * 4u + (unsigned)plane * 16u : F_OOB;      // pixel row = plC * 4 bytes: per 8 channels [8 hi | 8 lo]
    pl_dst[k] = __builtin_amdgcn_readfirstlane(lds_base + piece * 1024);
  }
  // fp32 slots: per-lane walk of the padded grid (n, h, w) -> pixel of the UNPADDED tensor; a pad position is an out-of-range lane
  int f_n[NF], f_h[NF], f_w[NF], f_pix[NF];
  unsigned f_choff[NF], f_dst[NF];
  bool f_live[NF];
#pragma unroll
  for (int k = 0; k < NF; ++k) {
    const int piece = CFG == 1 ? (k == 0 ? 8 + wave : 12) : wave;
    const int r = 4 * (piece - (CFG == 1 ? 8 : 0)) + lrow;   // image row
    const int cq = pg ^ (((r >> 1) & 1) << 3);               // logical 4-channel granule
    const int ch = (APL ? ci_base : m0) + cq * 4;
    f_live[k] = ch < fC && r < (CFG == 1 ? 18 : 16) && (k == 0 || wave == 0);
    f_choff[k] = (unsigned)ch * 4u;
    f_dst[k] = __builtin_amdgcn_readfirstlane(lds_base + piece * 1024);
    // flat padded pixel of this lane's row in the block's first chunk (one image is added so that the division sees q >= 0)
    const int per = g.Hp * g.Wp;
    const long q = (long)c_begin * 16 + (CFG == 1 ? shiftB : 0) + r + per;
    const int n1 = (int)(q / per);
    const int rem = (int)(q - (long)n1 * per);
    f_n[k] = n1 - 1;
    f_h[k] = rem / g.Wp;
    f_w[k] = rem - f_h[k] * g.Wp;
    f_pix[k] = (f_n[k] * g.H + f_h[k]) * g.W + f_w[k];
  }
  auto dma_s = [&](unsigned voff, unsigned soff, __amdgpu_buffer_rsrc_t r, unsigned dst) {
    if (ABL & 2) voff = F_OOB;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(dst) : "memory");
  };
  int c_issue = c_begin;
  auto issue = [&](int stage) {                              // the DMAs of chunk c_issue into `stage`
    const unsigned so = stage * STAGE_B;
    const unsigned soff = (unsigned)(g.guard + c_issue * 16 + (APL ? 0 : shiftB)) * (unsigned)plC * 4u;
#pragma unroll
    for (int k = 0; k < NPL; ++k)
      if (k < 2 || wave == 0) dma_s(pl_voff[k], soff, rs_pl, pl_dst[k] + so);
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      if (k == 0 || wave == 0) {
        const bool ok = f_live[k] && (unsigned)f_n[k] < (unsigned)g.N && f_h[k] < g.H && f_w[k] < g.W;
        dma_s(ok ? (unsigned)f_pix[k] * (unsigned)g.ldf * 4u + f_choff[k] : F_OOB, 0u, rs_f, f_dst[k] + so);
        f_w[k] += 16;
        f_pix[k] += 16;
        while (f_w[k] >= g.Wp) {
          f_w[k] -= g.Wp;
          f_pix[k] -= g.Wp;
          if (++f_h[k] == g.Hp) {
            f_h[k] = 0;
            ++f_n[k];
          } else {
            f_pix[k] += g.W;
          }
        }
      }
    }
    ++c_issue;
  };
  const int npw = 3 + (wave == 0 ? 1 : 0);                   // DMAs of this wave per chunk
  auto wait_newer = [&](int newer) {                         // all of this wave's DMAs but those of the `newer` newest chunks have landed
    if (newer >= 2) {
      if (npw == 4) wait_vmcnt<8>();
      else wait_vmcnt<6>();
    } else if (newer == 1) {
      if (npw == 4) wait_vmcnt<4>();
      else wait_vmcnt<3>();
    } else {
      wait_vmcnt<0>();
    }
  };

  // ---- in-place split of the fp32 pieces this wave fetched (+ the bias column sums, from the raw dy rows in CFG 2)
  const bool want_bias = bias_partial != nullptr && tile_n == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto convert = [&](int stage) {
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      if (k == 0 || wave == 0) {
        const int piece = CFG == 1 ? (k == 0 ? 8 + wave : 12) : wave;
        u32x4* slot = reinterpret_cast<u32x4*>(lds + stage * STAGE_B + piece * 1024 + lane * 16);
        const float4 v = __builtin_bit_cast(float4, *slot);
        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
        const bf16x2_t h01 = {(__bf16)v.x, (__bf16)v.y}, h23 = {(__bf16)v.z, (__bf16)v.w};
        const unsigned uh01 = __builtin_bit_cast(unsigned, h01), uh23 = __builtin_bit_cast(unsigned, h23);
        const bf16x2_t l01 = {(__bf16)(v.x - __uint_as_float(uh01 << 16)), (__bf16)(v.y - __uint_as_float(uh01 & 0xffff0000u))};
        const bf16x2_t l23 = {(__bf16)(v.z - __uint_as_float(uh23 << 16)), (__bf16)(v.w - __uint_as_float(uh23 & 0xffff0000u))};
        const unsigned ul01 = __builtin_bit_cast(unsigned, l01), ul23 = __builtin_bit_cast(unsigned, l23);
        const bool odd = lrow & 1;                           // image row parity (pieces start at multiples of 4 rows)
        u32x4 o;
        o.x = odd ? ul01 : uh01;
        o.y = odd ? ul23 : uh23;
        o.z = odd ? uh01 : ul01;
        o.w = odd ? uh23 : ul23;
        *slot = o;
        if (CFG == 2 && want_bias) {                         // dy is the fp32 operand: sum the raw values
          bsum[0] += v.x; bsum[1] += v.y; bsum[2] += v.z; bsum[3] += v.w;
        }
      }
    }
    if (CFG == 1 && want_bias) {                             // dy planes: this wave's two pieces are 8 rows of ONE plane (waves 0, 1: hi; 2, 3: lo)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const u32x4 p = *reinterpret_cast<const u32x4*>(lds + stage * STAGE_B + (2 * wave + k) * 1024 + lane * 16);
        const unsigned pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bsum[2 * j] += __uint_as_float(pv[j] << 16);
          bsum[2 * j + 1] += __uint_as_float(pv[j] & 0xffff0000u);
        }
      }
    }
  };

  // ---- fragment addresses (bytes from the stage base).  Lane: i = lane & 15 -> (pixel row i >> 2, channel quad i & 3) of its 16-lane
  // group's [4][16] block; group bit 0 = channels +16, group bit 1 = pixels +8 (the MFMA's k half)
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 15, g1 = (lane >> 4) & 1, khalf = lane >> 5, l31 = lane & 31;
  const int rq = li >> 2, q4 = li & 3;
  unsigned a_hi[TM], b_hi[TN];                               // lo fragments: + A_LO / B_LO (planes), ^ 8 (split fp32 granules)
  {
    const int r = 8 * khalf + rq;                            // dy image row (second read: + 4)
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      if (CFG == 1) {                                        // planes: 128 channels per row, segment = 32 channels
        const int lseg = wm * 2 + t;
        a_hi[t] = A_OFF + r * 256 + ((lseg ^ (r & 3)) << 6) + 32 * g1 + 8 * q4;
      } else {                                               // split fp32: 64 channels per row, granule = 4 channels
        const int cq = t * 8 + 4 * g1 + q4;
        const int pcq = cq ^ (((r >> 1) & 1) << 3);
        a_hi[t] = A_OFF + r * 256 + pcq * 16 + 8 * (r & 1);
      }
    }
#pragma unroll
    for (int u = 0; u < TN; ++u) {                           // u = kw: x image row = k + kw
      const int s = r + u;
      if (CFG == 2) {
        b_hi[u] = B_OFF + s * 256 + ((wn ^ (s & 3)) << 6) + 32 * g1 + 8 * q4;
      } else {
        const int cq = wn * 8 + 4 * g1 + q4;
        const int pcq = cq ^ (((s >> 1) & 1) << 3);
        b_hi[u] = B_OFF + s * 256 + pcq * 16 + 8 * (s & 1);
      }
    }
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

  // ---- main loop: chunk kc is multiplied while chunk kc + 1 is split in place and chunks kc + 2, kc + 3 are in flight.  One barrier
  // per chunk: it publishes chunk kc + 1 (landed and split by its owners) and releases chunk kc's stage to the DMA of chunk kc + 4.
  if (nk > 0) {
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    wait_newer(min(nk, 3) - 1);
    convert(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    for (int kc = 0; kc < nk; ++kc) {
      const int st = kc & 3;
      if (kc + 3 < nk) issue((kc + 3) & 3);
      const unsigned sb = lds_base + st * STAGE_B;
      bf16x8_t ah[TM], al[TM];
#pragma unroll
      for (int t = 0; t < TM; ++t) {
        ah[t] = tr_frag<ABL>(sb + a_hi[t]);
        al[t] = tr_frag<ABL>(sb + (CFG == 1 ? a_hi[t] + A_LO : a_hi[t] ^ 8u));
      }
      // tap by tap (kw = u): two x fragments live at a time; the three products of an accumulator tile keep the order al*bh, ah*bl, ah*bh
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        const bf16x8_t bh = tr_frag<ABL>(sb + b_hi[u]);
        const bf16x8_t bl = tr_frag<ABL>(sb + (CFG == 2 ? b_hi[u] + B_LO : b_hi[u] ^ 8u));
        if (u == 1 && kc + 1 < nk) {                         // the next chunk's fp32 pieces: landed by now, split between the taps
          wait_newer(min(nk - (kc + 2), 2));
          if (!(ABL & 16)) convert((kc + 1) & 3);
        }
        if (ABL & 1) {
          asm volatile("" ::"v"(bh), "v"(bl), "v"(ah[0]), "v"(al[0]), "v"(ah[1]), "v"(al[1]));
        } else {
#pragma unroll
          for (int t = 0; t < TM; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[t], bh, acc[t][u], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < TM; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bl, acc[t][u], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < TM; ++t) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[t], bh, acc[t][u], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  // ---- partial tile -> LDS (wave-private 32x32 region) -> row-contiguous 16-byte stores; sub-tile u is tap (kh, kw = u)
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
        const int n = (kh * 3 + u) * g.C + ci_base + wn * 32 + cq * 4;
        if (m < g.K && !(ABL & 8)) *reinterpret_cast<float4*>(out + (size_t)m * g.Ktot + n) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (want_bias) {                                           // per-lane sums -> LDS -> one thread per channel adds its 16 entries in a fixed order
    __syncthreads();
    float* bl_ = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < 8; ++j) bl_[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < BM && m0 + tid < g.K) {
      float s = 0.f;
      if (CFG == 1) {                                        // entry (wave w, row lane rl): granule lg of a row with (row & 3) == rl
        const int lgq = tid >> 3, e = tid & 7;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int rl = 0; rl < 4; ++rl) {
            const int pgi = (((lgq >> 2) ^ rl) << 2) | (lgq & 3);
            s += bl_[(w * 64 + rl * 16 + pgi) * 8 + e];
          }
      } else {                                               // rows 4 w + rl of the fp32 image
        const int cq = tid >> 2, e = tid & 3;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int rl = 0; rl < 4; ++rl) {
            const int pcq = cq ^ ((((4 * w + rl) >> 1) & 1) << 3);
            s += bl_[(w * 64 + rl * 16 + pcq) * 8 + e];
          }
      }
      bias_partial[(size_t)split * g.K + m0 + tid] = s;
    }
  }
}

// ================================================================================================================================ //
// wgrad_flat8_kernel (round 5, second form): BOTH operands as padded planes, 8 waves, one block per CU, two wave groups running the
// chunk loop ONE BARRIER APART -- while group 0 multiplies chunk c (18 MFMAs per wave, matrix pipe only), group 1 issues its DMAs and
// reads its fragments of chunk c (memory / LDS pipes only), then they swap (the guide's 256^2 8-phase GEMM template, section 5).
// The 4-wave form above has every wave alternate load and multiply phases itself, three lock-stepped blocks per CU: its ablations
// (tools/ablate_wgrad_flat.py) show the phases ADDING UP -- MFMAs 72 us, DMA issue 47, fragment reads 40, in-place split 28, loop
// skeleton 36 of 182 -- because co-resident waves run the same phase at the same time.  Here the two waves of a SIMD are in opposite
// phases by construction.
//   CFG 1: block = 256 co x (kh, 3 kw, 64 ci): group g owns co [128 g, 128 g + 128); dy = planes [., K], x = planes [., C] (64-ch slice)
//   CFG 2: block =  64 co x (kh, 3 kw, 256 ci): group g owns ci [128 g, 128 g + 128) of the slice; dy = planes [., 64], x = planes [., C]
// A wave's tile, fragments and accumulators are those of the 4-wave form.  Stage (4 of them, DMA three chunks ahead):
//   CFG 1: [group 0: dy hi 16 x 256 B | dy lo] [group 1: same] [x hi 18 x 128 B | x lo | pad]            = 16 + 5 KiB
//   CFG 2: [dy hi 16 x 128 B | dy lo] [group 0: x hi 18 x 256 B | x lo] [group 1: same]                  = 4 + 18 KiB
// 128-byte rows (64 channels) swizzle their two 64-byte segments with (row >> 1) & 1: the four rows of a transposing read then fall
// on the four quarters of two bank lines.  Every wave issues exactly three 1-KiB DMAs per chunk (spare slots fetch nothing into a
// scratch KiB), so one counted vmcnt serves all waves.
// Bias gradient: the block whose tile_n == chunk % ntn sums that chunk's dy rows (each wave the pieces it fetched): the ntn blocks
// that share a split see the same chunks, so every chunk is summed once and the work is spread evenly; bias_partial has
// nsplit * ntn rows.
// F32: the 64-channel operand (x in CFG 1, dy in CFG 2) is an fp32 NHWC tensor (row stride g.ldf), fetched as it lies and split IN PLACE by
// the wave that fetched it, one chunk ahead of its first reader, inside the load phase (the 4-wave form's image: 256-byte rows,
// 16-byte granule = [4 hi | 4 lo], odd rows [lo | hi], 128-byte halves swapped when (row >> 1) & 1).  Saves the stand-alone pp_from_f32
// pass (24 MB read + 24 MB written per operand) the step would otherwise run for the block input x and the tail's gradient du.
template <int CFG, int ABL = 0, int F32 = 0>
__global__ __launch_bounds__(512) void wgrad_flat8_kernel(FlatGeom g, FlatBatch bt) {
  static_assert(CFG == 1 || CFG == 2, "two tile shapes");
  constexpr int TM = 2, TN = 3;
  constexpr int A_B = CFG == 1 ? 16 * 1024 : 4 * 1024;       // dy region of a stage
  constexpr int STAGE_B = CFG == 1 ? 21 * 1024 : 22 * 1024, NSTAGE = 4;
  constexpr int SCRATCH = NSTAGE * STAGE_B;                  // 1 KiB nobody reads
  __shared__ __attribute__((aligned(1024))) char lds[NSTAGE * STAGE_B + 1024];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  const void* xp = bt.x[0];
  const void* dyp = bt.dy[0];
  float* partial = bt.partial[0];
  float* bias_partial = bt.bias_partial[0];
  int bid0 = blockIdx.x;
  if (bt.nprob > 1) {
    const int prob = __builtin_amdgcn_readfirstlane((int)blockIdx.x / bt.bpp);
    bid0 = (int)blockIdx.x - prob * bt.bpp;
    xp = bt.x[prob];
    dyp = bt.dy[prob];
    partial = bt.partial[prob];
    bias_partial = bt.bias_partial[prob];
  }
  constexpr int BMB = CFG == 1 ? 256 : 64, CISB = CFG == 1 ? 64 : 256;
  const int ncs = g.C / CISB, ntn = 3 * ncs, ntm = (g.K + BMB - 1) / BMB;
  int tile_n, tile_m, split;                                 // the tiles of a split share an XCD (bid & 7) while whole groups of 8 splits last
  {
    const int tps = ntm * ntn, nfull = g.nsplit & ~7;
    int bid = bid0;
    if (bid < nfull * tps) {
      const int j = bid >> 3;
      split = (j / tps) * 8 + (bid & 7);
      bid = j % tps;
    } else {
      bid -= nfull * tps;
      split = nfull + bid / tps;
      bid %= tps;
    }
    tile_n = bid % ntn;
    tile_m = bid / ntn;
  }
  const int kh = tile_n / ncs, cs = tile_n - kh * ncs;
  const int m0 = tile_m * BMB, ci_base = cs * CISB;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  const int c_begin = split * g.cps;
  const int c_end = min(c_begin + g.cps, g.nchunks);
  const int nk = max(c_end - c_begin, 0);
  const int shiftB = (kh - 1) * g.Wp - 1;

  __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(dyp), 0, (F32 && CFG == 2) ? g.f32_bytes : g.pp_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(xp), 0, (F32 && CFG == 1) ? g.f32_bytes : g.x_pp_bytes, 0x00020000);
  const unsigned y_plane = g.plane_bytes, x_plane = g.x_plane_bytes;      // byte distance hi -> lo
  // ---- the three DMA slots of this wave: (descriptor is x?, per-lane byte offset inside the operand, LDS destination inside a stage)
  unsigned d_voff[3], d_dst[3];
  bool d_isx[3];
  {
    // wide image (256-byte rows, 128 channels): piece = 4 rows x 16 granules; narrow (128-byte rows, 64 channels): 8 rows x 8 granules
    auto wide = [&](int piece_in_plane_pair, int nrows, int chan0, int chanC, unsigned, unsigned& voff) {
      const int rr = 4 * piece_in_plane_pair + (lane >> 4);                    // row of the [hi rows | lo rows] image
      const int plane = rr >= nrows ? 1 : 0, r = rr - nrows * plane;
      const int pgl = lane & 15;
      const int lg = (((pgl >> 2) ^ (r & 3)) << 2) | (pgl & 3);
      const int ch = chan0 + lg * 8;
      voff = (rr < 2 * nrows && ch < chanC) ? (unsigned)(r * chanC + ch) * 4u + (unsigned)plane * 16u : F_OOB;
    };
    auto narrow = [&](int piece_in_plane_pair, int nrows, int chan0, int chanC, unsigned, unsigned& voff) {
      const int rr = 8 * piece_in_plane_pair + (lane >> 3);
      const int plane = rr >= nrows ? 1 : 0, r = rr - nrows * plane;
      const int pgl = lane & 7;
      const int lg = (((pgl >> 2) ^ ((r >> 1) & 1)) << 2) | (pgl & 3);
      const int ch = chan0 + lg * 8;
      voff = (rr < 2 * nrows && ch < chanC) ? (unsigned)(r * chanC + ch) * 4u + (unsigned)plane * 16u : F_OOB;
    };
    if (CFG == 1) {
      // slots 0, 1: this group's dy image (8 pieces: hi 0-3, lo 4-7): pieces w4 and 4 + w4;  slot 2: x image piece `wave` (5 pieces), else scratch
      wide(w4, 16, m0 + 128 * grp, g.K, y_plane, d_voff[0]);
      wide(4 + w4, 16, m0 + 128 * grp, g.K, y_plane, d_voff[1]);
      d_dst[0] = grp * 8192 + w4 * 1024;
      d_dst[1] = grp * 8192 + (4 + w4) * 1024;
      d_isx[0] = d_isx[1] = false;
      narrow(wave < 5 ? wave : 0, 18, ci_base, g.C, x_plane, d_voff[2]);
      if (wave >= 5) d_voff[2] = F_OOB;
      d_dst[2] = wave < 5 ? A_B + wave * 1024 : SCRATCH;
      d_isx[2] = true;
    } else {
      // slots 0, 1: this group's x image (9 pieces: 36 rows of 256 B): pieces 2 w4, 2 w4 + 1;  slot 2: waves 0-3: dy image piece `wave` (4 pieces:
      // hi rows 0-7, 8-15, lo 0-7, 8-15), waves 4, 5: the ninth x piece of group 0 / 1, waves 6, 7: scratch
      wide(2 * w4, 18, ci_base + 128 * grp, g.C, x_plane, d_voff[0]);
      wide(2 * w4 + 1, 18, ci_base + 128 * grp, g.C, x_plane, d_voff[1]);
      d_dst[0] = A_B + grp * 9216 + (2 * w4) * 1024;
      d_dst[1] = A_B + grp * 9216 + (2 * w4 + 1) * 1024;
      d_isx[0] = d_isx[1] = true;
      if (wave < 4) {
        narrow(wave, 16, m0, g.K, y_plane, d_voff[2]);
        d_dst[2] = wave * 1024;
        d_isx[2] = false;
      } else if (wave < 6) {
        wide(8, 18, ci_base + 128 * (wave - 4), g.C, x_plane, d_voff[2]);
        d_dst[2] = A_B + (wave - 4) * 9216 + 8 * 1024;
        d_isx[2] = true;
      } else {
        d_voff[2] = F_OOB;
        d_dst[2] = SCRATCH;
        d_isx[2] = true;
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) d_dst[k] = __builtin_amdgcn_readfirstlane(lds_base + d_dst[k]);
  }
  // F32: slot 2 of waves 0-4 (CFG 1: x rows 4 w .. 4 w + 3 of 18) / 0-3 (CFG 2: dy rows 4 w .. of 16) is a piece of the fp32 operand:
  // a per-lane walk of the padded grid (n, h, w) -> pixel of the UNPADDED tensor; a pad position is an out-of-range lane
  const bool f_wave = F32 && wave < (CFG == 1 ? 5 : 4);
  int f_n = 0, f_h = 0, f_w = 0, f_pix = 0;
  unsigned f_choff = 0;
  bool f_live = false;
  if (F32) {
    const int r = 4 * wave + (lane >> 4);                    // image row
    const int pcq = lane & 15;
    const int cq = pcq ^ (((r >> 1) & 1) << 3);              // logical 4-channel granule
    const int ch = (CFG == 1 ? ci_base : m0) + cq * 4;
    const int fC = CFG == 1 ? g.C : g.K;
    f_live = f_wave && ch < fC && r < (CFG == 1 ? 18 : 16);
    f_choff = (unsigned)ch * 4u;
    const int per = g.Hp * g.Wp;
    const long q = (long)c_begin * 16 + (CFG == 1 ? shiftB : 0) + r + per;   // (one image is added so that the division sees q >= 0)
    const int n1 = (int)(q / per);
    const int rem = (int)(q - (long)n1 * per);
    f_n = n1 - 1;
    f_h = rem / g.Wp;
    f_w = rem - f_h * g.Wp;
    f_pix = (f_n * g.H + f_h) * g.W + f_w;
    if (f_wave) d_dst[2] = __builtin_amdgcn_readfirstlane(lds_base + (CFG == 1 ? A_B : 0) + wave * 1024);
  }
  auto dma_s = [&](unsigned voff, unsigned soff, __amdgpu_buffer_rsrc_t r, unsigned dst) {
    if (ABL & 2) voff = F_OOB;
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(r), "s"(soff), "s"(dst) : "memory");
  };
  const unsigned yC2 = (unsigned)g.K * 4u, xC2 = (unsigned)g.C * 4u;     // bytes per pixel row of the padded tensors
  auto issue = [&](int c, int stage) {                       // chunk c (relative to c_begin; >= nk: nothing is fetched, the count stays)
    const bool live = c < nk;
    const unsigned so_y = (unsigned)(g.guard + (c_begin + c) * 16) * yC2;
    const unsigned so_x = (unsigned)(g.guard + (c_begin + c) * 16 + shiftB) * xC2;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const bool isx = d_isx[k];                             // wave-uniform
      const unsigned dst = d_dst[k] == lds_base + SCRATCH ? d_dst[k] : d_dst[k] + stage * STAGE_B;
      if (F32 && k == 2 && f_wave) {                         // the fp32 operand's piece: per-lane pixel, no scalar offset
        const bool ok = live && f_live && (unsigned)f_n < (unsigned)g.N && f_h < g.H && f_w < g.W;
        const unsigned vo = ok ? (unsigned)f_pix * (unsigned)g.ldf * 4u + f_choff : F_OOB;
        if (CFG == 1) dma_s(vo, 0u, rs_x, dst);
        else dma_s(vo, 0u, rs_y, dst);
        f_w += 16;
        f_pix += 16;
        while (f_w >= g.Wp) {
          f_w -= g.Wp;
          f_pix -= g.Wp;
          if (++f_h == g.Hp) {
            f_h = 0;
            ++f_n;
          } else {
            f_pix += g.W;
          }
        }
        continue;
      }
      if (isx) dma_s(live ? d_voff[k] : F_OOB, live ? so_x : 0u, rs_x, dst);
      else dma_s(live ? d_voff[k] : F_OOB, live ? so_y : 0u, rs_y, dst);
    }
  };

  // ---- fragment addresses (bytes from the stage base)
  constexpr int WN = CFG == 1 ? 2 : 4;
  const int wm = w4 / WN, wn = w4 - wm * WN;
  const int li = lane & 15, g1 = (lane >> 4) & 1, khalf = lane >> 5, l31 = lane & 31;
  const int rq = li >> 2, q4 = li & 3;
  unsigned a_hi[TM], b_hi[TN];
  constexpr int A_LO = CFG == 1 ? 16 * 256 : 16 * 128, B_LO = CFG == 1 ? 18 * 128 : 18 * 256;          // hi -> lo image of a plane operand
  constexpr int A_R4 = (CFG == 1 || F32) ? 1024 : 512, B_R4 = (CFG == 1 && !F32) ? 512 : 1024;     // + 4 pixel rows
  constexpr bool A_F32 = F32 && CFG == 2, B_F32 = F32 && CFG == 1;
  {
    const int r = 8 * khalf + rq;
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      if (CFG == 1) a_hi[t] = grp * 8192 + r * 256 + (((wm * 2 + t) ^ (r & 3)) << 6) + 32 * g1 + 8 * q4;
      else if (F32) a_hi[t] = r * 256 + (((t * 8 + 4 * g1 + q4) ^ (((r >> 1) & 1) << 3)) << 4) + 8 * (r & 1);
      else a_hi[t] = r * 128 + ((t ^ ((r >> 1) & 1)) << 6) + 32 * g1 + 8 * q4;
    }
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      const int s = r + u;
      if (CFG == 1 && F32) b_hi[u] = A_B + s * 256 + (((wn * 8 + 4 * g1 + q4) ^ (((s >> 1) & 1) << 3)) << 4) + 8 * (s & 1);
      else if (CFG == 1) b_hi[u] = A_B + s * 128 + ((wn ^ ((s >> 1) & 1)) << 6) + 32 * g1 + 8 * q4;
      else b_hi[u] = A_B + grp * 9216 + s * 256 + ((wn ^ (s & 3)) << 6) + 32 * g1 + 8 * q4;
    }
  }
  auto frag = [&](unsigned addr, int r4) {                   // two transposing reads, 4 pixel rows apart
    bf16x8_t z;
    if (ABL & 4) {
      asm volatile("" : "=v"(z) : "v"(addr));
      return z;
    }
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(addr));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(addr + r4));
    return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int u = 0; u < TN; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
  const bool want_bias = bias_partial != nullptr;
  const bool bias_wave = CFG == 1 || wave < 4;               // waves that fetch dy pieces
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // F32: in-place split of this wave's fp32 piece of the stage (+ the bias column sums from the raw dy values, CFG 2)
  auto convert_load = [&](int stage) {                       // this wave's raw fp32 granule (waves without an fp32 piece: nothing)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f_wave) v = __builtin_bit_cast(float4, *reinterpret_cast<const u32x4*>(lds + stage * STAGE_B + (CFG == 1 ? A_B : 0) + wave * 1024 + lane * 16));
    return v;
  };
  auto convert_store = [&](int stage, const float4& v, bool sum_bias) {
    if (!f_wave) return;
    u32x4* slot = reinterpret_cast<u32x4*>(lds + stage * STAGE_B + (CFG == 1 ? A_B : 0) + wave * 1024 + lane * 16);
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t h01 = {(__bf16)v.x, (__bf16)v.y}, h23 = {(__bf16)v.z, (__bf16)v.w};
    const unsigned uh01 = __builtin_bit_cast(unsigned, h01), uh23 = __builtin_bit_cast(unsigned, h23);
    const bf16x2_t l01 = {(__bf16)(v.x - __uint_as_float(uh01 << 16)), (__bf16)(v.y - __uint_as_float(uh01 & 0xffff0000u))};
    const bf16x2_t l23 = {(__bf16)(v.z - __uint_as_float(uh23 << 16)), (__bf16)(v.w - __uint_as_float(uh23 & 0xffff0000u))};
    const unsigned ul01 = __builtin_bit_cast(unsigned, l01), ul23 = __builtin_bit_cast(unsigned, l23);
    const bool odd = (lane >> 4) & 1;                        // image row parity (pieces start at multiples of 4 rows)
    u32x4 o;
    o.x = odd ? ul01 : uh01;
    o.y = odd ? ul23 : uh23;
    o.z = odd ? uh01 : ul01;
    o.w = odd ? uh23 : ul23;
    *slot = o;
    if (CFG == 2 && sum_bias) {
      bsum[0] += v.x; bsum[1] += v.y; bsum[2] += v.z; bsum[3] += v.w;
    }
  };
  auto convert = [&](int stage, bool sum_bias) { convert_store(stage, convert_load(stage), sum_bias); };
  auto bias_chunk = [&](int kc) { return want_bias && (c_begin + kc) % ntn == tile_n; };

  // ---- prologue: three chunks in flight, chunk 0 landed for everybody, group 1 one barrier behind
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  wait_vmcnt<6>();
  if (F32) {
    convert(0, bias_chunk(0));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (grp == 1) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  for (int kc = 0; kc < nk; ++kc) {
    // ---- load phase of chunk kc (the other group multiplies)
    float4 raw = make_float4(0.f, 0.f, 0.f, 0.f);
    if (F32) {                                              // chunk kc + 1's own pieces have landed (kc + 2 may be in flight): fetch the raw fp32
      wait_vmcnt<3>();                                      // granule FIRST, so that its LDS latency hides under the DMA issue and the
      if (kc + 1 < nk) raw = convert_load((kc + 1) & 3);    // fragment reads below; it is split and written back at the end of the phase
    }
    issue(kc + 3, (kc + 3) & 3);
    const unsigned sb = lds_base + (kc & 3) * STAGE_B;
    bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      ah[t] = frag(sb + a_hi[t], A_R4);
      al[t] = frag(sb + (A_F32 ? a_hi[t] ^ 8u : a_hi[t] + A_LO), A_R4);
    }
#pragma unroll
    for (int u = 0; u < TN; ++u) {
      bh[u] = frag(sb + b_hi[u], B_R4);
      bl[u] = frag(sb + (B_F32 ? b_hi[u] ^ 8u : b_hi[u] + B_LO), B_R4);
    }
    if (!A_F32 && want_bias && bias_wave && (c_begin + kc) % ntn == tile_n) {     // this block's share of the bias gradient: the dy pieces this wave fetched
#pragma unroll
      for (int k = 0; k < (CFG == 1 ? 2 : 1); ++k) {
        const unsigned off = (CFG == 1 ? grp * 8192 + (4 * k + w4) * 1024 : wave * 1024) + lane * 16;
        const u32x4 p = *reinterpret_cast<const u32x4*>(lds + (kc & 3) * STAGE_B + off);
        const unsigned pv[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bsum[2 * j] += __uint_as_float(pv[j] << 16);
          bsum[2 * j + 1] += __uint_as_float(pv[j] & 0xffff0000u);
        }
      }
    }
    if (F32) {                                              // one barrier (group 0) / two (group 1) ahead of the split granule's first reader
      if (kc + 1 < nk) convert_store((kc + 1) & 3, raw, bias_chunk(kc + 1));
    } else if (grp == 1) {
      wait_vmcnt<6>();                                      // own DMAs of chunk kc + 1 landed before the barrier that publishes it
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- multiply phase (the other group loads)
    if (!(ABL & 1)) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 3 * TM * TN; ++i) {
        const int pr = i / (TM * TN), t = (i % (TM * TN)) / TN, u = i % TN;
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pr == 0 ? al[t] : ah[t], pr == 1 ? bl[u] : bh[u], acc[t][u], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
    } else {
#pragma unroll
      for (int t = 0; t < TM; ++t) asm volatile("" ::"v"(ah[t]), "v"(al[t]));
#pragma unroll
      for (int u = 0; u < TN; ++u) asm volatile("" ::"v"(bh[u]), "v"(bl[u]));
    }
    if (!F32 && grp == 0) wait_vmcnt<6>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  if (grp == 0) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  wait_vmcnt<0>();                                           // the dead chunks' zero fills

  // ---- partial tile -> LDS (wave-private 32x32 region) -> row-contiguous 16-byte stores
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
        const int m = m0 + (CFG == 1 ? 128 * grp : 0) + wm * 64 + t * 32 + row;
        const int n = (kh * 3 + u) * g.C + ci_base + (CFG == 2 ? 128 * grp : 0) + wn * 32 + cq * 4;
        if (m < g.K && !(ABL & 8)) *reinterpret_cast<float4*>(out + (size_t)m * g.Ktot + n) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (want_bias) {                                           // lane sums -> LDS -> one thread per channel adds its entries in a fixed order
    __syncthreads();
    float* bl_ = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int j = 0; j < 8; ++j) bl_[tid * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < BMB && m0 + tid < g.K) {
      float s = 0.f;
      const int e = tid & 7;
      if (CFG == 1) {                                        // channel tid: group tid >> 7, granule lgq of rows with (row & 3) == rl
        const int gq = tid >> 7, lgq = (tid & 127) >> 3;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int rl = 0; rl < 4; ++rl) {
            const int pgi = (((lgq >> 2) ^ rl) << 2) | (lgq & 3);
            s += bl_[((gq * 4 + w) * 64 + rl * 16 + pgi) * 8 + e];
          }
      } else if (F32) {                                      // waves 0-3: rows 4 w + rl of the fp32 image, 4-channel granules
        const int cq = tid >> 2, e4 = tid & 3;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int rl = 0; rl < 4; ++rl) {
            const int pcq = cq ^ ((((4 * w + rl) >> 1) & 1) << 3);
            s += bl_[(w * 64 + rl * 16 + pcq) * 8 + e4];
          }
      } else {                                               // waves 0-3: pieces (plane, 8-row half); rows 8 (w & 1) + rl
        const int lgq = tid >> 3;
#pragma unroll
        for (int w = 0; w < 4; ++w)
#pragma unroll
          for (int rl = 0; rl < 8; ++rl) {
            const int r = 8 * (w & 1) + rl;
            const int pgi = (((lgq >> 2) ^ ((r >> 1) & 1)) << 2) | (lgq & 3);
            s += bl_[(w * 64 + rl * 8 + pgi) * 8 + e];
          }
      }
      bias_partial[((size_t)split * ntn + tile_n) * g.K + m0 + tid] = s;
    }
  }
}

// ---- fp32 NHWC <-> padded planes (tests, the bench's operand set-up, and any producer that has no planes epilogue) ----------- //
__global__ __launch_bounds__(256) void pp_from_f32_kernel(const float* __restrict__ x, __bf16* __restrict__ pp, int n, int h, int w, int c, int ldx,
                                                          int guard, long plane_elems) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;      // one 8-channel granule per thread
  const int c8 = c >> 3;
  const long total = (long)n * h * w * c8;
  if (i >= total) return;
  const int g8 = (int)(i % c8);
  const long pix = i / c8;
  const int ww = (int)(pix % w);
  const long t = pix / w;
  const int hh = (int)(t % h), nn = (int)(t / h);
  const float4 v0 = *reinterpret_cast<const float4*>(x + pix * ldx + g8 * 8);
  const float4 v1 = *reinterpret_cast<const float4*>(x + pix * ldx + g8 * 8 + 4);
  bf16x8_t hi, lo;
  split_bf16x8(v0, v1, hi, lo);
  const long row = guard + ((long)nn * (h + 1) + hh) * (w + 1) + ww;
  *reinterpret_cast<bf16x8_t*>(pp + row * 2 * c + g8 * 16) = hi;
  *reinterpret_cast<bf16x8_t*>(pp + row * 2 * c + g8 * 16 + 8) = lo;
}
__global__ __launch_bounds__(256) void pp_to_f32_kernel(const __bf16* __restrict__ pp, float* __restrict__ x, int n, int h, int w, int c, int ldx,
                                                        int guard, long plane_elems) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int c8 = c >> 3;
  const long total = (long)n * h * w * c8;
  if (i >= total) return;
  const int g8 = (int)(i % c8);
  const long pix = i / c8;
  const int ww = (int)(pix % w);
  const long t = pix / w;
  const int hh = (int)(t % h), nn = (int)(t / h);
  const long row = guard + ((long)nn * (h + 1) + hh) * (w + 1) + ww;
  const bf16x8_t hi = *reinterpret_cast<const bf16x8_t*>(pp + row * 2 * c + g8 * 16);
  const bf16x8_t lo = *reinterpret_cast<const bf16x8_t*>(pp + row * 2 * c + g8 * 16 + 8);
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (float)hi[j] + (float)lo[j];
  *reinterpret_cast<float4*>(x + pix * ldx + g8 * 8) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<float4*>(x + pix * ldx + g8 * 8 + 4) = make_float4(o[4], o[5], o[6], o[7]);
}

// ---- host side -------------------------------------------------------------------------------------------------------------- //
extern int g_conv_math;
int g_flat_abl = 0;         // srhip_debug_set(13, bits): timing-only ablations of wgrad_flat_kernel
int g_flat_f32_k8 = 1;      // srhip_debug_set(14, v): 0 = one fp32 operand always takes the 4-wave kernel (1: the 8-wave kernel where its tile fits)
int g_flat_blocks = 768;      // srhip_debug_set(12, n): split-K block target of wgrad_flat_kernel
bool launch_reduce4_shared(int nprob, const float* const* partial, const float* const* bias_partial, float* const* dw, float* const* db,
                           int nsplit, int cout, int cin, int khkw, int ktot, int accumulate, hipStream_t st, int nbias);

int pp_guard(int w) { return ((w + 2) + 15) / 16 * 16; }
long pp_plane_pixels(int n, int h, int w) {            // guard | N (H+1)(W+1) | tail: a last chunk's 16 + the largest shift + slack
  return (long)pp_guard(w) + (long)n * (h + 1) * (w + 1) + (w + 1) + 48;
}
int pp_from_f32(const float* x, void* pp, int n, int h, int w, int c, int ldx, void* stream) {
  SRHIP_REQUIRE(x && pp && c % 8 == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)pp) & 15) == 0, "pp_from_f32: C %% 8 == 0, 16-byte aligned tensors");
  const long total = (long)n * h * w * (c / 8);
  if (total <= 0) return SRHIP_OK;
  hipLaunchKernelGGL(pp_from_f32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), x, static_cast<__bf16*>(pp), n, h, w, c, ldx,
                     pp_guard(w), pp_plane_pixels(n, h, w) * c);
  return check_launch("pp_from_f32");
}
int pp_to_f32(const void* pp, float* x, int n, int h, int w, int c, int ldx, void* stream) {
  SRHIP_REQUIRE(x && pp && c % 8 == 0 && ldx % 4 == 0 && (((uintptr_t)x | (uintptr_t)pp) & 15) == 0, "pp_to_f32: C %% 8 == 0, 16-byte aligned tensors");
  const long total = (long)n * h * w * (c / 8);
  if (total <= 0) return SRHIP_OK;
  hipLaunchKernelGGL(pp_to_f32_kernel, dim3(cdiv(total, 256)), dim3(256), 0, as_stream(stream), static_cast<const __bf16*>(pp), x, n, h, w, c, ldx,
                     pp_guard(w), pp_plane_pixels(n, h, w) * c);
  return check_launch("pp_to_f32");
}

static int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  }
  return n;
}
// operand formats a 3x3 s1 p1 weight gradient of this shape is served in (bit mask): 1 = x fp32 + dy planes (4-wave kernel, CFG 1),
// 2 = x planes + dy fp32 (4-wave, CFG 2), 4 = both planes (8-wave kernel)
int flat_wgrad_ok(int n, int h, int w, int cin, int cout) {
  if (g_conv_math != 1) return 0;
  int m = 0;
  const long px = pp_plane_pixels(n, h, w);
  const bool fits = px * (cin > cout ? cin : cout) * 4L < (1L << 31);
  if (!fits) return 0;
  if (cout % 8 == 0 && cout >= 128 && cin % 64 == 0) m |= 1;
  if (cout == 64 && cin % 128 == 0) m |= 2;
  if ((cout % 8 == 0 && cout >= 256 && cin % 64 == 0) || (cout == 64 && cin % 256 == 0)) m |= 4;
  return m;
}
struct FlatPlan {
  int kernel;      // 4 or 8 waves
  int cfg, tiles, ntn, nsplit;
};
static FlatPlan flat_plan(int x_pp, int dy_pp, int cin, int cout, int nchunks, int nprob) {
  FlatPlan p;
  const bool k8shape = (cout % 8 == 0 && cout >= 256 && cin % 64 == 0) || (cout == 64 && cin % 256 == 0);
  p.kernel = ((x_pp && dy_pp) || (k8shape && g_flat_f32_k8)) ? 8 : 4;
  if (p.kernel == 8) {
    p.cfg = cout >= 256 ? 1 : 2;
    const int ncs = cin / (p.cfg == 1 ? 64 : 256);
    p.ntn = 3 * ncs;
    p.tiles = cdiv(cout, p.cfg == 1 ? 256 : 64) * p.ntn;
    int ns = (g_flat_blocks > 0 && g_flat_blocks != 768 ? g_flat_blocks : num_cus()) / (p.tiles * nprob);   // one block per CU
    if (ns < 1) ns = 1;
    while (ns > 1 && cdiv(nchunks, ns) < 8) --ns;
    p.nsplit = ns;
  } else {
    p.cfg = dy_pp ? 1 : 2;
    p.ntn = 3 * (cin / (p.cfg == 1 ? 64 : 128));
    p.tiles = cdiv(cout, p.cfg == 1 ? 128 : 64) * p.ntn;
    int ns = g_flat_blocks / (p.tiles * nprob);
    ns = ns / 8 * 8;
    if (ns < 8) ns = 8;
    while (ns > 8 && cdiv(nchunks, ns) < 8) ns -= 8;
    p.nsplit = ns;
  }
  return p;
}
static bool flat_served(int x_pp, int dy_pp, int n, int h, int w, int cin, int cout) {
  const int m = flat_wgrad_ok(n, h, w, cin, cout);
  return (x_pp && dy_pp) ? (m & 4) != 0 : dy_pp ? (m & 1) != 0 : x_pp ? (m & 2) != 0 : false;
}
size_t flat_wgrad_workspace(int nprob, int x_pp, int dy_pp, int n, int h, int w, int cin, int cout) {
  if (nprob < 1 || !flat_served(x_pp, dy_pp, n, h, w, cin, cout)) return 0;
  const int nchunks = cdiv((long)n * (h + 1) * (w + 1), 16);
  const FlatPlan p = flat_plan(x_pp, dy_pp, cin, cout, nchunks, nprob);
  return (size_t)nprob * p.nsplit * ((size_t)cout * 9 * cin + (size_t)p.ntn * cout) * sizeof(float);
}
int flat_wgrad(int nprob, const void* const* x, const void* const* dy, int x_pp, int dy_pp, float* const* dw, float* const* db, int accumulate,
               void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout, int ldf, void* stream) {
  hipStream_t st = as_stream(stream);
  SRHIP_REQUIRE(nprob >= 1 && nprob <= 4, "conv2d_wgrad_pp: 1..4 problems per launch");
  SRHIP_REQUIRE(flat_served(x_pp, dy_pp, n, h, w, cin, cout),
                "conv2d_wgrad_pp: shape / arithmetic mode / operand formats not served (srhip_conv2d_wgrad_pp_ok)");
  FlatGeom g;
  g.N = n; g.H = h; g.W = w; g.Hp = h + 1; g.Wp = w + 1;
  g.C = cin; g.K = cout; g.ldf = ldf;
  g.guard = pp_guard(w);
  const long ppx = pp_plane_pixels(n, h, w);
  g.nchunks = cdiv((long)n * g.Hp * g.Wp, 16);
  const FlatPlan p = flat_plan(x_pp, dy_pp, cin, cout, g.nchunks, nprob);
  g.x_plane_bytes = (unsigned)(ppx * cin * 2L);           // (half the tensor: [rows][per 8 channels: 8 hi | 8 lo])
  g.x_pp_bytes = 2u * g.x_plane_bytes;
  if (p.kernel == 8) {
    g.plane_bytes = (unsigned)(ppx * cout * 2L);               // dy
    g.pp_bytes = 2u * g.plane_bytes;
    g.f32_bytes = 0;
    if (!(x_pp && dy_pp)) {
      const int fC = dy_pp ? cin : cout;                        // channels of the fp32 operand
      const long fb = ((long)n * h * w - 1) * (long)ldf * 4L + (long)fC * 4L;
      SRHIP_REQUIRE(fb < (1L << 31) && ldf % 4 == 0 && ldf >= fC, "conv2d_wgrad_pp: fp32 operand >= 2 GiB or bad row stride");
      g.f32_bytes = (unsigned)fb;
    }
  } else {
    const int plC = p.cfg == 1 ? cout : cin, fC = p.cfg == 1 ? cin : cout;
    g.plane_bytes = (unsigned)(ppx * plC * 2L);
    g.pp_bytes = 2u * g.plane_bytes;
    const long fb = ((long)n * h * w - 1) * (long)ldf * 4L + (long)fC * 4L;
    SRHIP_REQUIRE(fb < (1L << 31) && ldf % 4 == 0 && ldf >= fC, "conv2d_wgrad_pp: fp32 operand >= 2 GiB or bad row stride");
    g.f32_bytes = (unsigned)fb;
  }
  g.nsplit = p.nsplit;
  g.cps = cdiv(g.nchunks, g.nsplit);
  g.Ktot = 9 * cin;
  const int nbias = p.kernel == 8 ? g.nsplit * p.ntn : g.nsplit;
  const size_t per = (size_t)g.nsplit * ((size_t)cout * g.Ktot + (size_t)p.ntn * cout);
  SRHIP_REQUIRE(workspace && workspace_bytes >= per * nprob * sizeof(float), "conv2d_wgrad_pp: workspace too small");
  FlatBatch bt;
  bt.nprob = nprob;
  bt.bpp = p.tiles * g.nsplit;
  for (int i = 0; i < 4; ++i) {
    const int k = i < nprob ? i : 0;
    SRHIP_REQUIRE(x[k] && dy[k] && dw[k] && ((((uintptr_t)x[k]) | ((uintptr_t)dy[k])) & 15) == 0, "conv2d_wgrad_pp: null / unaligned tensor");
    float* part = static_cast<float*>(workspace) + per * k;
    bt.x[i] = x[k];
    bt.dy[i] = dy[k];
    bt.partial[i] = part;
    bt.bias_partial[i] = (db && db[k]) ? part + (size_t)g.nsplit * cout * g.Ktot : nullptr;
  }
  const int blocks = bt.bpp * nprob;
  const int cfg = p.cfg;
  if (p.kernel == 8) {
#define SRHIP_F8(ABL_)                                                                                       \
  if (g_flat_abl == ABL_) {                                                                                 \
    if (cfg == 1) hipLaunchKernelGGL((wgrad_flat8_kernel<1, ABL_>), dim3(blocks), dim3(512), 0, st, g, bt);  \
    else hipLaunchKernelGGL((wgrad_flat8_kernel<2, ABL_>), dim3(blocks), dim3(512), 0, st, g, bt);           \
  } else
    SRHIP_F8(1) SRHIP_F8(2) SRHIP_F8(4) SRHIP_F8(8) SRHIP_F8(3) SRHIP_F8(5) SRHIP_F8(6) SRHIP_F8(7) SRHIP_F8(15)
#undef SRHIP_F8
    if (!(x_pp && dy_pp)) {
      if (cfg == 1) hipLaunchKernelGGL((wgrad_flat8_kernel<1, 0, 1>), dim3(blocks), dim3(512), 0, st, g, bt);
      else hipLaunchKernelGGL((wgrad_flat8_kernel<2, 0, 1>), dim3(blocks), dim3(512), 0, st, g, bt);
    } else if (cfg == 1) hipLaunchKernelGGL((wgrad_flat8_kernel<1>), dim3(blocks), dim3(512), 0, st, g, bt);
    else hipLaunchKernelGGL((wgrad_flat8_kernel<2>), dim3(blocks), dim3(512), 0, st, g, bt);
  } else {
#define SRHIP_FA(ABL_)                                                                                   \
  if (g_flat_abl == ABL_) {                                                                             \
    if (cfg == 1) hipLaunchKernelGGL((wgrad_flat_kernel<1, ABL_>), dim3(blocks), dim3(256), 0, st, g, bt); \
    else hipLaunchKernelGGL((wgrad_flat_kernel<2, ABL_>), dim3(blocks), dim3(256), 0, st, g, bt);          \
  } else
    SRHIP_FA(1) SRHIP_FA(2) SRHIP_FA(4) SRHIP_FA(8) SRHIP_FA(16) SRHIP_FA(3) SRHIP_FA(5) SRHIP_FA(6) SRHIP_FA(7) SRHIP_FA(31) SRHIP_FA(23)
#undef SRHIP_FA
    if (cfg == 1) hipLaunchKernelGGL((wgrad_flat_kernel<1>), dim3(blocks), dim3(256), 0, st, g, bt);
    else hipLaunchKernelGGL((wgrad_flat_kernel<2>), dim3(blocks), dim3(256), 0, st, g, bt);
  }
  int rc = check_launch("wgrad_flat");
  if (rc) return rc;
  SRHIP_REQUIRE(launch_reduce4_shared(nprob, bt.partial, bt.bias_partial, dw, db, g.nsplit, cout, cin, 9, g.Ktot, accumulate, st, nbias),
                "conv2d_wgrad_pp: reduce launch refused (alignment)");
  return check_launch("wgrad_flat_reduce");
}

}  // namespace srhip
