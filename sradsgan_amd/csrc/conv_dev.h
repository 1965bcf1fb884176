// Device-side helpers shared by the fast convolution translation units (conv_fast.hip, conv_patch_pers.hip):
// geometry structs, the split-bf16 fragment arithmetic, LDS-DMA issue helpers and the patch-tile pixel map.
#pragma once
#include "conv_internal.h"

namespace srhip {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int FBK = 16;            // K values per chunk
constexpr int FLS = FBK + 4;       // LDS row stride in floats (80 B: b128 reads hit all 64 banks once)
constexpr unsigned F_OOB = 0x80000000u;

struct FastGeom {
  int N, Hs, Ws, C, lds;           // source tensor, NHWC, row stride lds (elements)
  int OH, OW, M;                   // virtual output grid, M = N*OH*OW
  int ss;                          // source pixel = (oh*ss + dh, ow*ss + dw)
  int TH, TW, dh0, dhs, dw0, dws;  // tap grid: dh = dh0 + th*dhs, dw = dw0 + tw*dws
  int kh0, khs, kw0, kws, KW;      // weight tap = (kh0 + th*khs)*KW + (kw0 + tw*kws)
  int Hd, Wd, dsd, ph, pw, ldd, K; // destination pixel = (n, oh*dsd + ph, ow*dsd + pw), K channels
  int ldw, ldr;
  float slope;
  int flags, accumulate, dst_identity;
  unsigned src_bytes, w_bytes;
  // padded split-bf16 planes (conv_wgrad_flat.hip) as source / destination of the persistent patch kernel (round 5): the tensor is
  // [guard + N (H+1) (W+1) + tail] pixel rows of channels * 4 bytes (per 8 channels: 8 hi | 8 lo halves); Hs / Ws (Hd / Wd) stay the LOGICAL image size
  int src_pp = 0, dst_pp = 0, src_guard = 0, dst_guard = 0;
  unsigned src_plane_bytes = 0, dst_plane_bytes = 0;
  // a SECOND destination: the fp32 output also as padded planes (the kernels with the row-group epilogue of conv_fast.hip only --
  // the attention tail's 1x1 conv leaves the next block's input x in both forms); NULL: not wanted
  void* dst2_pp = nullptr;
  int dst2_guard = 0;
  // two MORE residuals of the destination's geometry (row stride ldr), added after `residual` in this order: the gradients a block
  // input collects from its other consumers (group skip, trunk bus), folded into the data gradient that is computed last instead of
  // two element-wise add passes (persistent patch kernel with an fp32 destination only; NULL: none)
  const float* res2 = nullptr;
  const float* res3 = nullptr;
};
struct ResRequest {                 // rides beside ONE data-gradient call (conv_api.hip: srhip_conv2d_dgrad_res3 / _pp_res3)
  const float* r2 = nullptr;
  const float* r3 = nullptr;
  int served = 0;
};
extern thread_local ResRequest g_res_req;
struct SignRequest {                // rides beside ONE srhip_conv2d_fwd_pp / _dgrad_pp call (conv_api.hip: the _signs entries): the LeakyReLU
  void* words = nullptr;            // mask as sign words of the persistent patch kernel's tiles (conv_patch_pers.hip, SIGNS)
  size_t bytes = 0;
  int mode = 0;                     // 1: the forward writes them, 2: the masked data gradient reads them
  int served = 0;
};
extern thread_local SignRequest g_sign_req;
long pp_sign_tiles(int n, int h, int w, int cout);   // tiles of the 128-wide persistent walk over [n, h, w, cout] (0: not served); 2048 bytes of sign words each
// The stride^2 phases of a strided data gradient as ONE launch of fast_conv_dma_kernel (round 5): block ranges [first[k], first[k+1])
// run phase k with its own virtual output grid and tap list.  Four quarter-filled launches of the discriminator's 14^2 / 27^2 layers
// become one that fills the chip.  n = 0: a plain launch.
struct PhaseSet {
  int n = 0;
  int first[5] = {0, 0, 0, 0, 0};
  struct P {
    int ph, pw, OH, OW, kh0, kw0, TH, TW, dh0, dw0, M, nblk_m;
  } p[4];
};
struct PhaseRequest {               // rides beside the FIRST phase's run_fast call of fast_conv2d_dgrad; launched = 1: all phases are out
  PhaseSet ps;
  int active = 0, launched = 0;
};
extern thread_local PhaseRequest g_phase_req;
struct Dst2Request {                // rides beside ONE srhip_conv2d_fwd call (conv_api.hip: srhip_conv2d_fwd_dual), like PoolRequest
  void* pp = nullptr;
  int served = 0;
};
extern thread_local Dst2Request g_dst2_req;

__device__ inline float4 bufload4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return __builtin_bit_cast(float4, v);
}

__device__ inline f32x16 mfma32f(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// bijective "XCD b%8 gets a contiguous tile range" remap
__device__ inline int xcd_tile(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// split-bf16 ("bf16x3") product: a*b ~= ah*bh + ah*bl + al*bh with ah = bf16(a), al = bf16(a - ah); the dropped
// al*bl term and the rounding of al/bl are ~2^-16 relative, accumulation stays fp32 in the MFMA.
__device__ inline void split_bf16x8(const float4& v0, const float4& v1, bf16x8_t& hi, bf16x8_t& lo) {
  const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    hi[j] = h;
    lo[j] = (__bf16)(v[j] - (float)h);
  }
}

// 16-bit operand arithmetic of the bf16-layout kernels ("PROD"): 0 = split-bf16, three bf16 products per multiply
// (SRHIP_MATH_BF16X3); 1 = one bf16 product, 2 = one fp16 product (SRHIP_MATH_HALF on gradient resp. activation data).
// Fragments travel as 16-byte bags typed bf16x8_t; PROD 2 reinterprets them as 8 halves.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <int PROD>
__device__ __forceinline__ f32x16 mma16(const bf16x8_t& a, const bf16x8_t& b, const f32x16& c) {
  if constexpr (PROD == 2)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int PROD>
__device__ __forceinline__ bf16x8_t round16x8(const float4& v0, const float4& v1) {
  const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  if constexpr (PROD == 2) {
    f16x8_t h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)v[j];
    return __builtin_bit_cast(bf16x8_t, h);
  } else {
    bf16x8_t h;
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = (__bf16)v[j];
    return h;
  }
}
// products of one (A tile t, B tile u) pair: PROD 0: al*bh, ah*bl, ah*bh interleaved over the tiles by the caller
template <int PROD>
constexpr int nprod() { return PROD == 0 ? 3 : 1; }

__device__ inline void lds_dma16(const float* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst_uniform)
      : "memory");
}

// LDS-DMA through a buffer descriptor: an out-of-range offset (>= num_records) makes the hardware deliver zeros, so
// padding needs no pointer select (one 32-bit offset per lane instead of a 64-bit address and two v_cndmask).
__device__ inline void lds_dma16_buf(unsigned voff, __amdgpu_buffer_rsrc_t r, unsigned lds_dst_uniform) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(r), "s"(lds_dst_uniform) : "memory");
}

struct PatchGeom {
  int PH, PW, tiles_h, tiles_w;    // output patch and patches per image
  int PWP, PR, npieces;            // patch width incl. halo, patch rows, 16-row DMA pieces
  int lo_h, lo_w;                  // source pixel of patch row (0,0) = (oh0 + lo_h, ow0 + lo_w)
  unsigned gmap;                   // 8 x 4 bits: pixel group held by lane-row group i (patch_pixel)
};

template <int N>
__device__ inline void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int I>
struct IC {
  static constexpr int value = I;
};

// Lane-row r (0..127) of a patch tile -> output pixel (orow, ocol) of the PH x PW tile; orow == PH marks a dead row.
// Patch rows are 64 bytes with the 16-byte quad XOR-swizzled by (row >> 2) & 3, and a ds_read_b128 is served in four
// lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): a group is conflict-free exactly when its 16 patch rows are
// distinct mod 16, which 32 CONSECUTIVE patch rows give.  With the plain row-major order (r / PW, r % PW) every run of
// lanes that crosses the end of a tile row jumps by the halo (+2 rows) and lanes collide -- for the 54 x 54 images that
// was every group: 44 % of the LDS-active cycles were bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
// Here the 128 lane-rows are 8 groups of 16: "pixel group" q < PH * (PW / 16) is a 16-pixel run of ONE image row, the
// PW % 16 left-over columns of all rows are packed into the remaining group(s); gmap says which pixel group each
// lane-row group holds, and plan_patch orders them so that the two runs sharing lanes 0-31 / 32-63 of an MFMA row block
// start at patch rows that are congruent mod 16 (rows i and i + 4 for the 20-row pitch of PW = 18).
// PH * PW <= 128 implies at most 8 groups.
__device__ __forceinline__ void patch_pixel(int r, int PH, int PW, unsigned gmap, int& orow, int& ocol) {
  const int a = PW >> 4, b = PW & 15;
  const int grp = (gmap >> (4 * (r >> 4))) & 15, j = r & 15;
  const int nfull = PH * a;
  if (grp < nfull) {
    orow = grp / a;
    ocol = (grp - orow * a) * 16 + j;
  } else if (b > 0) {
    const int idx = (grp - nfull) * 16 + j;
    orow = idx / b;
    ocol = 16 * a + (idx - orow * b);
    if (orow >= PH) { orow = PH; ocol = 0; }
  } else {
    orow = PH;
    ocol = 0;
  }
}

// conv_patch_pers.hip: persistent tile-walking form of conv_patch_kernel; -1 = not applicable (caller launches conv_patch_kernel)
// A pooling request rides beside ONE forward conv call (srhip_conv2d_fwd_pool, conv_api.hip): the persistent patch kernel serves it
// from its epilogue when the geometry allows (64 destination channels, <= POOL_MAXSEG / 2 tiles per image) and reports the number of
// partial segments per image it wrote; otherwise the caller runs the stand-alone pooling kernel.  thread_local: one request per calling thread.
struct PoolRequest {
  float* out = nullptr;            // [sum | max | arg] sections of sec_bytes each, [image][segment][64] inside a section
  unsigned sec_bytes = 0;
  int served_nseg = 0;             // set by the kernel launch that served the request
};
extern thread_local PoolRequest g_pool_req;
int launch_patch_pers(const float* src, const float* wt, const float* bias, const float* residual, const float* actmask,
                      float* dst, const FastGeom& g, const PatchGeom& pg, int nbm, int nbn, bool wide, int prod, int eflags,
                      hipStream_t st);     // g.src_pp / g.dst_pp: src / dst (and, with dst_pp, actmask) point at padded planes

// conv_patch8.hip: the 8-wave two-group patch kernel for >= 256 destination channels; -1 = not applicable
int launch_patch8(const float* src, const float* wt, const float* bias, const float* actmask, float* dst, const FastGeom& g,
                  const PatchGeom& pg, int nbm, int eflags, hipStream_t st);
extern int g_patch8;
extern int g_patch8_abl;

}  // namespace srhip
