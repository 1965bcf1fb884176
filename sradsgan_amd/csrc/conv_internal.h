// Internal declarations shared by the convolution translation units (not part of the C ABI).
#pragma once
#include "common.h"

namespace srhip {

// ---- generic implicit-GEMM path (conv_igemm.hip): any channel count / kernel size ------------- //
int legacy_packed_ld(int cdst);
int legacy_pack_weight(const float* w, float* packed, int cout, int cin, int kh, int kw, int mode, void* stream);
int legacy_conv2d_fwd(const float* x, const float* packed, const float* bias, const float* residual,
                      const float* rowscale, float* y, int n, int h, int w, int cin, int cout, int kh, int kw,
                      int stride, int pad, int ldx, int ldy, int ldr, float slope, int flags, void* stream);
int legacy_conv2d_dgrad(const float* dy, const float* packed, float* dx, int n, int h, int w, int cin, int cout,
                        int kh, int kw, int stride, int pad, int ldy, int ldx, int accumulate, void* stream);
size_t legacy_conv2d_wgrad_workspace(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad);
size_t legacy_conv2d_wgrad_bias_workspace(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad);
int legacy_conv2d_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int n,
                        int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int ldx, int ldy,
                        void* stream, float* db = nullptr, float* bias_ws = nullptr, int* bias_done = nullptr,
                        const float* ymask = nullptr, float slope = 0.f);

// ---- fast path (conv_fast.hip): source channels % 16 == 0, <= 32 taps ------------------------- //
// fprop/dgrad eligibility depends only on the conv's static shape, so the packed-weight layout
// chosen by srhip_pack_weight and the kernel chosen by srhip_conv2d_* always agree.
bool fast_fwd_ok(int cin, int cout, int kh, int kw);      // source channels = cin
bool fast_dgrad_ok(int cin, int cout, int kh, int kw);    // source channels = cout
bool fast_wgrad_ok(int cin, int cout, int kh, int kw);
int fast_pack_weight(const float* w, float* packed, int cout, int cin, int kh, int kw, int mode, hipStream_t st);

// A fast packed weight holds four sections; the first three have `total` floats each:
//   [0, total)         fp32, row = destination channel, columns (tap, source channel)        (fp32-MFMA kernels)
//   [total, 2 total)   the same matrix pre-split for the split-bf16 kernels: every aligned group of 8 columns is
//                      stored as 8 bf16 "hi" (round-to-nearest of v) followed by 8 bf16 "lo" (rn of v - hi), so the
//                      two 16-byte quads a lane reads are its MFMA B operands with no conversion in the kernel;
//                      the single-product bf16 arithmetic (SRHIP_MATH_HALF on gradient data) reads the hi quads only
//   [2 total, 3 total) the same geometry with 8 fp16 values in the hi quad (lo quad zero): SRHIP_MATH_HALF fprop
//   [3 total, 3 total + fast_tiled_elems)  round 4: the pre-split matrix again, TILED for conv_patch_pers_kernel: one 64-byte
//                      row per (16-channel chunk cc, tap, destination channel n), rows ordered [cc][tap][n] with n padded
//                      to a multiple of 16 and the four 16-byte quads of a row already XOR-swizzled by (n >> 2) & 3 -- the
//                      weight tile of one (chunk, tap) is then ONE contiguous run of full cache lines that an LDS-DMA copies
//                      lane-linearly (the row-major section serves the same tile as 64-byte pieces of 2304-byte rows: half
//                      of every 128-byte line fetched from L2 was the next chunk's, 9 taps too early to be kept)
inline long fast_tiled_elems(int ndst, int csrc, int khkw) { return (long)khkw * csrc * (((long)ndst + 15) / 16 * 16); }
__device__ inline void fast_pack_store(float* packed, long total, long idx, float v, int ndst, int csrc, int khkw) {
  packed[idx] = v;
  const __bf16 h = (__bf16)v;
  const __bf16 l = (__bf16)(v - (float)h);
  __bf16* sp = reinterpret_cast<__bf16*>(packed + total) + (idx >> 3) * 16 + (idx & 7);
  sp[0] = h;
  sp[8] = l;
  _Float16* hp = reinterpret_cast<_Float16*>(packed + 2 * total) + (idx >> 3) * 16 + (idx & 7);
  hp[0] = (_Float16)v;
  hp[8] = (_Float16)0.f;
  const int ktot = khkw * csrc;
  const int n = (int)(idx / ktot), kcol = (int)(idx - (long)n * ktot);
  const int tap = kcol / csrc, c = kcol - tap * csrc;
  const int cc = c >> 4, k16 = c & 15;
  const long ndst16 = ((long)ndst + 15) / 16 * 16;
  const int q = 2 * (k16 >> 3), swz = (n >> 2) & 3;
  __bf16* tp = reinterpret_cast<__bf16*>(packed + 3 * total) + (((long)cc * khkw + tap) * ndst16 + n) * 32 + (k16 & 7);
  tp[(q ^ swz) * 8] = h;
  tp[((q + 1) ^ swz) * 8] = l;
}
int fast_conv2d_fwd(const float* x, const float* packed, const float* bias, const float* residual,
                    const float* rowscale, const float* chanscale, float* y, int n, int h, int w, int cin, int cout, int kh, int kw,
                    int stride, int pad, int ldx, int ldy, int ldr, float slope, int flags, hipStream_t st);
int fast_conv2d_dgrad(const float* dy, const float* packed, float* dx, const float* residual, const float* actmask,
                      float slope, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int ldy,
                      int ldx, int ldr, int accumulate, hipStream_t st);
int fast_conv2d_fwd_pp(const void* x, int x_pp, const float* packed, const float* bias, void* y, int y_pp, int n, int h, int w, int cin,
                       int cout, int ldx, int ldy, float slope, int flags, hipStream_t st);
int fast_conv2d_dgrad_pp(const void* dy, int dy_pp, const float* packed, void* dx, int dx_pp, const float* residual, const void* actmask,
                         float slope, int n, int h, int w, int cin, int cout, int ldy, int ldx, int ldr, hipStream_t st);
size_t fast_conv2d_wgrad_workspace(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad);
int fast_conv2d_wgrad(const float* x, const float* dy, float* dw, float* db, const float* xrow, const float* xchan,
                      int accumulate, void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int ldx, int ldy,
                      hipStream_t st);

int fast_wgrad_multi_ok(int cin, int cout, int kh, int kw, int stride, int pad);
int fast_wgrad_multi_max(int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad);
size_t fast_conv2d_wgrad_multi_workspace(int nprob, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad);
int fast_conv2d_wgrad_multi(int nprob, const float* const* x, const float* const* dy, float* const* dw, float* const* db,
                            int accumulate, void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout,
                            int kh, int kw, int stride, int pad, int ldx, int ldy, hipStream_t st);

// ---- padded split-bf16 planes and the flat weight gradient on them (conv_wgrad_flat.hip, round 5) ---- //
int pp_guard(int w);
long pp_plane_pixels(int n, int h, int w);
int pp_from_f32(const float* x, void* pp, int n, int h, int w, int c, int ldx, void* stream);
int pp_to_f32(const void* pp, float* x, int n, int h, int w, int c, int ldx, void* stream);
int flat_wgrad_ok(int n, int h, int w, int cin, int cout);      // bit mask of served operand formats: 1 x fp32 + dy planes, 2 x planes + dy fp32, 4 both planes
size_t flat_wgrad_workspace(int nprob, int x_pp, int dy_pp, int n, int h, int w, int cin, int cout);
int flat_wgrad(int nprob, const void* const* x, const void* const* dy, int x_pp, int dy_pp, float* const* dw, float* const* db, int accumulate,
               void* workspace, size_t workspace_bytes, int n, int h, int w, int cin, int cout, int ldf, void* stream);
extern int g_flat_blocks;
extern int g_flat_abl;
extern int g_flat_f32_k8;

// column sums (elementwise.hip), used for the bias gradient on the generic path
size_t colsum_workspace_bytes(long rows, int c);
int colsum_launch(const float* dy, float* db, void* workspace, long rows, int c, int ld, hipStream_t st);

}  // namespace srhip
