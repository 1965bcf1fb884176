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

// A fast packed weight holds three sections of `total` floats each:
//   [0, total)         fp32, row = destination channel, columns (tap, source channel)        (fp32-MFMA kernels)
//   [total, 2 total)   the same matrix pre-split for the split-bf16 kernels: every aligned group of 8 columns is
//                      stored as 8 bf16 "hi" (round-to-nearest of v) followed by 8 bf16 "lo" (rn of v - hi), so the
//                      two 16-byte quads a lane reads are its MFMA B operands with no conversion in the kernel;
//                      the single-product bf16 arithmetic (SRHIP_MATH_HALF on gradient data) reads the hi quads only
//   [2 total, 3 total) the same geometry with 8 fp16 values in the hi quad (lo quad zero): SRHIP_MATH_HALF fprop
__device__ inline void fast_pack_store(float* packed, long total, long idx, float v) {
  packed[idx] = v;
  const __bf16 h = (__bf16)v;
  const __bf16 l = (__bf16)(v - (float)h);
  __bf16* sp = reinterpret_cast<__bf16*>(packed + total) + (idx >> 3) * 16 + (idx & 7);
  sp[0] = h;
  sp[8] = l;
  _Float16* hp = reinterpret_cast<_Float16*>(packed + 2 * total) + (idx >> 3) * 16 + (idx & 7);
  hp[0] = (_Float16)v;
  hp[8] = (_Float16)0.f;
}
int fast_conv2d_fwd(const float* x, const float* packed, const float* bias, const float* residual,
                    const float* rowscale, const float* chanscale, float* y, int n, int h, int w, int cin, int cout, int kh, int kw,
                    int stride, int pad, int ldx, int ldy, int ldr, float slope, int flags, hipStream_t st);
int fast_conv2d_dgrad(const float* dy, const float* packed, float* dx, const float* residual, const float* actmask,
                      float slope, int n, int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int ldy,
                      int ldx, int ldr, int accumulate, hipStream_t st);
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

// column sums (elementwise.hip), used for the bias gradient on the generic path
size_t colsum_workspace_bytes(long rows, int c);
int colsum_launch(const float* dy, float* db, void* workspace, long rows, int c, int ld, hipStream_t st);

}  // namespace srhip
