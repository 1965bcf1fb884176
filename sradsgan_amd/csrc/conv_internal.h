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
int legacy_conv2d_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int n,
                        int h, int w, int cin, int cout, int kh, int kw, int stride, int pad, int ldx, int ldy,
                        void* stream);

// ---- fast path (conv_fast.hip): source channels % 16 == 0, <= 32 taps ------------------------- //
// fprop/dgrad eligibility depends only on the conv's static shape, so the packed-weight layout
// chosen by srhip_pack_weight and the kernel chosen by srhip_conv2d_* always agree.
bool fast_fwd_ok(int cin, int cout, int kh, int kw);      // source channels = cin
bool fast_dgrad_ok(int cin, int cout, int kh, int kw);    // source channels = cout
bool fast_wgrad_ok(int cin, int cout, int kh, int kw);
int fast_pack_weight(const float* w, float* packed, int cout, int cin, int kh, int kw, int mode, hipStream_t st);
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

// column sums (elementwise.hip), used for the bias gradient on the generic path
size_t colsum_workspace_bytes(long rows, int c);
int colsum_launch(const float* dy, float* db, void* workspace, long rows, int c, int ld, hipStream_t st);

}  // namespace srhip
