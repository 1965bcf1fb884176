/* sradsgan_hip.h -- C ABI of libsradsgan_hip.so (MI355X / gfx950 only).
 *
 * The reference (Meng-333/SRADSGAN) is pure Python on stock PyTorch: it has no FFI layer of its
 * own.  Every entry point below therefore replaces an ATen/cuDNN call that the reference makes
 * implicitly through torch.nn in SRADSGAN/model/sradsgan.py (cited per function as file:line).
 *
 * Conventions
 *   - all tensors are fp32, device memory, caller-allocated; activations are NHWC
 *     ("channels_last": element (n,h,w,c) at ((n*H+h)*W+w)*ld + c, ld >= C);
 *     parameters and parameter gradients keep the framework layout (OIHW) so state_dicts stay
 *     interchangeable with the reference's `.pkl` files;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on that stream and
 *     re-entrant across streams; nothing allocates, frees or synchronises;
 *   - return value 0 = success; otherwise a negative code and srhip_last_error() (thread local)
 *     describes it.  No exceptions cross the boundary.
 */
#ifndef SRADSGAN_HIP_H
#define SRADSGAN_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRHIP_OK 0
#define SRHIP_ERR_ARG (-1)
#define SRHIP_ERR_LAUNCH (-2)
#define SRHIP_ERR_WORKSPACE (-3)

/* epilogue flags of srhip_conv2d_fwd */
#define SRHIP_EPI_BIAS 1      /* y += bias[co]                                             */
#define SRHIP_EPI_LRELU 2     /* y = y > 0 ? y : slope*y   (slope 0 => ReLU)               */
#define SRHIP_EPI_RESIDUAL 4  /* y += residual (after the activation): `out += x`          */
#define SRHIP_EPI_ROWSCALE 8  /* y = rowscale[pixel] * (W.x) (+bias...) : SLAM mask folded */

const char* srhip_last_error(void);
int srhip_abi_version(void);

/* ---- weight packing ------------------------------------------------------------------------ *
 * OIHW parameter -> GEMM "B" operand [KH*KW*Csrc][ld], ld = srhip_packed_ld(Cdst) (zero padded).
 * mode 0 (fprop):  row (kh,kw,ci), col co        = w[co][ci][kh][kw]
 * mode 1 (dgrad):  row (kh,kw,co), col ci        = w[co][ci][KH-1-kh][KW-1-kw]
 * Runs once per optimiser step per conv (weights change every iteration).                      */
int srhip_packed_ld(int cdst);
int srhip_pack_weight(const float* w_oihw, float* packed, int cout, int cin, int kh, int kw, int mode,
                      void* stream);

/* ---- nn.Conv2d forward (sradsgan.py:222-223,233,297,332-336,375,381,384,427,448,476,503;
 *      vgg19.features convs :92-95) with the elementwise tail of its call site fused:
 *      bias, LeakyReLU (:242,:337,:383,:428,:479) and the residual add (:274,:323).
 * x: [N,H,W,ldx>=Cin]  packed: srhip_pack_weight(mode 0)  y: [N,Ho,Wo,ldy>=Cout]
 * residual: [N,Ho,Wo,ldr] or NULL; rowscale: [N*Ho*Wo] or NULL.                                */
int srhip_conv2d_fwd(const float* x, const float* packed, const float* bias, const float* residual,
                     const float* rowscale, float* y, int n, int h, int w, int cin, int cout, int kh,
                     int kw, int stride, int pad, int ldx, int ldy, int ldr, float slope, int flags,
                     void* stream);

/* ---- conv backward-data (autograd of the same call sites; second-order use in
 *      SRADSGAN.gradient_penalty :621,:639).  dy: [N,Ho,Wo,ldy] packed: mode 1  dx: [N,H,W,ldx].
 * accumulate != 0 adds into dx (gradient fan-in of the dense bus :459).                        */
int srhip_conv2d_dgrad(const float* dy, const float* packed, float* dx, int n, int h, int w, int cin,
                       int cout, int kh, int kw, int stride, int pad, int ldy, int ldx, int accumulate,
                       void* stream);

/* ---- conv backward-weight: dw (OIHW) = sum over pixels of dy (x) window(x).  Deterministic
 *      two-pass split-K; `workspace` must hold srhip_conv2d_wgrad_workspace() bytes.           */
size_t srhip_conv2d_wgrad_workspace(int n, int h, int w, int cin, int cout, int kh, int kw, int stride,
                                    int pad);
int srhip_conv2d_wgrad(const float* x, const float* dy, float* dw_oihw, void* workspace,
                       size_t workspace_bytes, int n, int h, int w, int cin, int cout, int kh, int kw,
                       int stride, int pad, int ldx, int ldy, void* stream);

/* ---- bias gradient: db[c] = sum_rows dy[row][c]; workspace >= srhip_colsum_workspace() bytes -- */
size_t srhip_colsum_workspace(long rows, int c);
int srhip_colsum(const float* dy, float* db, void* workspace, size_t workspace_bytes, long rows, int c,
                 int ld, void* stream);

/* ---- elementwise / permutation pieces of the same call sites --------------------------------- */
/* dx = dy * (y > 0 ? 1 : slope): backward of the in-place LeakyReLU (:242,:479) from its OUTPUT  */
int srhip_lrelu_bwd(const float* dy, const float* y, float* dx, long count, float slope, void* stream);
/* nn.PixelShuffle(r) (:382,:385) on NHWC, fused with the LeakyReLU(slope) that follows it (:383):
 * out[n,h*r+i,w*r+j,c] = act(in[n,h,w,c*r*r+i*r+j]);  backward = inverse permutation * mask(out) */
int srhip_pixel_shuffle_fwd(const float* in, float* out, int n, int h, int w, int cout, int r,
                            float slope, int apply_act, void* stream);
int srhip_pixel_shuffle_bwd(const float* dout, const float* out, float* din, int n, int h, int w,
                            int cout, int r, float slope, int apply_act, void* stream);

/* ---- torch.optim.Adam (sradsgan.py:724-725, step at :858 and :887) over a flat fp32 arena, fused
 *      with the discriminator's weight clip `p.data.clamp_(-c, c)` (:891-892; clip <= 0: none).
 * p,g,m,v: [n] arenas (n % 4 == 0, 16-byte aligned); g is multiplied by grad_scale first (1/world
 * for the data-parallel mean).  state: float[4] on the device {step, lr/(1-b1^step),
 * sqrt(1-b2^step), -}; the call advances it on the device, so it is hipGraph-capturable.
 * Arithmetic order follows torch.optim.Adam (eps added after sqrt(v)/sqrt(bias_correction2)).     */
int srhip_adam_step(float* p, const float* g, float* m, float* v, float* state, long n, float lr, float b1,
                    float b2, float eps, float grad_scale, float clip, void* stream);

#ifdef __cplusplus
}
#endif
#endif
