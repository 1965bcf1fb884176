"""torch.autograd wrappers over the C ABI (include/sradsgan_hip.h).

Every op here replaces an ATen call the reference makes implicitly through torch.nn
(SRADSGAN/model/sradsgan.py:35-508).  Activations are NHWC in memory (torch `channels_last`
tensors keep the reference's logical NCHW shape at the module boundary); parameters stay OIHW so
state_dicts stay interchangeable.  All backward passes are themselves built from these Functions,
so the discriminator can be differentiated twice (gradient penalty, sradsgan.py:621,639).

No CPU path exists: a CPU tensor raises.
"""
import contextlib
import ctypes
import threading

import torch
from torch.autograd import Function

from . import _hip

EPI_BIAS, EPI_LRELU, EPI_RESIDUAL, EPI_ROWSCALE = 1, 2, 4, 8
CL = torch.channels_last

class _State:
    # a plain global, not threading.local: autograd runs Function.backward on its own device thread
    skip_param_grads = False


_state = _State()


def _skip_param_grads():
    return _state.skip_param_grads


@contextlib.contextmanager
def no_param_grads():
    """Inside this context conv backward computes only the data gradient.  Used for the first-order
    pass of the gradient penalty (autograd.grad w.r.t. the interpolates only, sradsgan.py:621): a
    custom Function cannot see which outputs the engine needs, so the caller says it."""
    prev = _skip_param_grads()
    _state.skip_param_grads = True
    try:
        yield
    finally:
        _state.skip_param_grads = prev


# --------------------------------------------------------------------------------------------- #
# plumbing
# --------------------------------------------------------------------------------------------- #


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError('%s: sradsgan_amd ops run on the MI355X HIP path only (got a %s tensor); '
                           'there is no CPU fallback' % (what, t.device))
    if t.dtype != torch.float32:
        raise TypeError('%s: fp32 tensors only, got %s' % (what, t.dtype))


def nhwc(t):
    """Dense NHWC memory for a logical NCHW tensor."""
    return t.contiguous(memory_format=CL)


def empty_nhwc(n, c, h, w, like):
    return torch.empty((n, c, h, w), device=like.device, dtype=torch.float32, memory_format=CL)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_pack_epoch = [0]


def bump_weight_epoch():
    """Called by the fused optimiser (it updates parameters through raw pointers, which does not
    bump tensor._version) so cached packed weights are rebuilt."""
    _pack_epoch[0] += 1


def packed_weight(w, mode):
    """OIHW parameter -> GEMM operand (srhip_pack_weight).  Cached per Parameter until it changes."""
    cout, cin, kh, kw = w.shape
    key = (mode, w._version, _pack_epoch[0], w.data_ptr())
    cacheable = isinstance(w, torch.nn.Parameter)
    if cacheable:
        ent = getattr(w, '_srhip_packed', None)
        if ent is not None and ent.get(mode, (None, None))[0] == key:
            return ent[mode][1]
    wd = w.detach().contiguous()
    lib = _hip.lib()
    csrc, cdst = (cin, cout) if mode == 0 else (cout, cin)
    ld = lib.srhip_packed_ld(cdst)
    packed = torch.empty((kh * kw * csrc, ld), device=w.device, dtype=torch.float32)
    _hip.check(lib.srhip_pack_weight(_p(wd), _p(packed), cout, cin, kh, kw, mode, _stream()), 'pack_weight')
    if cacheable:
        if getattr(w, '_srhip_packed', None) is None:
            w._srhip_packed = {}
        w._srhip_packed[mode] = (key, packed)
    return packed


def _out_hw(h, w, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


def conv2d_fwd_raw(x, w, bias, stride, pad, slope=None, residual=None, rowscale=None):
    _require_gpu(x, 'conv2d_fwd')
    x = nhwc(x)
    n, cin, h, wd = x.shape
    cout, cin_w, kh, kw = w.shape
    if cin_w != cin:
        raise ValueError('conv2d_fwd: weight expects %d input channels, got %d' % (cin_w, cin))
    ho, wo = _out_hw(h, wd, kh, stride, pad)
    y = empty_nhwc(n, cout, ho, wo, x)
    flags = 0
    if bias is not None:
        flags |= EPI_BIAS
        bias = bias.detach().contiguous()
    if slope is not None:
        flags |= EPI_LRELU
    if residual is not None:
        residual = nhwc(residual)
        flags |= EPI_RESIDUAL
    if rowscale is not None:
        rowscale = rowscale.contiguous()
        flags |= EPI_ROWSCALE
    lib = _hip.lib()
    _hip.check(lib.srhip_conv2d_fwd(_p(x), _p(packed_weight(w, 0)), _p(bias), _p(residual), _p(rowscale), _p(y),
                                    n, h, wd, cin, cout, kh, kw, stride, pad, cin, cout, cout,
                                    float(slope or 0.0), flags, _stream()), 'conv2d_fwd')
    return y


def conv2d_dgrad_raw(dy, w, x_shape, stride, pad):
    _require_gpu(dy, 'conv2d_dgrad')
    dy = nhwc(dy)
    n, cin, h, wd = x_shape
    cout, _, kh, kw = w.shape
    dx = empty_nhwc(n, cin, h, wd, dy)
    lib = _hip.lib()
    _hip.check(lib.srhip_conv2d_dgrad(_p(dy), _p(packed_weight(w, 1)), _p(dx), n, h, wd, cin, cout, kh, kw, stride,
                                      pad, cout, cin, 0, _stream()), 'conv2d_dgrad')
    return dx


def conv2d_wgrad_raw(x, dy, w_shape, stride, pad):
    _require_gpu(x, 'conv2d_wgrad')
    x, dy = nhwc(x), nhwc(dy)
    n, cin, h, wd = x.shape
    cout, _, kh, kw = w_shape
    lib = _hip.lib()
    nbytes = lib.srhip_conv2d_wgrad_workspace(n, h, wd, cin, cout, kh, kw, stride, pad)
    ws = torch.empty((max(nbytes, 4) + 3) // 4, device=x.device, dtype=torch.float32)
    dw = torch.empty(tuple(w_shape), device=x.device, dtype=torch.float32)
    _hip.check(lib.srhip_conv2d_wgrad(_p(x), _p(dy), _p(dw), _p(ws), ws.numel() * 4, n, h, wd, cin, cout, kh, kw,
                                      stride, pad, cin, cout, _stream()), 'conv2d_wgrad')
    return dw


def colsum_raw(dy):
    """[N,C,H,W] (NHWC memory) -> [C] sum over N,H,W."""
    _require_gpu(dy, 'colsum')
    dy = nhwc(dy)
    n, c, h, w = dy.shape
    rows = n * h * w
    lib = _hip.lib()
    nbytes = lib.srhip_colsum_workspace(rows, c)
    ws = torch.empty((nbytes + 3) // 4, device=dy.device, dtype=torch.float32)
    out = torch.empty(c, device=dy.device, dtype=torch.float32)
    _hip.check(lib.srhip_colsum(_p(dy), _p(out), _p(ws), ws.numel() * 4, rows, c, c, _stream()), 'colsum')
    return out


def lrelu_bwd_raw(dy, y, slope):
    _require_gpu(dy, 'lrelu_bwd')
    dy, y = nhwc(dy), nhwc(y)
    dx = torch.empty_like(dy, memory_format=CL)
    _hip.check(_hip.lib().srhip_lrelu_bwd(_p(dy), _p(y), _p(dx), dy.numel(), float(slope), _stream()), 'lrelu_bwd')
    return dx


# --------------------------------------------------------------------------------------------- #
# differentiable (twice) convolution family: fwd <-> dgrad <-> wgrad are closed under autograd
# --------------------------------------------------------------------------------------------- #


class _LReluBwd(Function):
    """g = dy * (y > 0 ? 1 : slope); linear in dy, the mask is a constant."""

    @staticmethod
    def forward(ctx, dy, y, slope):
        ctx.slope = slope
        ctx.save_for_backward(y)
        return lrelu_bwd_raw(dy, y, slope)

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return _LReluBwd.apply(gg, y, ctx.slope), None, None


class _ColSum(Function):
    @staticmethod
    def forward(ctx, dy):
        ctx.shape = dy.shape
        return colsum_raw(dy)

    @staticmethod
    def backward(ctx, g):
        return g.view(1, -1, 1, 1).expand(ctx.shape)


class _ConvFwd(Function):
    """y = act(conv(x, w) + b) [+ residual]   (act and residual are never combined by the model)."""

    @staticmethod
    def forward(ctx, x, w, b, residual, stride, pad, slope):
        if slope is not None and residual is not None:
            raise ValueError('conv2d: fused activation and residual are mutually exclusive')
        y = conv2d_fwd_raw(x, w, b, stride, pad, slope, residual)
        ctx.stride, ctx.pad, ctx.slope = stride, pad, slope
        ctx.has_bias, ctx.has_res = b is not None, residual is not None
        ctx.save_for_backward(x, w, y if slope is not None else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        g = dy if ctx.slope is None else _LReluBwd.apply(dy, y, ctx.slope)
        skip = _skip_param_grads()
        dx = _ConvDgrad.apply(g, w, tuple(x.shape), ctx.stride, ctx.pad) if ctx.needs_input_grad[0] else None
        dw = _ConvWgrad.apply(x, g, tuple(w.shape), ctx.stride, ctx.pad) if (ctx.needs_input_grad[1] and not skip) else None
        db = _ColSum.apply(g) if (ctx.has_bias and ctx.needs_input_grad[2] and not skip) else None
        dres = dy if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None, None, None


class _ConvDgrad(Function):
    """dx = conv_transpose(dy, w): bilinear in (dy, w)."""

    @staticmethod
    def forward(ctx, dy, w, x_shape, stride, pad):
        ctx.stride, ctx.pad, ctx.x_shape = stride, pad, x_shape
        ctx.save_for_backward(dy, w)
        return conv2d_dgrad_raw(dy, w, x_shape, stride, pad)

    @staticmethod
    def backward(ctx, ddx):
        dy, w = ctx.saved_tensors
        skip = _skip_param_grads()
        d_dy = _ConvFwd.apply(ddx, w, None, None, ctx.stride, ctx.pad, None) if ctx.needs_input_grad[0] else None
        d_w = _ConvWgrad.apply(ddx, dy, tuple(w.shape), ctx.stride, ctx.pad) if (ctx.needs_input_grad[1] and not skip) else None
        return d_dy, d_w, None, None, None


class _ConvWgrad(Function):
    """dw = wgrad(x, dy): bilinear in (x, dy)."""

    @staticmethod
    def forward(ctx, x, dy, w_shape, stride, pad):
        ctx.stride, ctx.pad, ctx.w_shape = stride, pad, w_shape
        ctx.save_for_backward(x, dy)
        return conv2d_wgrad_raw(x, dy, w_shape, stride, pad)

    @staticmethod
    def backward(ctx, ddw):
        x, dy = ctx.saved_tensors
        ddw = ddw.contiguous()
        d_x = _ConvDgrad.apply(dy, ddw, tuple(x.shape), ctx.stride, ctx.pad) if ctx.needs_input_grad[0] else None
        d_dy = _ConvFwd.apply(x, ddw, None, None, ctx.stride, ctx.pad, None) if ctx.needs_input_grad[1] else None
        return d_x, d_dy, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0, act_slope=None, residual=None):
    """nn.Conv2d forward with the elementwise tail of its call site fused (bias, LeakyReLU/ReLU,
    residual add).  x: logical NCHW; returns logical NCHW in NHWC memory."""
    return _ConvFwd.apply(x, weight, bias, residual, stride, padding, act_slope)


# --------------------------------------------------------------------------------------------- #
# pixel shuffle (+ LeakyReLU) -- sradsgan.py:381-386
# --------------------------------------------------------------------------------------------- #


class _PixelShuffleAct(Function):
    @staticmethod
    def forward(ctx, x, r, slope):
        _require_gpu(x, 'pixel_shuffle')
        x = nhwc(x)
        n, c, h, w = x.shape
        cout = c // (r * r)
        out = empty_nhwc(n, cout, h * r, w * r, x)
        _hip.check(_hip.lib().srhip_pixel_shuffle_fwd(_p(x), _p(out), n, h, w, cout, r, float(slope or 0.0),
                                                      int(slope is not None), _stream()), 'pixel_shuffle_fwd')
        ctx.r, ctx.slope, ctx.in_shape = r, slope, (n, c, h, w)
        ctx.save_for_backward(out if slope is not None else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dout = nhwc(dout)
        n, c, h, w = ctx.in_shape
        din = empty_nhwc(n, c, h, w, dout)
        _hip.check(_hip.lib().srhip_pixel_shuffle_bwd(_p(dout), _p(out), _p(din), n, h, w, c // (ctx.r * ctx.r), ctx.r,
                                                      float(ctx.slope or 0.0), int(ctx.slope is not None), _stream()),
                   'pixel_shuffle_bwd')
        return din, None, None


def pixel_shuffle_act(x, r, slope=None):
    return _PixelShuffleAct.apply(x, r, slope)


# --------------------------------------------------------------------------------------------- #
# attention / normalisation glue.  TRANSITIONAL: the functions below are still compositions of
# torch device ops (ATen HIP kernels) around the HIP convs; each is being replaced by a fused HIP
# kernel behind the same Python signature (DESIGN.md lists what is still on ATen).
# --------------------------------------------------------------------------------------------- #


def _mlp_1x1(v, fc1_w, fc2_w):
    """v: [B,C,1,1]; the two bias-free 1x1 convs of CLAM (sradsgan.py:110-112) as matmuls."""
    b, c = v.shape[0], v.shape[1]
    hid = torch.relu(v.reshape(b, c) @ fc1_w.reshape(fc1_w.shape[0], c).t())
    return (hid @ fc2_w.reshape(c, fc1_w.shape[0]).t()).reshape(b, c, 1, 1)


def clam(x, fc1_w, fc2_w, pool_mode='Avg|Max'):
    """sradsgan.py:117-127 / base_networks.py:387-403."""
    logits = 0
    if 'Avg' in pool_mode:
        logits = logits + _mlp_1x1(x.mean((2, 3), keepdim=True), fc1_w, fc2_w)
    if 'Max' in pool_mode:
        logits = logits + _mlp_1x1(torch.nn.functional.adaptive_max_pool2d(x, 1), fc1_w, fc2_w)
    return torch.sigmoid(logits) * x


def slam(x, w7, pool_mode='Avg|Max'):
    """sradsgan.py:141-151 / base_networks.py:440-457: the 7x7 (2->1) conv runs on the HIP igemm."""
    maps = []
    if 'Avg' in pool_mode:
        maps.append(x.mean(dim=1, keepdim=True))
    if 'Max' in pool_mode:
        maps.append(x.max(dim=1, keepdim=True)[0])
    pooled = torch.cat(maps, dim=1)
    return torch.sigmoid(conv2d(pooled, w7, None, 1, w7.shape[-1] // 2)) * x


def cgam(x, gamma):
    """sradsgan.py:202-212; softmax(rowmax(E)-E) == softmax(-E) (shift invariance)."""
    b, c, h, w = x.shape
    xf = x.permute(0, 2, 3, 1).reshape(b, h * w, c)            # NHWC memory: free view [b, n, c]
    energy = xf.transpose(1, 2) @ xf                            # [b, c, c]
    att = torch.softmax(energy.max(dim=-1, keepdim=True)[0] - energy, dim=-1)
    out = xf @ att.transpose(1, 2)                              # [b, n, c]
    return gamma * out.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


def sgam(x, q, k, v, gamma):
    """sradsgan.py:165-175 with q,k,v already projected (NHWC memory). Materialises N x N for now."""
    b, c, h, w = x.shape
    n = h * w
    qf = q.permute(0, 2, 3, 1).reshape(b, n, -1)
    kf = k.permute(0, 2, 3, 1).reshape(b, n, -1)
    vf = v.permute(0, 2, 3, 1).reshape(b, n, c)
    att = torch.softmax(qf @ kf.transpose(1, 2), dim=-1)        # [b, n(query), n(key)]
    out = att @ vf                                              # [b, n, c]
    return gamma * out.reshape(b, h, w, c).permute(0, 3, 1, 2) + x


def batch_norm_act(x, bn, slope=None):
    """Train-mode BatchNorm2d + LeakyReLU (sradsgan.py:478-479); updates running stats like
    nn.BatchNorm2d (momentum 0.1, unbiased running_var).  Twice differentiable."""
    if not bn.training:
        raise NotImplementedError('the reference never puts the discriminator in eval() (SURVEY a11)')
    n = x.numel() // x.shape[1]
    mean = x.mean((0, 2, 3))
    var = x.var((0, 2, 3), unbiased=False)
    with torch.no_grad():
        m = bn.momentum
        bn.running_mean.mul_(1 - m).add_(mean, alpha=m)
        bn.running_var.mul_(1 - m).add_(var * (n / max(n - 1, 1)), alpha=m)
        bn.num_batches_tracked += 1
    inv = torch.rsqrt(var + bn.eps)
    y = (x - mean.view(1, -1, 1, 1)) * (inv * bn.weight).view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1)
    return y if slope is None else torch.nn.functional.leaky_relu(y, slope)


def max_pool2x2(x):
    return torch.nn.functional.max_pool2d(x, 2, 2)


def l1_mean(a, b):
    """nn.L1Loss() (sradsgan.py:686,834,838)."""
    return (a - b).abs().mean()


def gp_penalty(grads):
    """sradsgan.py:630-637: L2 norm over the channel dim (per pixel), LS penalty, mean."""
    return (grads.norm(2, 1) - 1).pow(2).mean()
