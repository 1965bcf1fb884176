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
import os
import threading
import weakref

import torch
from torch.autograd import Function

from . import _hip

EPI_BIAS, EPI_LRELU, EPI_RESIDUAL, EPI_ROWSCALE, EPI_CHANSCALE, EPI_GRADDATA = 1, 2, 4, 8, 16, 64
CL = torch.channels_last

class _State:
    # a plain global, not threading.local: autograd runs Function.backward on its own device thread
    skip_param_grads = False
    direct_grads = False
    wgrad_stream = None             # side HIP stream for weight-gradient kernels (direct_param_grads mode only)
    skip_ids = frozenset()          # id()s of parameters whose gradients the current backward must not produce
    stop_ids = frozenset()          # data_ptr()s of tensors the current backward must not propagate into
    capturing = False               # a stream capture is being recorded (TrainStep._capture): side-stream forks go through torch events
    wgrad_group = 1                 # weight gradients of one shape launched together (srhip_conv2d_wgrad_multi); 1 = off
    pending = None                  # shape key -> [(x, dy, gw, gb, stride, pad)] waiting for partners (direct_param_grads mode)
    held = None                     # gradients kept referenced until the backward ends, see _passed_through
    last_out_pp = None              # _RabBlock.forward -> rab_block(): the output's padded planes (emit_pp)
    carry = None                    # token -> [gradients stashed for a block input by its other consumers] (carry_open)
    carry_expect = None             # token -> number of consumers that committed to stash at forward time
    carry_token = 0
    wgrad_seq = 0                   # weight-gradient requests so far (the age of a pending launch, _age_pending)
    ready_pairs = None              # complete flat-kernel launches waiting for their slot behind a conv1 data gradient (_WGRAD_SLOTS)


_state = _State()


def _skip_param_grads(*params):
    """True when the running backward must not produce gradients for these parameters."""
    if _state.skip_param_grads:
        return True
    return bool(_state.skip_ids) and any(p is not None and id(p) in _state.skip_ids for p in params)


@contextlib.contextmanager
def backward_scope(skip_params=(), stop_at=()):
    """Prunes a backward pass that autograd cannot prune inside custom Functions: parameters listed in
    `skip_params` get no weight gradient (their wgrad kernels are not launched), and the data gradient
    is not propagated into the tensors listed in `stop_at`."""
    prev = (_state.skip_ids, _state.stop_ids)
    _state.skip_ids = frozenset(id(p) for p in skip_params)
    _state.stop_ids = frozenset(t.data_ptr() for t in stop_at)
    try:
        yield
    finally:
        _state.skip_ids, _state.stop_ids = prev


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


@contextlib.contextmanager
def direct_param_grads(side_stream=None, group=1):
    """Inside this context the fused backward kernels ACCUMULATE parameter gradients straight into the
    parameters' existing .grad buffers (the gradient arena of dp.ParamArena) and hand autograd None for
    them: one wgrad launch per conv instead of wgrad + one `grad += new` launch per parameter
    (~600 tiny launches per step).  Only valid when every such parameter already owns a dense .grad
    and nobody asks autograd for these gradients explicitly (TrainStep guarantees both)."""
    prev = (_state.direct_grads, _state.wgrad_stream, _state.wgrad_group, _state.pending, _state.held, _state.ready_pairs)
    _state.direct_grads, _state.wgrad_stream = True, side_stream
    _state.wgrad_group, _state.pending, _state.held, _state.ready_pairs = (group if side_stream is not None else 1), {}, [], []
    try:
        yield
        flush_pending_wgrads()
        if _state.pending:                             # before the restore below wipes the evidence
            raise RuntimeError('direct_param_grads: %d grouped weight gradient(s) still waiting for a partner after the flush'
                               % len(_state.pending))
    finally:
        _state.direct_grads, _state.wgrad_stream, _state.wgrad_group, _state.pending, _state.held, _state.ready_pairs = prev


_HOLD = int(os.environ.get('SRHIP_HOLD', '1'))     # 1: passed-through gradients (default); 2: every side-stream operand; 0: off (test knob)


def _hold_for_side(side, *tensors):
    """SRHIP_HOLD=2 only: keeps EVERY operand gradient of a side-stream launch referenced until the backward ends (see
    _passed_through for why a reference matters).  The default holds only the gradients that are known to be aliased; holding
    all of them costs 1.5 % of the step (the allocator hands out cold blocks instead of the ones just freed:
    profiles/r04_step_ab_small_changes.txt) and protects against graph shapes the model does not have -- a gradient that one
    producer hands to several consumers (torch's AddBackward, ops._SumN) is shared until the last of them has run, and the
    trunk's bus gradient is held by the head convs' buffers until the end of the backward."""
    if _HOLD >= 2 and _state.held is not None:
        _state.held.extend(t for t in tensors if t is not None)


def _passed_through(g):
    """Call on an incoming gradient that a backward returns UNCHANGED as the gradient of one of its inputs while a weight-gradient
    kernel on the side stream reads it.  record_stream() protects the tensor's MEMORY; this protects its CONTENTS: autograd's
    input buffer adds later contributions IN PLACE into a gradient it holds the only reference to
    (torch/csrc/autograd/input_buffer.cpp, use_count() == 1), on the main stream, which the side stream may trail by a
    millisecond.  The attention tail hands `g` back as the gradient of `skip`, a conv with a fused residual hands it back as the
    residual's: when the ResGroup's first RAB delivered its data gradient 1 ms later the engine ran `g += dx` under the tail
    conv's weight-gradient kernel -- ResGroup tail conv gradients off by 10-30 % whenever the side stream lagged
    (tests/test_model_gpu.py: the first-step test after two pool streams, and the lagging-side-stream test).  With a second
    reference alive until the backward ends the engine allocates the sum instead, which is what it does for every gradient
    that is not aliased (12 + a few tensors of 24 MB per step)."""
    if _HOLD and _state.wgrad_stream is not None and _state.held is not None and g is not None:
        _state.held.append(g)
    return g


_CARRY = os.environ.get('SRHIP_CARRY', '1') == '1'          # A/B knob: 0 = autograd sums the gradients of a group input (two add passes per group)


def carry_open(x):
    """Called by a consumer structure (model.ResGroup) on a block input `x` whose FIRST consumer is a fused RAB (rab_block): the input's
    other consumers -- the group's skip connection (attention_tail) and the trunk's bus (sum_tensors) -- then do not hand their gradient
    to autograd (which would add the three contributions with two element-wise passes over the 24 MB tensor) but stash it, and the RAB's
    conv1 data gradient, computed last, takes them as extra residuals of its epilogue (srhip_conv2d_dgrad_res3 / _pp_res3).  The order of
    the backward makes this safe: the RAB's backward depends on the group tail's, and the whole trunk's on the bus's; it is CHECKED:
    the RAB counts the stashes it finds against the consumers that committed at forward time and raises when one is missing.
    The tag travels as an attribute of the tensor object (a fresh token per call: no stale matches)."""
    if not _CARRY or not x.is_cuda or not (torch.is_grad_enabled() and x.requires_grad):
        return
    if _state.carry is None or len(_state.carry_expect) > 4096:     # (forwards that never ran a backward leave their counters behind)
        _state.carry, _state.carry_expect = {}, {}
    _state.carry_token += 1
    x._srhip_carry = _state.carry_token
    _state.carry_expect[_state.carry_token] = 0


def _carry_commit(t):
    """forward time, a consumer that can stash: returns the token it will stash under (None: the tensor carries no tag)."""
    token = getattr(t, '_srhip_carry', None) if _CARRY else None
    if token is None or _state.carry_expect is None or token not in _state.carry_expect or not (torch.is_grad_enabled() and t.requires_grad):
        return None
    _state.carry_expect[token] += 1
    return token


def _carry_stash(token, g):
    """backward time: True = `g` is stashed for the RAB (return None to autograd), False = hand it to autograd as usual."""
    if token is None or token not in _state.carry_expect:
        return False
    if torch.is_grad_enabled():                                  # a differentiable backward: autograd must see the edge
        _state.carry_expect[token] -= 1
        return False
    _state.carry.setdefault(token, []).append(g)
    return True


def _carry_take(token):
    if token is None or _state.carry_expect is None:
        return ()
    extras = _state.carry.pop(token, [])
    expect = _state.carry_expect.pop(token, 0)
    if len(extras) != expect:
        raise RuntimeError('rab_block backward: %d of %d gradients of the block input have arrived -- a consumer that committed to '
                           'stash its gradient (ops.carry_open) has not run its backward yet' % (len(extras), expect))
    return tuple(extras)


def _grad_slot(p):
    """The buffer to accumulate into, or None when direct accumulation is off / impossible for p."""
    if not _state.direct_grads or p is None or not p.requires_grad:
        return None
    g = p.grad
    if g is None or not g.is_contiguous() or g.dtype != torch.float32 or g.device != p.device:
        return None
    return g


# --------------------------------------------------------------------------------------------- #
# plumbing
# --------------------------------------------------------------------------------------------- #


CONV_MATH_MODES = {'fp32': 0, 'bf16x3': 1, 'half': 2}


def set_conv_math(mode):
    """Arithmetic of the conv contraction (include/sradsgan_hip.h, srhip_set_conv_math): 'fp32' = fp32 MFMA, 'bf16x3' =
    split-bf16 (three bf16 MFMA products per fp32 product, fp32 accumulate), 'half' = ONE 16-bit product (fp16 for
    activations, bf16 wherever gradients are multiplied; BASELINE configs[4], outside the 1e-3 parity contract).  Stands
    where torch.backends.cudnn.allow_tf32 stands for the reference's nn.Conv2d.  Returns the previous mode."""
    if mode not in CONV_MATH_MODES:
        raise ValueError('conv math mode must be one of %s' % sorted(CONV_MATH_MODES))
    lib = _hip.lib()
    prev = lib.srhip_get_conv_math()
    _hip.check(lib.srhip_set_conv_math(CONV_MATH_MODES[mode]))
    return [k for k, v in CONV_MATH_MODES.items() if v == prev][0]


def get_conv_math():
    return [k for k, v in CONV_MATH_MODES.items() if v == _hip.lib().srhip_get_conv_math()][0]


@contextlib.contextmanager
def conv_math(mode):
    prev = set_conv_math(mode)
    try:
        yield
    finally:
        set_conv_math(prev)


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
    # the raw handle of torch's current stream: torch.cuda.current_stream() builds a Stream object per call (~9 us, 300
    # calls per step on each of the two launching threads)
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


class _PackRegistry:
    """Persistent packed images of trainable conv weights.  Each (Parameter, mode) gets ONE buffer that is
    rewritten in place: by a single batched launch after every optimiser step (repack_all, called from
    bump_weight_epoch) instead of ~280 tiny pack launches scattered through the next iteration."""

    def __init__(self):
        self.entries = []        # [weakref(w), mode, packed, data_ptr, version, pack event or None, packing stream, streams that waited]
        self.table = None
        self.dirty = True


_registry = _PackRegistry()


def _pack_now(w, mode, packed):
    cout, cin, kh, kw = w.shape
    wd = w.detach().contiguous()
    _hip.check(_hip.lib().srhip_pack_weight(_p(wd), _p(packed), cout, cin, kh, kw, mode, _stream()), 'pack_weight')


def repack_all():
    reg = _registry
    if any(ent[0]() is None for ent in reg.entries):            # parameters of discarded models
        reg.entries = [ent for ent in reg.entries if ent[0]() is not None]
        reg.dirty = True
    if not reg.entries:
        return
    lib = _hip.lib()
    live = [ent[0]() for ent in reg.entries]
    if reg.dirty or reg.table is None:
        import struct
        assert lib.srhip_pack_entry_bytes() == 40
        blob = bytearray()
        for ent, w in zip(reg.entries, live):
            mode, packed = ent[1], ent[2]
            cout, cin, kh, kw = w.shape
            fast = lib.srhip_packed_is_fast(cout, cin, kh, kw, mode)
            blob += struct.pack('<QQiiiiii', w.data_ptr(), packed.data_ptr(), cout, cin, kh, kw, mode, fast)
            ent[3] = w.data_ptr()
        reg.table = torch.frombuffer(blob, dtype=torch.uint8).to(live[0].device)
        reg.dirty = False
    _hip.check(lib.srhip_pack_weights_batched(_p(reg.table), len(reg.entries), _stream()), 'pack_weights_batched')
    for ent, w in zip(reg.entries, live):
        ent[4] = w._version
        ent[5] = None            # re-packed here, at the step boundary every other stream synchronises with (see _pack_fence)


def bump_weight_epoch():
    """Called by the fused optimiser (it updates parameters through raw pointers, which does not bump
    tensor._version): re-packs every registered conv weight in one launch."""
    repack_all()


_NO_PACK_FENCE = os.environ.get('SRHIP_NO_PACK_FENCE') == '1'     # debug: reproduces the unordered first step


def _pack_fence(ent):
    """A packed image is written by a kernel on the stream that first needed it (lazily, in the first step: the weight-
    gradient stream packs VGG's weights for the real batch, the D stream packs the discriminator's data-gradient images
    for the penalty's first-order backward) and read by every stream afterwards.  Nothing else orders those reads behind
    the pack kernel: in the first step of a model the main stream could run D(gen)'s data gradients on images the D stream
    had not packed yet -- zeros on fresh memory (a silently wrong first step), NaNs on recycled memory
    (tests/test_model_gpu.py::test_first_step_of_a_model_does_not_depend_on_allocator_history).  The entry carries the pack's event; a stream waits for it once."""
    ev = ent[5]
    if ev is None or _NO_PACK_FENCE:
        return
    cs = torch.cuda.current_stream()
    sid = cs.cuda_stream
    if sid == ent[6] or sid in ent[7]:
        return
    if torch.cuda.is_current_stream_capturing():
        return                   # a capture starts long after the eager warm-up iteration that packed
    cs.wait_event(ev)
    ent[7].add(sid)


def _pack_event(ent):
    ev = torch.cuda.Event()
    ev.record()
    ent[5], ent[6], ent[7] = ev, torch.cuda.current_stream().cuda_stream, set()


def mark_static(module):
    """Parameters of `module` are never updated (VGG): pack once, keep out of the per-step batched repack."""
    for p in module.parameters():
        p._srhip_static = True


def packed_weight(w, mode):
    """OIHW weight -> GEMM operand.  Parameters get a persistent buffer (see _PackRegistry); other tensors
    (e.g. the weight cotangent of a second-order pass) are packed on the fly."""
    cout, cin, kh, kw = w.shape
    lib = _hip.lib()
    if not isinstance(w, torch.nn.Parameter):
        packed = torch.empty(lib.srhip_packed_elems(cout, cin, kh, kw, mode), device=w.device, dtype=torch.float32)
        _pack_now(w, mode, packed)
        return packed
    slots = getattr(w, '_srhip_packed', None)
    if slots is None:
        slots = w._srhip_packed = {}
    ent = slots.get(mode)
    if ent is None or ent[3] != w.data_ptr():
        packed = torch.empty(lib.srhip_packed_elems(cout, cin, kh, kw, mode), device=w.device, dtype=torch.float32)
        _pack_now(w, mode, packed)
        new = [weakref.ref(w), mode, packed, w.data_ptr(), w._version, None, 0, set()]
        if w.is_cuda and not torch.cuda.is_current_stream_capturing():
            _pack_event(new)
        if ent is not None and ent in _registry.entries:
            _registry.entries.remove(ent)
        slots[mode] = new
        if not getattr(w, '_srhip_static', False):
            _registry.entries.append(new)
            _registry.dirty = True
        return packed
    if ent[4] != w._version:                       # changed by a torch op (load_state_dict, torch optimiser, ...)
        _pack_fence(ent)                           # (write after the reads other streams may still have in flight is the caller's order)
        _pack_now(w, mode, ent[2])
        ent[4] = w._version
        if not torch.cuda.is_current_stream_capturing():
            _pack_event(ent)
    else:
        _pack_fence(ent)
    return ent[2]


def _out_hw(h, w, k, stride, pad):
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


def conv2d_fwd_raw(x, w, bias, stride, pad, slope=None, residual=None, rowscale=None, chanscale=None, graddata=False, out_pp=None):
    """graddata: x holds gradients (a second-order pass): 'half' arithmetic then rounds to bf16 instead of fp16.
    out_pp: a PP buffer that ALSO receives y as padded planes (srhip_conv2d_fwd_dual; a conversion pass when the kernel that took the
    launch has no second destination)."""
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
    if chanscale is not None:
        chanscale = chanscale.contiguous()
        flags |= EPI_CHANSCALE
    if graddata:
        flags |= EPI_GRADDATA
    lib = _hip.lib()
    if out_pp is not None:
        served = ctypes.c_int(0)
        _hip.check(lib.srhip_conv2d_fwd_dual(_p(x), _p(packed_weight(w, 0)), _p(bias), _p(residual), _p(rowscale),
                                             _p(chanscale), _p(y), _p(out_pp.buf), ctypes.byref(served),
                                             n, h, wd, cin, cout, kh, kw, stride, pad, cin, cout, cout,
                                             float(slope or 0.0), flags, _stream()), 'conv2d_fwd_dual')
        if not served.value:
            pp_from_f32(y, out=out_pp)
        return y
    _hip.check(lib.srhip_conv2d_fwd(_p(x), _p(packed_weight(w, 0)), _p(bias), _p(residual), _p(rowscale),
                                    _p(chanscale), _p(y),
                                    n, h, wd, cin, cout, kh, kw, stride, pad, cin, cout, cout,
                                    float(slope or 0.0), flags, _stream()), 'conv2d_fwd')
    return y


_POOL_EPI = os.environ.get('SRHIP_POOL_EPI', '1') == '1'      # A/B knob: 0 = the RAB tails run their own pooling pass over u


def pool_epilogue_ok(x, w):
    """RAB conv2 + its CLAM pooling partials in one call (srhip_conv2d_fwd_pool): split-bf16, stride-1 3x3 to 64 channels."""
    return (_POOL_EPI and x.is_cuda and get_conv_math() == 'bf16x3' and tuple(w.shape[2:]) == (3, 3) and w.shape[0] == 64
            and w.shape[1] % 32 == 0)


def conv2d_fwd_pool_raw(x, w, bias):
    """y = conv3x3(x, w) + bias (stride 1, pad 1, 64 output channels) and the CLAM pooling partials of y.
    Returns (y, (pool, section_bytes, nseg)): pool = [sum | max | arg] sections, nseg partial segments per image."""
    _require_gpu(x, 'conv2d_fwd_pool')
    x = nhwc(x)
    n, cin, h, wd = x.shape
    cout = w.shape[0]
    y = empty_nhwc(n, cout, h, wd, x)
    lib = _hip.lib()
    sec = n * lib.srhip_clam_pool_max_segments() * 64 * 4
    pool = torch.empty(3 * sec // 4, device=x.device, dtype=torch.float32)
    nseg = ctypes.c_int(0)
    b = bias.detach().contiguous() if bias is not None else None
    _hip.check(lib.srhip_conv2d_fwd_pool(_p(x), _p(packed_weight(w, 0)), _p(b), _p(y), _p(pool), sec, ctypes.byref(nseg), n, h, wd, cin,
                                         cout, cin, cout, EPI_BIAS if b is not None else 0, _stream()), 'conv2d_fwd_pool')
    return y, (pool, sec, nseg.value)


def conv2d_dgrad_raw(dy, w, x_shape, stride, pad, residual=None, actmask=None, slope=0.0, extra=()):
    """dx = conv_transpose(dy, w) [* lrelu'(actmask)] [+ residual] [+ extra[0] + extra[1]]: the optional tail fuses the backward
    of the LeakyReLU that produced this conv's input and the skip-path gradient add; `extra`: up to two more gradients of dx's
    shape (ops.carry_open), added after the residual in the order given."""
    _require_gpu(dy, 'conv2d_dgrad')
    dy = nhwc(dy)
    n, cin, h, wd = x_shape
    cout, _, kh, kw = w.shape
    dx = empty_nhwc(n, cin, h, wd, dy)
    if residual is not None:
        residual = nhwc(residual)
    if actmask is not None:
        actmask = nhwc(actmask)
    lib = _hip.lib()
    if extra:
        if residual is None or actmask is not None or len(extra) > 2:
            raise ValueError('conv2d_dgrad: extra residuals ride beside a residual, without an activation mask, at most two')
        extra = [nhwc(e) for e in extra]
        _hip.check(lib.srhip_conv2d_dgrad_res3(_p(dy), _p(packed_weight(w, 1)), _p(dx), _p(residual), _p(extra[0]),
                                               _p(extra[1]) if len(extra) > 1 else None, n, h, wd, cin, cout, kh, kw, stride, pad,
                                               _stream()), 'conv2d_dgrad_res3')
        return dx
    _hip.check(lib.srhip_conv2d_dgrad(_p(dy), _p(packed_weight(w, 1)), _p(dx), _p(residual), _p(actmask), float(slope),
                                      n, h, wd, cin, cout, kh, kw, stride, pad, cout, cin, cin, 0, _stream()),
               'conv2d_dgrad')
    return dx


_SIDE_WS = {}
_FORK_C = os.environ.get('SRHIP_FORK_C', '1') == '1'      # A/B knob: 0 = fork the side stream through torch events + a stream context


def _side_workspace(nbytes, device, stream):
    """One persistent split-K workspace per side stream: the kernels of one stream run in order, so they can share it, and a
    buffer that is never freed needs no record_stream (a per-call buffer allocated on the launching stream and used on the
    side stream would)."""
    key = (device.index, stream.cuda_stream)
    ws = _SIDE_WS.get(key)
    need = (max(int(nbytes), 4) + 3) // 4
    if ws is None or ws.numel() < need:
        if ws is not None:
            ws.record_stream(stream)                     # kernels already enqueued may still use the old buffer
        ws = _SIDE_WS[key] = torch.empty(max(need, 1 << 20), device=device, dtype=torch.float32)
    return ws


def conv2d_wgrad_raw(x, dy, w_shape, stride, pad, with_bias=False, xrowscale=None, xchanscale=None, out=None, on_stream=None):
    """(dw [OIHW], db [Cout] or None): the bias gradient comes out of the same kernel pass.
    out=(dw_buf, db_buf or None): accumulate into these buffers instead (dw_buf += dw, db_buf += db).
    on_stream: launch on this torch stream (already ordered behind the operands' producers by the caller) instead of the
    current one, with that stream's persistent workspace."""
    _require_gpu(x, 'conv2d_wgrad')
    x, dy = nhwc(x), nhwc(dy)
    n, cin, h, wd = x.shape
    cout, _, kh, kw = w_shape
    lib = _hip.lib()
    nbytes = lib.srhip_conv2d_wgrad_workspace(n, h, wd, cin, cout, kh, kw, stride, pad)
    if on_stream is not None:
        ws = _side_workspace(nbytes, x.device, on_stream)
        dw, db = out
        _hip.check(lib.srhip_conv2d_wgrad(_p(x), _p(dy), _p(dw), _p(db), _p(xrowscale), _p(xchanscale), 1, _p(ws),
                                          ws.numel() * 4, n, h, wd, cin, cout, kh, kw, stride, pad, cin, cout,
                                          ctypes.c_void_p(on_stream.cuda_stream)), 'conv2d_wgrad')
        return dw, db
    ws = torch.empty((max(nbytes, 4) + 3) // 4, device=x.device, dtype=torch.float32)
    if out is None:
        dw = torch.empty(tuple(w_shape), device=x.device, dtype=torch.float32)
        db = torch.empty(cout, device=x.device, dtype=torch.float32) if with_bias else None
        acc = 0
    else:
        dw, db = out
        acc = 1
    _hip.check(lib.srhip_conv2d_wgrad(_p(x), _p(dy), _p(dw), _p(db), _p(xrowscale), _p(xchanscale), acc, _p(ws),
                                      ws.numel() * 4, n, h, wd, cin, cout, kh, kw, stride, pad, cin, cout, _stream()),
               'conv2d_wgrad')
    return dw, db


def _fork_side(side, handles=None):
    """The side stream waits for everything enqueued so far on the current stream (and on the streams whose raw handles are
    given: the producers of operands that were queued for a grouped launch)."""
    hs = {_stream().value}
    if handles:
        hs.update(handles)
    hs.discard(side.cuda_stream)
    for h in hs:
        if _state.capturing or not _FORK_C:
            side.wait_stream(torch.cuda.ExternalStream(h) if h else torch.cuda.default_stream())
        else:
            _hip.check(_hip.lib().srhip_stream_fork(ctypes.c_void_p(h), ctypes.c_void_p(side.cuda_stream)), 'stream_fork')


def conv2d_wgrad_multi_raw(items, on_stream=None):
    """items: [(x, dy, dw_buf, db_buf or None, stride, pad)] -- 2..4 weight gradients of ONE shape, accumulated into their
    buffers by one grouped launch (srhip_conv2d_wgrad_multi) on `on_stream` (default: the current stream)."""
    x0, dy0, dw0, _, stride, pad = items[0]
    n, cin, h, wd = x0.shape
    cout, _, kh, kw = dw0.shape
    lib = _hip.lib()
    k = len(items)
    tab = ctypes.c_void_p * k
    xs = tab(*[it[0].data_ptr() for it in items])
    dys = tab(*[it[1].data_ptr() for it in items])
    dws = tab(*[it[2].data_ptr() for it in items])
    dbs = tab(*[(it[3].data_ptr() if it[3] is not None else None) for it in items])
    nbytes = lib.srhip_conv2d_wgrad_workspace(n, h, wd, cin, cout, kh, kw, stride, pad)
    if on_stream is not None:
        ws = _side_workspace(nbytes, x0.device, on_stream)
        st = ctypes.c_void_p(on_stream.cuda_stream)
    else:
        ws = torch.empty((max(nbytes, 4) + 3) // 4, device=x0.device, dtype=torch.float32)
        st = _stream()
    _hip.check(lib.srhip_conv2d_wgrad_multi(k, xs, dys, dws, dbs, 1, _p(ws), ws.numel() * 4, n, h, wd, cin, cout, kh, kw,
                                           stride, pad, cin, cout, st), 'conv2d_wgrad_multi')


# ---- padded split-bf16 planes (include/sradsgan_hip.h, ABI 9) ---------------------------------------------------------- #
class PP:
    """Padded planes of an NHWC tensor [n, c, h, w]: `buf` = bf16 [srhip_pp_plane_pixels(n, h, w), 2 c] (per 8 channels: 8 hi | 8 lo
    halves), pad / guard / tail rows zero.  The kernels never write those rows, so a buffer is zeroed ONCE (pp_empty) and can then be reused for any
    tensor of the same geometry."""
    __slots__ = ('buf', 'n', 'c', 'h', 'w')

    def __init__(self, buf, n, c, h, w):
        self.buf, self.n, self.c, self.h, self.w = buf, n, c, h, w

    @property
    def shape(self):
        return (self.n, self.c, self.h, self.w)

    def data_ptr(self):
        return self.buf.data_ptr()

    def record_stream(self, s):
        self.buf.record_stream(s)


def pp_empty(n, c, h, w, device):
    px = _hip.lib().srhip_pp_plane_pixels(n, h, w)
    return PP(torch.zeros(px, 2 * c, device=device, dtype=torch.bfloat16), n, c, h, w)


def pp_from_f32(x, out=None):
    """fp32 [n, c, h, w] (NHWC memory) -> padded planes (a stand-alone pass: tests, the bench's operand set-up)."""
    _require_gpu(x, 'pp_from_f32')
    x = nhwc(x)
    n, c, h, w = x.shape
    pp = out if out is not None else pp_empty(n, c, h, w, x.device)
    _hip.check(_hip.lib().srhip_pp_from_f32(_p(x), _p(pp.buf), n, h, w, c, c, _stream()), 'pp_from_f32')
    return pp


def pp_to_f32(pp):
    out = torch.empty(pp.n, pp.h, pp.w, pp.c, device=pp.buf.device, dtype=torch.float32).permute(0, 3, 1, 2)
    _hip.check(_hip.lib().srhip_pp_to_f32(_p(pp.buf), _p(out), pp.n, pp.h, pp.w, pp.c, pp.c, _stream()), 'pp_to_f32')
    return out


def conv2d_pp_ok(n, cin, h, w, cout):
    """The persistent patch kernel takes this 3x3 stride-1 pad-1 conv with padded-plane operands (split-bf16 mode)."""
    return bool(_hip.lib().srhip_conv2d_pp_ok(n, h, w, cin, cout))


def pp_sign_words(n, h, w, channels, device):
    """An (uninitialised) sign-word buffer for the LeakyReLU mask of an [n, channels, h, w] plane tensor (srhip_conv2d_pp_sign_bytes:
    1 bit per element, in the persistent patch kernel's own order), or None when the shape is not served."""
    nb = _hip.lib().srhip_conv2d_pp_sign_bytes(n, h, w, channels)
    return torch.empty(nb // 8, device=device, dtype=torch.int64) if nb else None


def conv2d_fwd_pp_raw(x, w, bias, slope=None, out_pp=None, pool=False, signs=None):
    """3x3 stride-1 pad-1 forward with padded-plane operands: x a PP or fp32 NHWC; out_pp = PP buffer to fill (then the result is that
    PP) or None (fp32 result).  pool=True (fp32 result of 64 channels): also the CLAM pooling partials, as conv2d_fwd_pool_raw.
    signs (pp_sign_words; bias + LeakyReLU onto planes): the launch also leaves the activation's sign words there."""
    cout, cin = w.shape[0], w.shape[1]
    n, _, h, wd = x.shape
    xpp = isinstance(x, PP)
    if not xpp:
        _require_gpu(x, 'conv2d_fwd_pp')
        x = nhwc(x)
    dev = x.buf.device if xpp else x.device
    if signs is not None:
        _hip.check(_hip.lib().srhip_conv2d_fwd_pp_signs(ctypes.c_void_p(x.data_ptr()), int(xpp), _p(packed_weight(w, 0)), _p(bias.detach().contiguous()),
                                                        ctypes.c_void_p(out_pp.data_ptr()), _p(signs), signs.numel() * 8, n, h, wd, cin, cout,
                                                        float(slope), _stream()), 'conv2d_fwd_pp_signs')
        return out_pp
    y = out_pp if out_pp is not None else torch.empty(n, h, wd, cout, device=dev, dtype=torch.float32).permute(0, 3, 1, 2)
    flags = 0
    b = None
    if bias is not None:
        flags |= EPI_BIAS
        b = bias.detach().contiguous()
    if slope is not None:
        flags |= EPI_LRELU
    lib = _hip.lib()
    pl, sec, nseg = None, 0, ctypes.c_int(0)
    if pool:
        sec = n * lib.srhip_clam_pool_max_segments() * 64 * 4
        pl = torch.empty(3 * sec // 4, device=dev, dtype=torch.float32)
    _hip.check(lib.srhip_conv2d_fwd_pp(ctypes.c_void_p(x.data_ptr()), int(xpp), _p(packed_weight(w, 0)), _p(b), ctypes.c_void_p(y.data_ptr()),
                                      int(out_pp is not None), _p(pl), sec, ctypes.byref(nseg), n, h, wd, cin, cout, float(slope or 0.0), flags,
                                      _stream()), 'conv2d_fwd_pp')
    return (y, (pl, sec, nseg.value)) if pool else y


def conv2d_dgrad_pp_raw(dy, w, residual=None, actmask=None, slope=0.0, out_pp=None, extra=(), signs=None):
    """3x3 stride-1 pad-1 data gradient with padded-plane operands: dy a PP or fp32 NHWC; out_pp = PP buffer to fill, with
    actmask = the PP of the LeakyReLU output that fed the forward conv, or signs = the sign words its producer left
    (conv2d_fwd_pp_raw(signs=)); or fp32 result (+ residual)."""
    cout, cin = w.shape[0], w.shape[1]
    n, _, h, wd = dy.shape
    ypp = isinstance(dy, PP)
    if not ypp:
        _require_gpu(dy, 'conv2d_dgrad_pp')
        dy = nhwc(dy)
    dev = dy.buf.device if ypp else dy.device
    if signs is not None:
        _hip.check(_hip.lib().srhip_conv2d_dgrad_pp_signs(ctypes.c_void_p(dy.data_ptr()), int(ypp), _p(packed_weight(w, 1)), ctypes.c_void_p(out_pp.data_ptr()),
                                                          _p(signs), signs.numel() * 8, float(slope), n, h, wd, cin, cout, _stream()), 'conv2d_dgrad_pp_signs')
        return out_pp
    dx = out_pp if out_pp is not None else torch.empty(n, h, wd, cin, device=dev, dtype=torch.float32).permute(0, 3, 1, 2)
    if residual is not None:
        residual = nhwc(residual)
    if extra:                                             # (ops.carry_open) up to two more gradients, added after the residual in this order
        if residual is None or actmask is not None or out_pp is not None or len(extra) > 2:
            raise ValueError('conv2d_dgrad_pp: extra residuals ride beside a residual of an fp32 destination, at most two')
        extra = [nhwc(e) for e in extra]
        _hip.check(_hip.lib().srhip_conv2d_dgrad_pp_res3(ctypes.c_void_p(dy.data_ptr()), int(ypp), _p(packed_weight(w, 1)), _p(dx), _p(residual),
                                                         _p(extra[0]), _p(extra[1]) if len(extra) > 1 else None, n, h, wd, cin, cout,
                                                         _stream()), 'conv2d_dgrad_pp_res3')
        return dx
    _hip.check(_hip.lib().srhip_conv2d_dgrad_pp(ctypes.c_void_p(dy.data_ptr()), int(ypp), _p(packed_weight(w, 1)), ctypes.c_void_p(dx.data_ptr()),
                                                int(out_pp is not None), _p(residual), ctypes.c_void_p(actmask.data_ptr()) if actmask is not None else None,
                                                float(slope), n, h, wd, cin, cout, _stream()), 'conv2d_dgrad_pp')
    return dx


def conv2d_wgrad_pp_raw(items, accumulate=True, on_stream=None):
    """items: [(x, dy, dw_buf, db_buf or None)] -- 1..4 weight gradients of ONE 3x3 stride-1 pad-1 shape by one launch of the flat
    kernels (srhip_conv2d_wgrad_pp): x and / or dy are PP objects (padded planes), an operand that is not is fp32 NHWC."""
    x0, dy0, dw0, _ = items[0]
    cout, cin = dw0.shape[0], dw0.shape[1]
    n, _, h, wd = x0.shape
    lib = _hip.lib()
    xpp, ypp = int(isinstance(x0, PP)), int(isinstance(dy0, PP))
    for it in items[1:]:
        if int(isinstance(it[0], PP)) != xpp or int(isinstance(it[1], PP)) != ypp:
            raise RuntimeError('conv2d_wgrad_pp: the convolutions of one launch must share their operand formats (planes / fp32)')
    mask = lib.srhip_conv2d_wgrad_pp_ok(n, h, wd, cin, cout)
    want = 4 if (xpp and ypp) else 1 if ypp else 2 if xpp else 0
    if not (mask & want):
        raise RuntimeError('conv2d_wgrad_pp: shape %s -> %d not served with these operand formats (served mask %d)' % (tuple(x0.shape), cout, mask))
    k = len(items)
    tab = ctypes.c_void_p * k
    xs = tab(*[it[0].data_ptr() for it in items])
    dys = tab(*[it[1].data_ptr() for it in items])
    dws = tab(*[it[2].data_ptr() for it in items])
    dbs = tab(*[(it[3].data_ptr() if it[3] is not None else None) for it in items])
    nbytes = lib.srhip_conv2d_wgrad_pp_workspace(k, xpp, ypp, n, h, wd, cin, cout)
    if on_stream is not None:
        ws = _side_workspace(nbytes, dw0.device, on_stream)
        st = ctypes.c_void_p(on_stream.cuda_stream)
    else:
        ws = torch.empty((max(nbytes, 4) + 3) // 4, device=dw0.device, dtype=torch.float32)
        st = _stream()
    _hip.check(lib.srhip_conv2d_wgrad_pp(k, xs, dys, xpp, ypp, dws, dbs, 1 if accumulate else 0, _p(ws), ws.numel() * 4, n, h, wd, cin, cout,
                                        cout if xpp else cin, st), 'conv2d_wgrad_pp')


class _PlanePool:
    """Padded-plane buffers by geometry.  The kernels never write a buffer's pad / guard rows, so a buffer is zeroed once, when it
    is created, and handed out again for any tensor of its geometry.  A released buffer carries events of the streams that may
    still be reading it; get() takes the OLDEST released buffer whose events have completed and, when none has, a NEW buffer rather
    than a wait -- a wait would put the main stream behind the weight-gradient stream's last launch (the overlap the step lives on)."""

    def __init__(self):
        self.free = {}              # key -> [(PP, [events])], oldest first
        self.created = 0

    def get(self, n, c, h, w, device):
        key = (n, c, h, w, device.index)
        q = self.free.get(key)
        capturing = _state.capturing or torch.cuda.is_current_stream_capturing()
        if capturing:
            # a stream capture: buffers the capture creates live in the graph's private pool and must not leave it, buffers from
            # outside must not be queried (event queries are illegal while capturing): keep the two populations apart.  Inside the
            # capture a released buffer is taken back at once and the TAKING stream waits for the streams that may still read it
            # (wait_stream = record + wait, the one cross-stream primitive the capture already uses everywhere: it becomes a graph
            # edge) -- without it the side stream's weight-gradient node and the main stream's next writer of the same buffer were
            # unordered in the captured graph.  The wait covers everything the releasing stream has enqueued so far, not
            # just the reader: the replayed graph loses some overlap there, eager launches (the default) are not affected.
            # (Events recorded at release time and waited for here would be tighter, but an event that is recorded inside a capture
            # and never waited for -- a buffer nobody takes again -- crashed hipStreamEndCapture on ROCm 7.0.)
            key = key + ('capture',)
            q = self.free.get(key)
        if q:
            for i, (pp, evs) in enumerate(q):
                if capturing:
                    del q[i]
                    cur = torch.cuda.current_stream()
                    for st in evs:                      # (under capture the entries are the releasing STREAMS)
                        if st != cur:
                            cur.wait_stream(st)
                    return pp
                if all(ev.query() for ev in evs):
                    del q[i]
                    return pp
        self.created += 1
        return pp_empty(n, c, h, w, device)

    def release_all(self):
        """Drops every free buffer (after the streams that may still read them have passed their release events): a NEW model is
        about to run -- another scale of the chain sweep, another network -- and the old geometries' buffers would otherwise stay for
        the life of the process (chain sweep in one process: 115 -> 147 GB, profiles/r05_bench_chain_bf16x3.json)."""
        for key, q in self.free.items():
            for pp, evs in q:
                for ev in evs:
                    ev.synchronize()                   # (an event, or -- entries released inside a capture -- the stream itself)
        self.free.clear()

    def begin_capture(self):
        """A new stream capture starts: buffers (and the events they carry) of an earlier capture belong to that graph."""
        for key in [k for k in self.free if k[-1] == 'capture']:
            del self.free[key]

    def put(self, pp, streams=()):
        capturing = _state.capturing or torch.cuda.is_current_stream_capturing()
        evs = list(streams) if capturing else [s.record_event() for s in streams]
        key = (pp.n, pp.c, pp.h, pp.w, pp.buf.device.index) + (('capture',) if capturing else ())
        self.free.setdefault(key, []).append((pp, evs))


plane_pool = _PlanePool()
if os.environ.get('SRHIP_PERS_GRID'):                          # experiment: blocks of the persistent patch kernel (default: 3 per CU / one per tile)
    _hip.lib().srhip_debug_set(5, int(os.environ['SRHIP_PERS_GRID']))
if os.environ.get('SRHIP_FLAT_BLOCKS'):                        # experiment: blocks of the 8-wave weight-gradient kernel (default: one per CU)
    _hip.lib().srhip_debug_set(12, int(os.environ['SRHIP_FLAT_BLOCKS']))
if os.environ.get('SRHIP_POOL_EPI_ANY'):                       # experiment: conv2's CLAM pooling epilogue at any launch size (B = 32: 768 tiles)
    _hip.lib().srhip_debug_set(19, int(os.environ['SRHIP_POOL_EPI_ANY']))
if os.environ.get('SRHIP_FLAT_F32_K8'):                       # experiment: 0 = a weight gradient with one fp32 operand takes the 4-wave flat kernel
    _hip.lib().srhip_debug_set(14, int(os.environ['SRHIP_FLAT_F32_K8']))
if os.environ.get('SRHIP_TAIL_DBG'):                           # A/B knob: bit 32 = the round-5 launch sequence of the tail's backward (7x7 data gradient as its own launch)
    _hip.lib().srhip_debug_set(7, int(os.environ['SRHIP_TAIL_DBG']))
_X_PP = os.environ.get('SRHIP_X_PP', '1') == '1'                # A/B knob: 0 = a RAB's input never arrives as planes (conversion pass for its weight gradient)
_DU_PP = os.environ.get('SRHIP_DU_PP', '1') == '1'              # A/B knob: 0 = conv2's gradients read the fp32 du (split in the dgrad kernel, pp_from_f32 pass for the weight gradient)
_PP_RAB = os.environ.get('SRHIP_PP_RAB', '1') == '1'          # A/B knob: 0 = the RAB keeps t / dt as fp32 tensors (rounds 1-4)
_PP_SIGNS = os.environ.get('SRHIP_PP_SIGNS', '1') == '1'      # A/B knob: 0 = conv2's data gradient reads its LeakyReLU mask from t's hi plane (48 MB) instead of sign words (3 MB)


def rab_planes_ok(x, w1, w2):
    """The RAB keeps its two 256-channel tensors as padded planes: split-bf16 arithmetic, 64 -> Cmid -> 64 with both 3x3 convs and
    both weight gradients served on planes."""
    if not (_PP_RAB and x.is_cuda and get_conv_math() == 'bf16x3'):
        return False
    n, c, h, w = x.shape
    cm = w1.shape[0]
    lib = _hip.lib()
    return (tuple(w1.shape[2:]) == (3, 3) and tuple(w2.shape[2:]) == (3, 3) and w2.shape[0] == c and w2.shape[1] == cm
            and bool(lib.srhip_conv2d_pp_ok(n, h, w, c, cm)) and bool(lib.srhip_conv2d_pp_ok(n, h, w, cm, c))
            and (lib.srhip_conv2d_wgrad_pp_ok(n, h, w, c, cm) & 4) and (lib.srhip_conv2d_wgrad_pp_ok(n, h, w, cm, c) & 4))


_WGRAD_PP_CONVERT = os.environ.get('SRHIP_WGRAD_PP_CONVERT', '1') == '1'   # 1 (default): a pp_from_f32 pass of the 64-channel operand on the weight-gradient stream (in-step +0.4 % over the kernel's own in-place split: profiles/r05_wgrad_flat.txt)


def _to_planes(t, device):
    """fp32 NHWC operand of a weight gradient -> a pooled padded-plane copy (launched on the current stream)."""
    n, c, h, w = t.shape
    pp = plane_pool.get(n, c, h, w, device)
    return pp_from_f32(t, out=pp)


def _launch_wgrad_pp(items, side):
    """items: [(x, dy, gw, gb, release)] of one shape: fp32 operands are converted to planes and the flat kernel runs, all on `side`
    (None: the current stream); pooled buffers go back to the pool with an event of that stream."""
    main = torch.cuda.current_stream()
    run_on = side if side is not None else main
    with torch.cuda.stream(run_on):
        conv = []
        launch = []
        for x, dy, gw, gb, release in items:
            xo, dyo = x, dy
            if _WGRAD_PP_CONVERT:                        # A/B knob: a stand-alone pp_from_f32 pass instead of the kernel's in-place split
                if not isinstance(x, PP):
                    xo = _to_planes(x, gw.device)
                    conv.append(xo)
                if not isinstance(dy, PP):
                    dyo = _to_planes(dy, gw.device)
                    conv.append(dyo)
            launch.append((xo, dyo, gw, gb))
        conv2d_wgrad_pp_raw(launch, accumulate=True, on_stream=side)
        for pp in conv:
            plane_pool.put(pp, (run_on,))
        for x, dy, gw, gb, release in items:
            for pp in release:
                plane_pool.put(pp, (run_on,) if side is None else (run_on, main))
    if side is not None:
        for x, dy, gw, gb, release in items:
            for t in (x, dy):
                if not isinstance(t, PP):
                    t.record_stream(side)
            if not isinstance(dy, PP):
                _hold_for_side(side, dy)


def wgrad_pp_for_params(w, b, x, dy, want_b, release=()):
    """Weight (+ bias) gradient of a 3x3 stride-1 pad-1 conv whose 256-channel operand is a PP (the other fp32 NHWC), accumulated into
    the parameters' gradient slots by the flat kernel -- in pairs on the weight-gradient stream like wgrad_for_params.  `release`:
    pooled PP buffers that go back to the pool once the launch is enqueued.  Returns False when direct accumulation is not possible
    (the caller converts and takes the fp32 path)."""
    gw = _grad_slot(w)
    gb = _grad_slot(b) if (want_b and b is not None) else None
    if gw is None or (want_b and b is not None and gb is None):
        return False
    side = _state.wgrad_stream
    _age_pending()
    item = (x, dy, gw, gb, tuple(release))
    if side is None:
        _launch_wgrad_pp([item], None)
        return True
    if _state.wgrad_group > 1:
        key = ('pp', tuple(x.shape), w.shape[0], gb is not None, isinstance(x, PP), isinstance(dy, PP))   # one launch = one operand format
        q = _state.pending.setdefault(key, [])
        q.append(item + (_stream().value, _state.wgrad_seq))
        if len(q) >= _PP_GROUP:
            if _WGRAD_SLOTS and _state.ready_pairs is not None and not _state.capturing:
                _state.ready_pairs.append(_state.pending.pop(key))      # complete: goes out at the next slot (release_ready_pair)
            else:
                _flush_key(key)
        return True
    _fork_side(side)
    _launch_wgrad_pp([item], side)
    return True


_WGRAD_DEFER = os.environ.get('SRHIP_WGRAD_DEFER', '0') == '1'     # experiment: hold every groupable weight gradient until a flush point
_WGRAD_FLUSH_GROUP = int(os.environ.get('SRHIP_WGRAD_FLUSH_GROUP', '0'))   # convolutions per launch at a flush (0: the step's group size)


def _flush_key(key):
    items = _state.pending.pop(key, None)
    if not items:
        return
    side = _state.wgrad_stream
    if key[0] == 'pp':
        _fork_side(side, [it[5] for it in items])
        _launch_wgrad_pp([it[:5] for it in items], side)
        return
    _fork_side(side, [it[6] for it in items])
    items = [it[:6] for it in items]
    per = max(1, _WGRAD_FLUSH_GROUP or _state.wgrad_group)
    for i0 in range(0, len(items), per):
        chunk = items[i0:i0 + per]
        if len(chunk) == 1:
            x, dy, gw, gb, stride, pad = chunk[0]
            if _state.capturing or not _FORK_C:
                with torch.cuda.stream(side):
                    conv2d_wgrad_raw(x, dy, tuple(gw.shape), stride, pad, gb is not None, out=(gw, gb))
            else:
                conv2d_wgrad_raw(x, dy, tuple(gw.shape), stride, pad, gb is not None, out=(gw, gb), on_stream=side)
        elif _state.capturing or not _FORK_C:
            with torch.cuda.stream(side):
                conv2d_wgrad_multi_raw(chunk)
        else:
            conv2d_wgrad_multi_raw(chunk, on_stream=side)
    for it in items:
        it[0].record_stream(side)
        it[1].record_stream(side)
        _hold_for_side(side, it[1])


_WGRAD_MAX_AGE = int(os.environ.get('SRHIP_WGRAD_MAX_AGE', '14'))    # weight-gradient calls a launch may wait for a partner (0: until the next flush point)


def _age_pending():
    """Called at every weight-gradient request: a pending launch whose partner has not shown up within _WGRAD_MAX_AGE further requests
    goes out alone.  The RAB pairs meet within 4-5 requests; shapes that occur ONCE per backward (the up-sampler's conv at 54^2 and at
    108^2) used to wait for the flush at the END of the generator's backward and then ran, alone on the chip, for 0.5 ms after the
    main stream had finished (`tools/step_tail.py`, profiles/r05_step_tail_before.txt)."""
    _state.wgrad_seq += 1
    if not _WGRAD_MAX_AGE or _WGRAD_DEFER or not _state.pending:
        return
    for key in [k for k, q in _state.pending.items() if q and _state.wgrad_seq - q[0][-1] > _WGRAD_MAX_AGE]:
        _flush_key(key)


# RAB weight gradients of one shape per flat-kernel launch.  Round 6: 3 -- a ResGroup's three RABs -- instead of pairs: the launch keeps one
# block per CU, so every convolution gets a third instead of half of the split-K splits (28 instead of 42): a third fewer partial tiles written and
# reduced, and the launches line up with the group boundaries the exchange's parts are cut at.  +0.5 % same-box (profiles/r06_step_ab.txt; 4: the same)
_PP_GROUP = max(2, min(4, int(os.environ.get('SRHIP_PP_GROUP', '3'))))
_WGRAD_SLOTS = os.environ.get('SRHIP_WGRAD_SLOTS', '0') == '1'     # experiment, OFF: see release_ready_pair (kernel-level effect as predicted, step unchanged)


def _launch_ready(items):
    side = _state.wgrad_stream
    _fork_side(side, [it[5] for it in items])
    _launch_wgrad_pp([it[:5] for it in items], side)


def release_ready_pair(n=1):
    """Round 6: WHEN a complete pair of RAB weight gradients starts.  The 8-wave flat kernel holds one block on every CU for ~150 us
    (2 x 170 of the 512 registers per SIMD lane) and leaves room for ONE block of a main-stream conv beside it.  The 128-wide conv
    kernels live with that (12 MFMAs per barrier and wave: 87 us in the step against 79 alone); the 64-wide one -- conv1's data gradient,
    256 -> 64, 6 MFMAs per barrier -- does not: 149 us in the step against 77 alone, 36 times per step on the main stream's chain
    (profiles/r06_step_eager_kernel_stats.txt).  And the round-5 request order put the two exactly on top of each other: conv2's
    weight gradient was requested between conv2's and conv1's data gradient, so a pair that completed there started the moment
    conv1's data gradient did.  Now complete pairs wait in _state.ready_pairs and ONE goes out right after every conv1 data gradient
    has been enqueued (_RabBlock._backward_planes): it runs beside the next block's attention tail (streaming passes) and conv2 data
    gradient (128-wide) -- ~210 us, room for one pair -- and is mostly done when the next conv1 data gradient starts.  Same kernels,
    same accumulation order per parameter: bit-identical.
    MEASURED (profiles/r06_step_ab.txt): conv1's data gradient 147.7 -> 91.6 us in the step, exactly as intended -- and the step does not
    move (668.8 / 672.3 / 671.1 against 669.5 / 672.1 / 671.0 img/s): the tail's 1x1 data gradient (39 -> 48 us), the split-K reduces
    (23 -> 31) and the 1x1 weight gradients (35 -> 40) take up what it gives back.  During the backward the chip's throughput is the limit,
    not any kernel's place in a queue.  Kept as an option (SRHIP_WGRAD_SLOTS=1), off by default."""
    q = _state.ready_pairs
    while q and n > 0:
        _launch_ready(q.pop(0))
        n -= 1


def flush_pending_pp():
    """The RAB weight gradients of the flat kernel that are still waiting for partners go out now (end of a ResGroup's backward)."""
    if _state.pending:
        for key in [k for k in _state.pending if k[0] == 'pp']:
            if _WGRAD_SLOTS and _state.ready_pairs is not None and not _state.capturing:
                _state.ready_pairs.append(_state.pending.pop(key))
            else:
                _flush_key(key)


def flush_pending_wgrads():
    """Launches every weight gradient that is still waiting for a partner of its shape (and every complete pair that is waiting for
    its slot).  Called wherever something is about to order itself behind "all weight gradients so far": the exchange, the joins of
    the step, the end of direct_param_grads()."""
    if _state.ready_pairs:
        release_ready_pair(len(_state.ready_pairs))
    if _state.pending:
        for key in list(_state.pending):
            _flush_key(key)


def wgrad_for_params(w, b, x, dy, stride, pad, want_b, xrowscale=None, xchanscale=None):
    """(dw, db) to return to autograd for parameters (w, b).  In direct_param_grads() mode the kernel
    accumulates into w.grad / b.grad and this returns (None, None)."""
    cout, cin, kh, kw = w.shape
    gw = _grad_slot(w)
    gb = _grad_slot(b) if (want_b and b is not None) else None
    if gw is not None and (not want_b or gb is not None) and \
            _hip.lib().srhip_conv2d_wgrad_can_accumulate(cin, cout, kh, kw):
        side = _state.wgrad_stream
        if side is None:
            conv2d_wgrad_raw(x, dy, tuple(w.shape), stride, pad, want_b, xrowscale, xchanscale, out=(gw, gb))
            return None, None
        _age_pending()
        # Weight gradients are off the critical path of backward (nothing reads them before the optimiser):
        # run them on a side stream so the partially filled last wave of each data-gradient kernel and of
        # each wgrad kernel overlap.  Same-parameter accumulations stay ordered (one side stream).
        x, dy = nhwc(x), nhwc(dy)
        if (_state.wgrad_group > 1 and xrowscale is None and xchanscale is None and (gb is not None or not want_b)
                and _hip.lib().srhip_conv2d_wgrad_multi_ok(x.shape[0], x.shape[2], x.shape[3], cin, cout, kh, kw, stride, pad)
                >= _state.wgrad_group):
            # nothing reads a weight gradient before the optimiser: wait for a partner of the same shape (the next RAB's) and
            # launch them together -- one full wave of blocks serves both with half the split-K partials each
            key = (tuple(x.shape), cout, stride, pad, gb is not None)
            q = _state.pending.setdefault(key, [])
            q.append((x, dy, gw, gb, stride, pad, _stream().value, _state.wgrad_seq))     # + the stream that produced the operands, + the request number
            if len(q) >= _state.wgrad_group and not _WGRAD_DEFER:
                _flush_key(key)
            return None, None
        if _state.capturing or not _FORK_C:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                conv2d_wgrad_raw(x, dy, tuple(w.shape), stride, pad, want_b, xrowscale, xchanscale, out=(gw, gb))
        else:
            # one C call forks the side stream behind the current one (event ring inside the library) and the kernel is
            # launched on it by handle: no torch Event, no stream context, no per-call workspace (~150 launches per step)
            _hip.check(_hip.lib().srhip_stream_fork(_stream(), ctypes.c_void_p(side.cuda_stream)), 'stream_fork')
            conv2d_wgrad_raw(x, dy, tuple(w.shape), stride, pad, want_b, xrowscale, xchanscale, out=(gw, gb), on_stream=side)
        for t in (x, dy, xrowscale, xchanscale):
            if t is not None:
                t.record_stream(side)
        _hold_for_side(side, dy)
        return None, None
    if gw is not None and (not want_b or gb is not None):
        # Shapes the accumulating kernel does not take (the 3- and 2-channel convs, the 64 -> 3 tail conv): add in program order
        # here rather than through autograd's AccumulateGrad, whose order among several contributions to one parameter
        # follows per-thread node counters (tests/test_graph_gpu.py) -- and, like the other weight gradients, on the side
        # stream: the 3 -> 64 head convs' gradients (0.2 - 0.3 ms each at 216 x 216) sat in the discriminator's serial chain
        side = _state.wgrad_stream
        if side is None:
            dw, db = conv2d_wgrad_raw(x, dy, tuple(w.shape), stride, pad, want_b, xrowscale, xchanscale)
            gw.add_(dw)
            if db is not None:
                gb.add_(db)
            return None, None
        x, dy = nhwc(x), nhwc(dy)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            dw, db = conv2d_wgrad_raw(x, dy, tuple(w.shape), stride, pad, want_b, xrowscale, xchanscale)
            gw.add_(dw)
            if db is not None:
                gb.add_(db)
        for t in (x, dy, xrowscale, xchanscale):
            if t is not None:
                t.record_stream(side)
        _hold_for_side(side, dy)
        return None, None
    return conv2d_wgrad_raw(x, dy, tuple(w.shape), stride, pad, want_b, xrowscale, xchanscale)


_WGRAD_ACT = os.environ.get('SRHIP_WGRAD_ACT', '1') == '1'      # A/B knob


def _wgrad_act_direct(w, b, x, dy, y, slope, stride, pad):
    """Weight + bias gradient of a fused conv + LeakyReLU from the gradient at the activated output, accumulated into the
    parameters' arena slots on the weight-gradient stream (srhip_conv2d_wgrad_act).  False when the shape is not served."""
    cout, cin, kh, kw = w.shape
    gw, gb = _grad_slot(w), _grad_slot(b)
    if gw is None or gb is None:
        return False
    n, _, h, wd = x.shape
    lib = _hip.lib()
    if not lib.srhip_conv2d_wgrad_act_ok(n, h, wd, cin, cout, kh, kw, stride, pad):
        return False
    x, dy, y = nhwc(x), nhwc(dy), nhwc(y)
    side = _state.wgrad_stream
    cur = torch.cuda.current_stream()
    run_on = side if side is not None else cur
    if side is not None:
        side.wait_stream(cur)
    with torch.cuda.stream(run_on):
        nbytes = lib.srhip_conv2d_wgrad_workspace(n, h, wd, cin, cout, kh, kw, stride, pad)
        ws = torch.empty((max(nbytes, 4) + 3) // 4, device=x.device, dtype=torch.float32)
        dw = torch.empty(tuple(w.shape), device=x.device, dtype=torch.float32)
        db = torch.empty(cout, device=x.device, dtype=torch.float32)
        _hip.check(lib.srhip_conv2d_wgrad_act(_p(x), _p(dy), _p(y), float(slope), _p(dw), _p(db), _p(ws), ws.numel() * 4,
                                              n, h, wd, cin, cout, kh, kw, stride, pad, cin, cout, _stream()), 'conv2d_wgrad_act')
        gw.add_(dw)
        gb.add_(db)
    if side is not None:
        for t in (x, dy, y):
            t.record_stream(side)
        _hold_for_side(side, dy)
    return True


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


_LRELU_BITS = os.environ.get('SRHIP_LRELU_BITS', '1') == '1'    # A/B knob: 0 = every application of a LeakyReLU backward reads y


def lrelu_bwd_bits_raw(dy, y, mask, slope):
    """dx = dy * LeakyReLU'(y) through srhip_lrelu_bwd_bits: y given -> reads y and WRITES its sign bits into `mask`; y None -> reads
    `mask` instead of y."""
    _require_gpu(dy, 'lrelu_bwd_bits')
    dy = nhwc(dy)
    dx = torch.empty_like(dy, memory_format=CL)
    _hip.check(_hip.lib().srhip_lrelu_bwd_bits(_p(dy), _p(nhwc(y)) if y is not None else None, _p(mask), _p(dx), dy.numel(), float(slope),
                                               _stream()), 'lrelu_bwd_bits')
    return dx


class _LReluBwd(Function):
    """g = dy * (y > 0 ? 1 : slope); linear in dy, the mask is a constant.  Applied under a recorded graph (create_graph: the gradient
    penalty's first-order pass) to a large activation it also leaves the SIGN BITS of y behind (1 / 32 of y's bytes), and its own
    backward -- the same mask on another tensor -- reads those instead of y (`mask` travels as a non-differentiable tensor argument)."""

    @staticmethod
    def forward(ctx, dy, y, slope, mask=None):
        ctx.slope = slope
        if mask is not None:                           # the double backward of an earlier application: y's sign bits instead of y
            ctx.save_for_backward(y, mask)
            return lrelu_bwd_bits_raw(dy, None, mask, slope)
        if (_LRELU_BITS and ctx.needs_input_grad[0] and dy.is_cuda and dy.numel() % 4 == 0
                and dy.numel() >= (1 << 22)):          # a graph is being recorded over this application: its backward will want the mask again
            nbytes = _hip.lib().srhip_lrelu_mask_bytes(dy.numel())
            mask = torch.empty((nbytes + 7) // 8, device=dy.device, dtype=torch.int64)
            ctx.save_for_backward(y, mask)
            return lrelu_bwd_bits_raw(dy, y, mask, slope)
        ctx.save_for_backward(y, None)
        return lrelu_bwd_raw(dy, y, slope)

    @staticmethod
    def backward(ctx, gg):
        y, mask = ctx.saved_tensors
        return _LReluBwd.apply(gg, y, ctx.slope, mask), None, None, None


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
    def forward(ctx, x, w, b, residual, stride, pad, slope, graddata=False):
        if slope is not None and residual is not None:
            raise ValueError('conv2d: fused activation and residual are mutually exclusive')
        y = conv2d_fwd_raw(x, w, b, stride, pad, slope, residual, graddata=graddata)
        ctx.stride, ctx.pad, ctx.slope = stride, pad, slope
        ctx.has_bias, ctx.has_res = b is not None, residual is not None
        ctx.save_for_backward(x, w, y if slope is not None else None, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, b = ctx.saved_tensors
        skip = _skip_param_grads(w)
        need_dx = ctx.needs_input_grad[0] and x.data_ptr() not in _state.stop_ids
        if ctx.slope is not None and not need_dx and not skip and ctx.needs_input_grad[1] and ctx.has_bias and \
                ctx.needs_input_grad[2] and _state.direct_grads and not torch.is_grad_enabled() and _WGRAD_ACT and \
                _wgrad_act_direct(w, b, x, dy, y, ctx.slope, ctx.stride, ctx.pad):
            # head conv of the discriminator in the D step: only its parameter gradients are wanted, and the kernel applies the
            # activation's backward while it reads dy -- no lrelu-backward pass over the 382 MB gradient in the serial chain
            return None, None, None, None, None, None, None, None
        g = dy if ctx.slope is None else _LReluBwd.apply(dy, y, ctx.slope)
        dx = _ConvDgrad.apply(g, w, tuple(x.shape), ctx.stride, ctx.pad) if need_dx else None
        dw = db = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and not skip
        if ctx.needs_input_grad[1] and not skip and _state.direct_grads and not torch.is_grad_enabled():
            dw, db = wgrad_for_params(w, b, x, g, ctx.stride, ctx.pad, want_b)
        elif ctx.needs_input_grad[1] and not skip:
            dw, db = _ConvWgrad.apply(x, g, tuple(w.shape), ctx.stride, ctx.pad, want_b)
        elif want_b:
            db = _ColSum.apply(g)
        dres = _passed_through(dy) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres, None, None, None, None


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
        skip = _skip_param_grads(w)
        d_dy = _ConvFwd.apply(ddx, w, None, None, ctx.stride, ctx.pad, None, True) if ctx.needs_input_grad[0] else None
        d_w = None
        if ctx.needs_input_grad[1] and not skip and _state.direct_grads and not torch.is_grad_enabled():
            d_w, _ = wgrad_for_params(w, None, ddx, dy, ctx.stride, ctx.pad, False)   # same stream/order as every other wgrad of w
        elif ctx.needs_input_grad[1] and not skip:
            d_w, _ = _ConvWgrad.apply(ddx, dy, tuple(w.shape), ctx.stride, ctx.pad, False)
        return d_dy, d_w, None, None, None


class _ConvWgrad(Function):
    """(dw, db) = wgrad(x, dy): dw bilinear in (x, dy), db = sum of dy over pixels."""

    @staticmethod
    def forward(ctx, x, dy, w_shape, stride, pad, with_bias):
        ctx.stride, ctx.pad, ctx.w_shape = stride, pad, w_shape
        ctx.save_for_backward(x, dy)
        dw, db = conv2d_wgrad_raw(x, dy, w_shape, stride, pad, with_bias)
        if db is None:
            ctx.mark_non_differentiable()
        return dw, db

    @staticmethod
    def backward(ctx, ddw, ddb):
        x, dy = ctx.saved_tensors
        d_x = d_dy = None
        if ddw is not None:
            ddw = ddw.contiguous()
            if ctx.needs_input_grad[0]:
                d_x = _ConvDgrad.apply(dy, ddw, tuple(x.shape), ctx.stride, ctx.pad)
            if ctx.needs_input_grad[1]:
                d_dy = _ConvFwd.apply(x, ddw, None, None, ctx.stride, ctx.pad, None, True)
        if ddb is not None and ctx.needs_input_grad[1]:
            e = ddb.view(1, -1, 1, 1).expand(dy.shape)
            d_dy = e if d_dy is None else d_dy + e
        return d_x, d_dy, None, None, None, None


def conv2d(x, weight, bias=None, stride=1, padding=0, act_slope=None, residual=None):
    """nn.Conv2d forward with the elementwise tail of its call site fused (bias, LeakyReLU/ReLU,
    residual add).  x: logical NCHW; returns logical NCHW in NHWC memory."""
    return _ConvFwd.apply(x, weight, bias, residual, stride, padding, act_slope)


_BUS_SUM = os.environ.get('SRHIP_BUS_SUM', '1') == '1'      # A/B knob: 0 = chained torch adds


class _SumN(Function):
    """((t0 + t1) + t2) + ... in one pass (srhip_sum_n); every term gets the incoming gradient, as with chained adds."""

    @staticmethod
    def forward(ctx, tokens, *ts):
        _require_gpu(ts[0], 'sum_n')
        ts = [nhwc(t) for t in ts]
        out = torch.empty_like(ts[0], memory_format=CL)
        tab = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        _hip.check(_hip.lib().srhip_sum_n(tab, len(ts), _p(out), out.numel(), _stream()), 'sum_n')
        ctx.n, ctx.tokens = len(ts), tokens
        return out

    @staticmethod
    def backward(ctx, g):
        # a term that is the tagged input of a fused RAB (carry_open) gets no gradient edge: the RAB's data gradient adds g itself
        return (None,) + tuple(None if _carry_stash(tok, g) else g for tok in ctx.tokens)


def sum_tensors(ts):
    """Sum of 2..16 same-shape NCHW tensors in the order given (the generator's bus); falls back to chained adds otherwise."""
    ts = list(ts)
    if (_BUS_SUM and 2 <= len(ts) <= 16 and ts[0].is_cuda and ts[0].dim() == 4 and ts[0].numel() % 4 == 0
            and all(t.shape == ts[0].shape and t.dtype == torch.float32 for t in ts)):
        return _SumN.apply(tuple(_carry_commit(t) for t in ts), *ts)
    out = ts[0]
    for t in ts[1:]:
        out = out + t
    return out


class _CatChannels(Function):
    """torch.cat(dim = 1) of NHWC tensors in one pass (srhip_cat_channels); the backward splits the gradient back into dense tensors
    (srhip_split_channels).  Round 6: ATen's channels-last cat of the multi-scale block's three branches (sradsgan.py:340-344) took
    169 us for 72 MB on the one stream the generator's forward runs on."""

    @staticmethod
    def forward(ctx, *ts):
        _require_gpu(ts[0], 'cat_channels')
        ts = [nhwc(t) for t in ts]
        n, _, h, w = ts[0].shape
        chans = [t.shape[1] for t in ts]
        out = empty_nhwc(n, sum(chans), h, w, ts[0])
        tab = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        ctab = (ctypes.c_int * len(ts))(*chans)
        _hip.check(_hip.lib().srhip_cat_channels(tab, ctab, len(ts), _p(out), n * h * w, _stream()), 'cat_channels')
        ctx.chans, ctx.nhw = chans, (n, h, w)
        return out

    @staticmethod
    def backward(ctx, g):
        g = nhwc(g)
        n, h, w = ctx.nhw
        outs = [empty_nhwc(n, c, h, w, g) for c in ctx.chans]
        tab = (ctypes.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
        ctab = (ctypes.c_int * len(outs))(*ctx.chans)
        _hip.check(_hip.lib().srhip_split_channels(_p(g), ctab, len(outs), tab, n * h * w, _stream()), 'split_channels')
        return tuple(outs)


_CAT = os.environ.get('SRHIP_CAT', '1') == '1'     # A/B knob: 0 = torch.cat


def cat_channels(ts):
    """torch.cat(ts, dim=1) for 2..8 same-size fp32 NCHW tensors whose channel counts are multiples of 4; torch.cat otherwise."""
    ts = list(ts)
    if (_CAT and 2 <= len(ts) <= 8 and ts[0].is_cuda and all(t.dim() == 4 and t.dtype == torch.float32 and t.shape[1] % 4 == 0
                                                                and t.shape[0] == ts[0].shape[0] and t.shape[2:] == ts[0].shape[2:] for t in ts)):
        return _CatChannels.apply(*ts)
    return torch.cat(ts, dim=1)


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
# fused local-attention tail of RAB / ResGroup: CLAM -> SLAM -> conv1x1 (+bias) -> += skip
# (sradsgan.py:254-274, 303-323).  y = s*u and z = m*y are never written to HBM.
# --------------------------------------------------------------------------------------------- #


_TAIL_FUSED = os.environ.get('SRHIP_TAIL_FUSED', '1') == '1'


def _tail_forward(u, skip, fc1_w, fc2_w, w7, wc, bc, pool=None, out_pp=None):
    """returns (out, tensors to save for _tail_backward).  pool: (buffer, section bytes, nseg) from conv2d_fwd_pool_raw -- the
    pooling partials of u left behind by the conv that produced it (else the tail runs its own pooling pass)."""
    n, c, h, w = u.shape
    lib = _hip.lib()
    dev = u.device
    f32 = dict(device=dev, dtype=torch.float32)
    avg, mx, s = torch.empty(n, c, **f32), torch.empty(n, c, **f32), torch.empty(n, c, **f32)
    arg = torch.empty(n, c, device=dev, dtype=torch.int32)
    pooled, m = torch.empty(n * h * w, 2, **f32), torch.empty(n * h * w, **f32)
    argc = torch.empty(n * h * w, device=dev, dtype=torch.int32)
    fc1c, fc2c, w7c = fc1_w.detach().contiguous(), fc2_w.detach().contiguous(), w7.detach().contiguous()
    if pool is not None:
        _hip.check(lib.srhip_attn_tail_fwd_pooled(_p(u), _p(pool[0]), pool[1], pool[2], _p(fc1c), _p(fc2c), _p(w7c), _p(avg), _p(mx), _p(arg),
                                                  _p(s), _p(pooled), _p(argc), _p(m), n, h, w, c, fc1_w.shape[0], _stream()),
                   'attn_tail_fwd_pooled')
    else:
        ws = torch.empty(lib.srhip_attn_tail_workspace(n) // 4, **f32)
        _hip.check(lib.srhip_attn_tail_fwd(_p(u), _p(fc1c), _p(fc2c), _p(w7c), _p(avg), _p(mx), _p(arg), _p(s),
                                           _p(pooled), _p(argc), _p(m), _p(ws), ws.numel() * 4, n, h, w, c,
                                           fc1_w.shape[0], _stream()), 'attn_tail_fwd')
    out = conv2d_fwd_raw(u, wc, bc, 1, 0, None, skip, m, s, out_pp=out_pp)      # out_pp: the block output also as padded planes
    return out, (avg, mx, arg, s, pooled, argc, m)


def _tail_backward(g, u, fc1_w, fc2_w, w7, wc, bc, saved, has_bias, skip_params=False, du_pp=None):
    """g: gradient at the tail's output (NHWC).  Returns (du, dfc1, dfc2, dw7, dwc, dbc).  du_pp: a PP buffer that receives du as
    padded planes INSTEAD of the fp32 tensor (du is then returned as None; fused path only: the caller checks `_TAIL_FUSED`)."""
    avg, mx, arg, s, pooled, argc, m = saved
    n, c, h, w = u.shape
    lib = _hip.lib()
    f32 = dict(device=u.device, dtype=torch.float32)
    dz = conv2d_dgrad_raw(g, wc, tuple(u.shape), 1, 0)            # gradient at z = m*s*u
    dwc = dbc = None
    if not skip_params:
        dwc, dbc = wgrad_for_params(wc, bc, u, g, 1, 0, has_bias, m, s)             # x operand = z, rebuilt on the fly
    du = torch.empty_like(u, memory_format=CL)
    hid = fc1_w.shape[0]
    g7 = None if skip_params else _grad_slot(w7)
    g1, g2 = (None, None) if skip_params else (_grad_slot(fc1_w), _grad_slot(fc2_w))
    direct = g1 is not None and g2 is not None
    dw7 = g7 if g7 is not None else torch.empty(w7.shape, **f32)
    dfc1 = g1 if direct else torch.empty(fc1_w.shape, **f32)
    dfc2 = g2 if direct else torch.empty(fc2_w.shape, **f32)
    if not _TAIL_FUSED:                                        # A/B knob: the three separate entry points (10 launches)
        ds = torch.empty(n, c, **f32)
        ws = torch.empty(lib.srhip_attn_tail_bwd_workspace(n, h, w) // 4, **f32)
        _hip.check(lib.srhip_attn_tail_bwd_spatial(_p(dz), _p(u), _p(s), _p(m), _p(pooled), _p(argc), _p(w7.detach().contiguous()),
                                                   _p(du), _p(ds), _p(dw7), int(g7 is not None), _p(ws), ws.numel() * 4, n, h, w, c,
                                                   _stream()), 'attn_tail_bwd_spatial')
        davg, dmax = torch.empty(n, c, **f32), torch.empty(n, c, **f32)
        ws2 = torch.empty(lib.srhip_attn_tail_mlp_workspace(n, hid) // 4, **f32)
        _hip.check(lib.srhip_attn_tail_bwd_mlp(_p(ds), _p(avg), _p(mx), _p(s), _p(fc1_w.detach().contiguous()),
                                               _p(fc2_w.detach().contiguous()), _p(davg), _p(dmax), _p(dfc1), _p(dfc2), int(direct),
                                               _p(ws2), ws2.numel() * 4, n, c, hid, _stream()), 'attn_tail_bwd_mlp')
        _hip.check(lib.srhip_attn_tail_bwd_channel(_p(du), _p(davg), _p(dmax), _p(arg), n, h, w, c, _stream()), 'attn_tail_bwd_channel')
        return du, (None if direct else dfc1), (None if direct else dfc2), (None if g7 is not None else dw7), dwc, dbc
    # spatial half (7x7 conv, per-pixel gate), channel half (sigmoid -> shared MLP) and the arg-max fix-up: one call
    ws = torch.empty(lib.srhip_attn_tail_bwd_fused_workspace(n, h, w, hid) // 4, **f32)
    _hip.check(lib.srhip_attn_tail_bwd_pp(_p(dz), _p(u), _p(s), _p(m), _p(pooled), _p(argc), _p(avg), _p(mx), _p(arg),
                                              _p(w7.detach().contiguous()), _p(fc1_w.detach().contiguous()),
                                              _p(fc2_w.detach().contiguous()), _p(du), _p(du_pp.buf) if du_pp is not None else None, _p(dw7),
                                              int(g7 is not None), _p(dfc1), _p(dfc2),
                                              int(direct), _p(ws), ws.numel() * 4, n, h, w, c, hid, _stream()), 'attn_tail_bwd')
    if g7 is not None:
        dw7 = None
    if direct:
        dfc1 = dfc2 = None
    if du_pp is not None:
        du = None                  # the fp32 tensor only holds the main pass's partial result: the planes are the gradient
    return du, dfc1, dfc2, dw7, dwc, dbc


_TAIL_EVAL = os.environ.get('SRHIP_TAIL_EVAL', '1') == '1'      # A/B knob: 0 = the training-mode launches in inference as well


def _tail_eval_ok(u):
    """Inference fast path (srhip_attn_tail_eval): grad mode off, split-bf16 arithmetic, a CUDA tensor."""
    return _TAIL_EVAL and not torch.is_grad_enabled() and u.is_cuda and get_conv_math() == 'bf16x3'


def _tail_forward_eval(u, skip, fc1_w, fc2_w, w7, wc, bc, pool=None):
    """conv1x1(SLAM(CLAM(u))) + bc + skip with nothing saved: two launches (sradsgan.py:254-274 in eval mode), one when the conv
    that produced u left the pooling partials behind (pool, see _tail_forward)."""
    u, skip = nhwc(u), nhwc(skip)
    n, c, h, w = u.shape
    lib = _hip.lib()
    out = torch.empty_like(u, memory_format=CL)
    if pool is not None:
        _hip.check(lib.srhip_attn_tail_eval_pooled(_p(u), _p(skip), _p(pool[0]), pool[1], pool[2], _p(fc1_w.detach().contiguous()),
                                                   _p(fc2_w.detach().contiguous()), _p(w7.detach().contiguous()), _p(packed_weight(wc, 0)),
                                                   _p(bc), _p(out), n, h, w, c, fc1_w.shape[0], _stream()), 'attn_tail_eval_pooled')
        return out
    ws = torch.empty(lib.srhip_attn_tail_workspace(n) // 4, device=u.device, dtype=torch.float32)
    _hip.check(lib.srhip_attn_tail_eval(_p(u), _p(skip), _p(fc1_w.detach().contiguous()), _p(fc2_w.detach().contiguous()),
                                        _p(w7.detach().contiguous()), _p(packed_weight(wc, 0)), _p(bc), _p(out), _p(ws), ws.numel() * 4,
                                        n, h, w, c, fc1_w.shape[0], _stream()), 'attn_tail_eval')
    return out


class _AttentionTail(Function):
    @staticmethod
    def forward(ctx, u, skip, fc1_w, fc2_w, w7, wc, bc, carry=None, emit_pp=False):
        _require_gpu(u, 'attention_tail')
        u, skip = nhwc(u), nhwc(skip)
        _state.last_out_pp = None
        out_pp = None
        if emit_pp:                                    # the output also as padded planes for the RAB that consumes it (attention_tail)
            n, c, h, wd = u.shape
            out_pp = plane_pool.get(n, c, h, wd, u.device)
        out, saved = _tail_forward(u, skip, fc1_w, fc2_w, w7, wc, bc, out_pp=out_pp)
        _state.last_out_pp = out_pp
        ctx.save_for_backward(u, fc1_w, fc2_w, w7, wc, bc, *saved)
        ctx.has_bias = bc is not None
        ctx.carry = carry              # token of a skip tensor whose gradient is stashed for the group's first RAB (carry_open)
        return out

    @staticmethod
    def backward(ctx, g):
        u, fc1_w, fc2_w, w7, wc, bc, *saved = ctx.saved_tensors
        g = nhwc(g)
        du, dfc1, dfc2, dw7, dwc, dbc = _tail_backward(g, u, fc1_w, fc2_w, w7, wc, bc, saved, ctx.has_bias,
                                                       _skip_param_grads())
        dskip = None if _carry_stash(ctx.carry, g) else _passed_through(g)
        return du, dskip, dfc1, dfc2, dw7, dwc, dbc, None, None


class _RabBlock(Function):
    """Whole residual attention block (sradsgan.py:250-275) as ONE autograd node:
         t = LeakyReLU_0.2(conv1(x)) ; u = conv2(t) ; out = conv1x1(SLAM(CLAM(u))) + x
    Backward chains the fused kernels directly: conv2's dgrad applies conv1's activation mask in its
    epilogue, conv1's dgrad adds the skip gradient in its epilogue -- no lrelu-backward pass, no
    gradient-accumulation adds, 3 saved activations (x, t, u) instead of ~12."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, fc1_w, fc2_w, w7, wc, bc, x_pp=None, emit_pp=False, carry=None, group_first=False):
        _require_gpu(x, 'rab_block')
        x = nhwc(x)
        ctx.carry = carry              # token under which the input's other consumers stash their gradients (carry_open)
        ctx.t_pp = ctx.x_pp = ctx.signs = None
        ctx.planes = False
        ctx.group_first = bool(group_first)   # the first block of a ResGroup: its backward closes the group's weight-gradient launch
        _state.last_out_pp = None
        if rab_planes_ok(x, w1, w2):
            # round 5: t stays in padded split-bf16 planes between the block's own kernels (conv1's epilogue writes them, conv2 reads
            # them without its in-place split, the backward's activation mask and weight gradient read them again).  x_pp: the block
            # input ALSO as planes, left by the previous block's tail (emit_pp): conv1 and its weight gradient read them
            n, _, h, wd = x.shape
            t_pp = plane_pool.get(n, w1.shape[0], h, wd, x.device)
            # conv1's epilogue also leaves the signs of t (1 bit per element): the mask conv2's data gradient applies
            ctx.signs = pp_sign_words(n, h, wd, w1.shape[0], x.device) if (_PP_SIGNS and b1 is not None and any(ctx.needs_input_grad)) else None
            conv2d_fwd_pp_raw(x_pp if x_pp is not None else x, w1, b1, 0.2, out_pp=t_pp, signs=ctx.signs)
            if pool_epilogue_ok(x, w2):
                u, pool = conv2d_fwd_pp_raw(t_pp, w2, b2, pool=True)
            else:
                u, pool = conv2d_fwd_pp_raw(t_pp, w2, b2), None
            out_pp = plane_pool.get(n, x.shape[1], h, wd, x.device) if emit_pp else None
            out, saved = _tail_forward(u, x, fc1_w, fc2_w, w7, wc, bc, pool, out_pp=out_pp)
            _state.last_out_pp = out_pp
            ctx.t_pp, ctx.x_pp = t_pp, x_pp
            ctx.planes = True
            ctx.save_for_backward(x, u, w1, b1, w2, b2, fc1_w, fc2_w, w7, wc, bc, *saved)
            ctx.has_b = (b1 is not None, b2 is not None, bc is not None)
            return out
        t = conv2d_fwd_raw(x, w1, b1, 1, 1, 0.2)
        if pool_epilogue_ok(t, w2):                      # conv2 leaves the CLAM pooling partials of u behind (its epilogue, when the patch walk takes it)
            u, pool = conv2d_fwd_pool_raw(t, w2, b2)
        else:
            u, pool = conv2d_fwd_raw(t, w2, b2, 1, 1), None
        out, saved = _tail_forward(u, x, fc1_w, fc2_w, w7, wc, bc, pool)
        ctx.save_for_backward(x, t, u, w1, b1, w2, b2, fc1_w, fc2_w, w7, wc, bc, *saved)
        ctx.has_b = (b1 is not None, b2 is not None, bc is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.t_pp is not None:
            return _RabBlock._backward_planes(ctx, g)
        if getattr(ctx, 'planes', False):
            # the pooled plane buffers (t, x, sign words) went back to the pool with the first backward: there is nothing to run a
            # second one on (retain_graph) -- say so instead of unpacking saved_tensors in the fp32 path's order
            raise RuntimeError('_RabBlock: backward called twice on the padded-plane path (its plane buffers are released by the first '
                               'backward); run the forward again, or set SRHIP_PP_RAB=0 for a graph that is walked repeatedly')
        x, t, u, w1, b1, w2, b2, fc1_w, fc2_w, w7, wc, bc, *saved = ctx.saved_tensors
        g = nhwc(g)
        skip = _skip_param_grads()
        du, dfc1, dfc2, dw7, dwc, dbc = _tail_backward(g, u, fc1_w, fc2_w, w7, wc, bc, saved, ctx.has_b[2], skip)
        dt = conv2d_dgrad_raw(du, w2, tuple(t.shape), 1, 1, None, t, 0.2)          # * LeakyReLU'(t)
        dw2 = db2 = dw1 = db1 = None
        if not skip:
            dw2, db2 = wgrad_for_params(w2, b2, t, du, 1, 1, ctx.has_b[1])
        dx = conv2d_dgrad_raw(dt, w1, tuple(x.shape), 1, 1, g, extra=_carry_take(ctx.carry)) if ctx.needs_input_grad[0] else None   # + skip gradient (+ the input's stashed gradients)
        if not skip:
            dw1, db1 = wgrad_for_params(w1, b1, x, dt, 1, 1, ctx.has_b[0])
        return dx, dw1, db1, dw2, db2, dfc1, dfc2, dw7, dwc, dbc, None, None, None, None

    @staticmethod
    def _backward_planes(ctx, g):
        """The backward with t and dt as padded planes: conv2's dgrad reads the mask from t's hi plane and writes dt as planes, conv1's
        dgrad reads them without its split, both weight gradients run on the flat 8-wave kernel (the 64-channel operands x / du are
        converted on the weight-gradient stream)."""
        x, u, w1, b1, w2, b2, fc1_w, fc2_w, w7, wc, bc, *saved = ctx.saved_tensors
        t_pp, ctx.t_pp = ctx.t_pp, None
        x_pp, ctx.x_pp = ctx.x_pp, None
        g = nhwc(g)
        skip = _skip_param_grads()
        n, _, h, wd = x.shape
        # the tail's backward leaves du as fp32 (the gradient of the block's residual stream) AND as padded planes: conv2's data and
        # weight gradient read the planes (no in-kernel split, no conversion pass)
        du_pp = plane_pool.get(n, u.shape[1], h, wd, x.device) if (_TAIL_FUSED and _DU_PP) else None
        du, dfc1, dfc2, dw7, dwc, dbc = _tail_backward(g, u, fc1_w, fc2_w, w7, wc, bc, saved, ctx.has_b[2], skip, du_pp)
        dt_pp = plane_pool.get(n, w1.shape[0], h, wd, x.device)
        signs, ctx.signs = ctx.signs, None
        conv2d_dgrad_pp_raw(du_pp if du_pp is not None else du, w2, actmask=t_pp if signs is None else None, slope=0.2, out_pp=dt_pp, signs=signs)   # * LeakyReLU'(t), planes out
        dw2 = db2 = dw1 = db1 = None
        main = torch.cuda.current_stream()
        t_done = dt_done = False
        slots = _WGRAD_SLOTS and _state.ready_pairs is not None
        if slots:                                         # conv1's data gradient FIRST, then one waiting pair of weight gradients (release_ready_pair)
            dx = conv2d_dgrad_pp_raw(dt_pp, w1, residual=g, extra=_carry_take(ctx.carry)) if ctx.needs_input_grad[0] else None
            release_ready_pair(1)
        if not skip:
            t_done = wgrad_pp_for_params(w2, b2, t_pp, du_pp if du_pp is not None else du, ctx.has_b[1],
                                         release=(t_pp,) if du_pp is None else (t_pp, du_pp))
            if not t_done:                                # autograd wants the gradients returned: the fp32 path on converted operands
                dw2, db2 = wgrad_for_params(w2, b2, pp_to_f32(t_pp), du if du is not None else pp_to_f32(du_pp), 1, 1, ctx.has_b[1])
        if not slots:
            dx = conv2d_dgrad_pp_raw(dt_pp, w1, residual=g, extra=_carry_take(ctx.carry)) if ctx.needs_input_grad[0] else None   # + skip gradient (+ the input's stashed gradients)
        if not skip:
            dt_done = wgrad_pp_for_params(w1, b1, x_pp if x_pp is not None else x, dt_pp, ctx.has_b[0],
                                          release=(dt_pp,) if x_pp is None else (dt_pp, x_pp))
            if not dt_done:
                dw1, db1 = wgrad_for_params(w1, b1, x, pp_to_f32(dt_pp), 1, 1, ctx.has_b[0])
        if not t_done:
            plane_pool.put(t_pp, (main,))
            if du_pp is not None:
                plane_pool.put(du_pp, (main,))
        if not dt_done:
            plane_pool.put(dt_pp, (main,))
            if x_pp is not None:
                plane_pool.put(x_pp, (main,))
        if slots and x_pp is None:                        # the trunk's first block = the backward's last: no further slot will come
            release_ready_pair(len(_state.ready_pairs))
        if ctx.group_first:
            flush_pending_pp()                            # a launch never straddles ResGroups: what the exchange's group-boundary flush finds is the same with and without it
        return dx, dw1, db1, dw2, db2, dfc1, dfc2, dw7, dwc, dbc, None, None, None, None


def rab_block(x, w1, b1, w2, b2, fc1_w, fc2_w, w7, wc, bc, emit_pp=False, group_first=False):
    if _tail_eval_ok(x):                                 # inference: three conv-sized launches + the pooling partials per block
        _require_gpu(x, 'rab_block')
        x = nhwc(x)
        if rab_planes_ok(x, w1, w2):
            n, _, h, wd = x.shape
            t_pp = plane_pool.get(n, w1.shape[0], h, wd, x.device)
            conv2d_fwd_pp_raw(x, w1, b1, 0.2, out_pp=t_pp)
            if pool_epilogue_ok(x, w2):
                u, pool = conv2d_fwd_pp_raw(t_pp, w2, b2, pool=True)
            else:
                u, pool = conv2d_fwd_pp_raw(t_pp, w2, b2), None
            plane_pool.put(t_pp, (torch.cuda.current_stream(),))
            return _tail_forward_eval(u, x, fc1_w, fc2_w, w7, wc, bc, pool)
        t = conv2d_fwd_raw(x, w1, b1, 1, 1, 0.2)
        if pool_epilogue_ok(t, w2):
            u, pool = conv2d_fwd_pool_raw(t, w2, b2)
        else:
            u, pool = conv2d_fwd_raw(t, w2, b2, 1, 1), None
        return _tail_forward_eval(u, x, fc1_w, fc2_w, w7, wc, bc, pool)
    # emit_pp (the caller knows that another RAB consumes the output): the tail's 1x1 conv leaves the output also as padded planes; they
    # travel as an attribute of the output tensor (with its version counter: an in-place change of the tensor voids them)
    x_pp = None
    tag = getattr(x, '_srhip_pp', None) if _X_PP else None
    if tag is not None and tag[1] == x._version and tag[0].shape == tuple(x.shape):
        x_pp = tag[0]
    out = _RabBlock.apply(x, w1, b1, w2, b2, fc1_w, fc2_w, w7, wc, bc, x_pp, bool(emit_pp and _X_PP), getattr(x, '_srhip_carry', None) if _CARRY else None,
                          bool(group_first))
    pp, _state.last_out_pp = getattr(_state, 'last_out_pp', None), None
    if pp is not None:
        out._srhip_pp = (pp, out._version)
    return out


def attention_tail_supported(u, fc1_w, w7, wc):
    return (u.is_cuda and u.shape[1] == 64 and fc1_w.shape[0] <= 16 and tuple(w7.shape) == (1, 2, 7, 7)
            and tuple(wc.shape[:2]) == (64, 64))


def attention_tail(u, skip, fc1_w, fc2_w, w7, wc, bc, emit_pp=False):
    """conv1x1(SLAM(CLAM(u))) + bc + skip  for la_mode 'CA-SA', pool 'Avg|Max', addconv, C = 64.  emit_pp (the caller knows that a
    fused RAB consumes the output: a ResGroup followed by a ResGroup): the 1x1 conv leaves the output also as padded planes, which
    travel as an attribute of the output tensor like rab_block's -- the next group's first RAB and its weight gradient read them
    instead of running a conversion pass."""
    if _tail_eval_ok(u):
        _require_gpu(u, 'attention_tail')
        return _tail_forward_eval(u, skip, fc1_w, fc2_w, w7, wc, bc)
    emit = bool(emit_pp and _X_PP and _PP_RAB and get_conv_math() == 'bf16x3' and u.is_cuda)
    out = _AttentionTail.apply(u, skip, fc1_w, fc2_w, w7, wc, bc, _carry_commit(skip), emit)
    pp, _state.last_out_pp = getattr(_state, 'last_out_pp', None), None
    if pp is not None:
        out._srhip_pp = (pp, out._version)
    return out


# --------------------------------------------------------------------------------------------- #
# channel / spatial attention outside the fused RAB tail: the discriminator's ChannelAttention(256) /
# SpatialAttention (base_networks.py:366-457, sradsgan.py:495-496) and the stand-alone CLAM / SLAM modules.
# Built from the srhip_cbam_* primitives, whose backward passes are again primitives: every Function below is
# differentiable as often as autograd asks (the gradient penalty differentiates the discriminator twice).
# --------------------------------------------------------------------------------------------- #


def _nhwc_dims(x):
    n, c, h, w = x.shape
    return n, h * w, c


class _PoolHW(Function):
    """[n,c,h,w] -> t [n,2,c] = (mean, max) over pixels; the first arg-max pixel is remembered."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, 'cbam_pool_hw')
        x = nhwc(x)
        n, hw, c = _nhwc_dims(x)
        t = torch.empty(n, 2, c, device=x.device, dtype=torch.float32)
        arg = torch.empty(n, c, device=x.device, dtype=torch.int32)
        _hip.check(_hip.lib().srhip_cbam_pool_hw(_p(x), _p(t), _p(arg), 0, n, hw, c, _stream()), 'cbam_pool_hw')
        ctx.arg, ctx.shape = arg, tuple(x.shape)
        return t

    @staticmethod
    def backward(ctx, gt):
        return _UnpoolHW.apply(gt, ctx.arg, ctx.shape)


class _UnpoolHW(Function):
    """t [n,2,c] -> [n,c,h,w]: t0 / hw everywhere + t1 at the remembered arg-max pixel (adjoint of the pooling)."""

    @staticmethod
    def forward(ctx, t, arg, shape):
        t = t.contiguous()
        n, c, h, w = shape
        out = torch.empty(shape, device=t.device, dtype=torch.float32, memory_format=CL)     # (not empty().contiguous(CL): that is a copy kernel)
        _hip.check(_hip.lib().srhip_cbam_unpool_hw(_p(t), _p(arg), _p(out), n, h * w, c, _stream()), 'cbam_unpool_hw')
        ctx.arg = arg
        return out

    @staticmethod
    def backward(ctx, g):
        return _RepoolHW.apply(g, ctx.arg), None, None


class _RepoolHW(Function):
    """pooling at a GIVEN arg-max (linear in x)."""

    @staticmethod
    def forward(ctx, x, arg):
        x = nhwc(x)
        n, hw, c = _nhwc_dims(x)
        t = torch.empty(n, 2, c, device=x.device, dtype=torch.float32)
        _hip.check(_hip.lib().srhip_cbam_pool_hw(_p(x), _p(t), _p(arg), 1, n, hw, c, _stream()), 'cbam_pool_hw')
        ctx.arg, ctx.shape = arg, tuple(x.shape)
        return t

    @staticmethod
    def backward(ctx, gt):
        return _UnpoolHW.apply(gt, ctx.arg, ctx.shape), None


class _PoolC(Function):
    """[n,c,h,w] -> t [n,2,h,w] (NHWC memory [n][hw][2]) = (mean, max) over channels; first arg-max channel remembered."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, 'cbam_pool_c')
        x = nhwc(x)
        n, c, h, w = x.shape
        t = empty_nhwc(n, 2, h, w, x)
        argc = torch.empty(n, h * w, device=x.device, dtype=torch.int32)
        _hip.check(_hip.lib().srhip_cbam_pool_c(_p(x), _p(t), _p(argc), 0, n, h * w, c, _stream()), 'cbam_pool_c')
        ctx.argc, ctx.shape = argc, tuple(x.shape)
        return t

    @staticmethod
    def backward(ctx, gt):
        return _UnpoolC.apply(gt, ctx.argc, ctx.shape)


class _UnpoolC(Function):
    @staticmethod
    def forward(ctx, t, argc, shape):
        t = nhwc(t)
        n, c, h, w = shape
        out = torch.empty(shape, device=t.device, dtype=torch.float32, memory_format=CL)     # (not empty().contiguous(CL): that is a copy kernel)
        _hip.check(_hip.lib().srhip_cbam_unpool_c(_p(t), _p(argc), _p(out), n, h * w, c, _stream()), 'cbam_unpool_c')
        ctx.argc = argc
        return out

    @staticmethod
    def backward(ctx, g):
        return _RepoolC.apply(g, ctx.argc), None, None


class _RepoolC(Function):
    @staticmethod
    def forward(ctx, x, argc):
        x = nhwc(x)
        n, c, h, w = x.shape
        t = empty_nhwc(n, 2, h, w, x)
        _hip.check(_hip.lib().srhip_cbam_pool_c(_p(x), _p(t), _p(argc), 1, n, h * w, c, _stream()), 'cbam_pool_c')
        ctx.argc, ctx.shape = argc, tuple(x.shape)
        return t

    @staticmethod
    def backward(ctx, gt):
        return _UnpoolC.apply(gt, ctx.argc, ctx.shape), None


class _Scale(Function):
    """x * s with s broadcast over pixels (mode 0, s [n,c]) or over channels (mode 1, s [n,hw]); bilinear."""

    @staticmethod
    def forward(ctx, x, s, mode):
        _require_gpu(x, 'cbam_scale')
        xc, sc = nhwc(x), s.contiguous()
        n, hw, c = _nhwc_dims(xc)
        out = torch.empty_like(xc, memory_format=CL)
        _hip.check(_hip.lib().srhip_cbam_scale(_p(xc), _p(sc), _p(out), n, hw, c, mode, _stream()), 'cbam_scale')
        ctx.mode = mode
        ctx.save_for_backward(x, s)
        return out

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        dx = _Scale.apply(g, s, ctx.mode) if ctx.needs_input_grad[0] else None
        ds = _Dot.apply(g, x, ctx.mode).view(s.shape) if ctx.needs_input_grad[1] else None
        return dx, ds, None


class _Dot(Function):
    """sum over pixels (mode 0 -> [n,c]) or over channels (mode 1 -> [n,hw]) of a * b; bilinear."""

    @staticmethod
    def forward(ctx, a, b, mode):
        _require_gpu(a, 'cbam_dot')
        ac, bc = nhwc(a), nhwc(b)
        n, hw, c = _nhwc_dims(ac)
        out = torch.empty((n, c) if mode == 0 else (n, hw), device=a.device, dtype=torch.float32)
        _hip.check(_hip.lib().srhip_cbam_dot(_p(ac), _p(bc), _p(out), n, hw, c, mode, _stream()), 'cbam_dot')
        ctx.mode = mode
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = _Scale.apply(b, g, ctx.mode) if ctx.needs_input_grad[0] else None
        db = _Scale.apply(a, g, ctx.mode) if ctx.needs_input_grad[1] else None
        return da, db, None


class _Sigmoid(Function):
    """pair == 0: elementwise; pair != 0: x [n,2,c] -> sigmoid(x[:,0] + x[:,1]) (the two MLP branches, base_networks.py:401)."""

    @staticmethod
    def forward(ctx, x, pair):
        _require_gpu(x, 'sigmoid')
        xc = x.contiguous()
        c = xc.shape[-1] if pair else 1
        y = torch.empty((xc.shape[0], c) if pair else tuple(xc.shape), device=x.device, dtype=torch.float32)
        _hip.check(_hip.lib().srhip_sigmoid_fwd(_p(xc), _p(y), y.numel(), c, int(pair), _stream()), 'sigmoid_fwd')
        ctx.pair, ctx.xshape = pair, tuple(x.shape)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return _SigmoidBwd.apply(g, y, ctx.pair, ctx.xshape), None


class _SigmoidBwd(Function):
    @staticmethod
    def forward(ctx, g, y, pair, xshape):
        gc = g.contiguous()
        dx = torch.empty(xshape, device=g.device, dtype=torch.float32)
        _hip.check(_hip.lib().srhip_sigmoid_bwd(_p(gc), _p(y), _p(dx), y.numel(), y.shape[-1] if pair else 1, int(pair), _stream()),
                   'sigmoid_bwd')
        ctx.pair = pair
        ctx.save_for_backward(g, y)
        return dx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gg):
        g, y = ctx.saved_tensors
        ggc, gc = gg.contiguous(), g.contiguous()
        dg = torch.empty_like(gc) if ctx.needs_input_grad[0] else None
        dy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        _hip.check(_hip.lib().srhip_sigmoid_bwd_bwd(_p(ggc), _p(gc), _p(y), _p(dg), _p(dy), y.numel(),
                                                    y.shape[-1] if ctx.pair else 1, int(ctx.pair), _stream()), 'sigmoid_bwd_bwd')
        return dg, dy, None, None


def clam(x, fc1_w, fc2_w, pool_mode='Avg|Max'):
    """sradsgan.py:117-127 / base_networks.py:387-403: sigmoid(MLP(avgpool x) + MLP(maxpool x)) * x; the shared
    bias-free MLP is the reference's two 1x1 convs, applied to the stacked (avg, max) rows in one pass."""
    if pool_mode != 'Avg|Max':
        raise NotImplementedError("clam: only pool_mode 'Avg|Max' is built by the SRADSGAN path (sradsgan.py:669-671)")
    n, c = x.shape[0], x.shape[1]
    t = _PoolHW.apply(x)                                                  # [n,2,c]
    rows = t.view(2 * n, 1, 1, c).permute(0, 3, 1, 2)                     # logical [2n,c,1,1], NHWC memory: a free view
    hid = conv2d(rows, fc1_w, None, 1, 0, act_slope=0.0)                  # ReLU fused
    logits = conv2d(hid, fc2_w, None, 1, 0)                               # [2n,c,1,1]
    s = _Sigmoid.apply(logits.permute(0, 2, 3, 1).reshape(n, 2, c), True)  # [n,c]
    return _Scale.apply(x, s, 0)


def slam(x, w7, pool_mode='Avg|Max'):
    """sradsgan.py:141-151 / base_networks.py:440-457: sigmoid(conv kxk([mean_c x, max_c x])) * x."""
    if pool_mode != 'Avg|Max':
        raise NotImplementedError("slam: only pool_mode 'Avg|Max' is built by the SRADSGAN path (sradsgan.py:669-671)")
    n, c, h, w = x.shape
    pooled = _PoolC.apply(x)                                              # [n,2,h,w]
    logit = conv2d(pooled, w7, None, 1, w7.shape[-1] // 2)                # [n,1,h,w]
    m = _Sigmoid.apply(logit.reshape(n, h * w), False)
    return _Scale.apply(x, m, 1)


def _ws(nbytes, like):
    return torch.empty((max(int(nbytes), 4) + 3) // 4, device=like.device, dtype=torch.float32)


def _gamma_slot(gamma):
    """(buffer, accumulate flag) for a scalar attention gamma: its arena slot in direct_param_grads() mode."""
    g = _grad_slot(gamma)
    if g is not None:
        return g, 1
    return torch.empty(1, device=gamma.device, dtype=torch.float32), 0


class _Cgam(Function):
    """Channel global attention, sradsgan.py:202-212 (srhip_cgam_fwd / srhip_cgam_bwd)."""

    @staticmethod
    def forward(ctx, x, gamma):
        _require_gpu(x, 'cgam')
        x = nhwc(x)
        n, c, h, w = x.shape
        lib = _hip.lib()
        y = torch.empty_like(x, memory_format=CL)
        att = torch.empty(n, c, c, device=x.device, dtype=torch.float32)
        ws = _ws(lib.srhip_cgam_workspace(n, h * w), x)
        gam = gamma.detach().reshape(1).contiguous()
        _hip.check(lib.srhip_cgam_fwd(_p(x), _p(gam), _p(y), _p(att), _p(ws), ws.numel() * 4, n, h * w, c, _stream()),
                   'cgam_fwd')
        ctx.save_for_backward(x, att, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, att, gamma = ctx.saved_tensors
        dy = nhwc(dy)
        n, c, h, w = x.shape
        lib = _hip.lib()
        dx = torch.empty_like(x, memory_format=CL)
        skip = _skip_param_grads(gamma) or not ctx.needs_input_grad[1]
        dgam, acc = (None, 0) if skip else _gamma_slot(gamma)
        ws = _ws(lib.srhip_cgam_workspace(n, h * w), x)
        gam = gamma.detach().reshape(1).contiguous()
        _hip.check(lib.srhip_cgam_bwd(_p(dy), _p(x), _p(att), _p(gam), _p(dx), _p(dgam), acc, _p(ws), ws.numel() * 4,
                                      n, h * w, c, _stream()), 'cgam_bwd')
        return dx, (None if (skip or acc) else dgam.view(gamma.shape))


def cgam(x, gamma):
    """sradsgan.py:202-212; softmax(rowmax(E)-E) == softmax(-E) (shift invariance)."""
    return _Cgam.apply(x, gamma)


class _SgamCore(Function):
    """Position global attention on projected q, k, v (sradsgan.py:165-175): flash-style, no N x N tensor
    (srhip_sgam_flash_fwd / srhip_sgam_flash_bwd).  Saves q, k, v, o = softmax(qk^T) v and the per-query
    log-sum-exp."""

    @staticmethod
    def forward(ctx, x, q, k, v, gamma):
        _require_gpu(x, 'sgam')
        x, q, k, v = nhwc(x), nhwc(q), nhwc(k), nhwc(v)
        n, c, h, w = x.shape
        lib = _hip.lib()
        y = torch.empty_like(x, memory_format=CL)
        o = torch.empty_like(x, memory_format=CL)
        lse = torch.empty(n, h * w, device=x.device, dtype=torch.float32)
        gam = gamma.detach().reshape(1).contiguous()
        _hip.check(lib.srhip_sgam_flash_fwd(_p(q), _p(k), _p(v), _p(x), _p(gam), _p(y), _p(o), _p(lse), n, h * w,
                                            q.shape[1], c, _stream()), 'sgam_flash_fwd')
        ctx.save_for_backward(q, k, v, o, lse, gamma)
        return y

    @staticmethod
    def backward(ctx, dy):
        q, k, v, o, lse, gamma = ctx.saved_tensors
        dy = nhwc(dy)
        n, c, h, w = v.shape
        lib = _hip.lib()
        dq, dk = torch.empty_like(q, memory_format=CL), torch.empty_like(k, memory_format=CL)
        dv = torch.empty_like(v, memory_format=CL)
        skip = _skip_param_grads(gamma) or not ctx.needs_input_grad[4]
        dgam, acc = (None, 0) if skip else _gamma_slot(gamma)
        ws = _ws(lib.srhip_sgam_flash_bwd_workspace(n, h * w), v)
        gam = gamma.detach().reshape(1).contiguous()
        _hip.check(lib.srhip_sgam_flash_bwd(_p(dy), _p(q), _p(k), _p(v), _p(o), _p(lse), _p(gam), _p(dq), _p(dk), _p(dv),
                                            _p(dgam), acc, _p(ws), ws.numel() * 4, n, h * w, q.shape[1], c, _stream()),
                   'sgam_flash_bwd')
        return _passed_through(dy), dq, dk, dv, (None if (skip or acc) else dgam.view(gamma.shape))


def sgam(x, q, k, v, gamma):
    """sradsgan.py:165-175 with q, k, v already projected by the 1x1 convs (:157-159)."""
    return _SgamCore.apply(x, q, k, v, gamma)


def _bn_reference_bwd(dy, x, y, gamma, eps, slope):
    """First-order backward of train-mode BN(+LeakyReLU) written with differentiable torch ops; only
    used to differentiate it once more (gradient penalty, sradsgan.py:621,639)."""
    n = x.numel() // x.shape[1]
    dz = dy if slope is None else dy * torch.where(y > 0, 1.0, float(slope))
    mean = x.mean((0, 2, 3), keepdim=True)
    var = x.var((0, 2, 3), unbiased=False, keepdim=True)
    invstd = torch.rsqrt(var + eps)
    xhat = (x - mean) * invstd
    dbeta = dz.sum((0, 2, 3))
    dgamma = (dz * xhat).sum((0, 2, 3))
    dx = (gamma.view(1, -1, 1, 1) * invstd) * (dz - dbeta.view(1, -1, 1, 1) / n - xhat * dgamma.view(1, -1, 1, 1) / n)
    return dx, dgamma, dbeta


_BN_BWD_X = os.environ.get('SRHIP_BN_BWD_X', '1') == '1'      # A/B knob: 0 = the BatchNorm backward reads y for the LeakyReLU mask (rounds 1-4)
_BN_FOLD = os.environ.get('SRHIP_BN_FOLD', '1') == '1'        # A/B knob: 0 = autograd sums a BatchNorm input's two gradients in the penalty's double backward


@contextlib.contextmanager
def bn_fold_second_order():
    """Around ONE backward through a gradient penalty's double-backward graph (sradsgan.py:621-639, 886): a BatchNorm input x receives two
    gradients there -- from the first-order backward's node (d penalty / d x through the batch statistics and the normalised value) and,
    later, from the forward node.  Inside this context the first is held back (not handed to autograd) and the forward node's backward
    adds it in its apply pass (srhip_bn_train_bwd_acc_xa, in place): one pass over three tensors less per BatchNorm layer.  The second-
    order nodes of a layer run before its forward node (their results feed it through the layers behind); a held-back gradient that
    no forward node has taken when the backward ends is an error, not a silent loss."""
    if not _BN_FOLD:
        yield
        return
    prev, _state.bn_fold = getattr(_state, 'bn_fold', None), {}
    ok = False
    try:
        yield
        ok = True
    finally:
        left, _state.bn_fold = _state.bn_fold, prev
        if ok and left:
            raise RuntimeError('bn_fold_second_order: %d BatchNorm input gradient(s) of the double backward were never taken by a forward node' % len(left))


class _BNTrainBwd(Function):
    @staticmethod
    def forward(ctx, dy, x, y, gamma, mean, invstd, eps, slope, acc_gamma=None, acc_beta=None, beta=None, addend=None):
        # addend (bn_fold_second_order): x's other gradient, added in the apply pass, in place (dx IS that buffer afterwards)
        # acc_gamma / acc_beta: the parameters' gradient slots (direct_param_grads mode): the kernel adds into them itself
        # (srhip_bn_train_bwd_acc) instead of two add launches per BatchNorm backward
        # NB: save the tensors autograd handed us (not layout-converted copies), or the second-order
        # graph through x / dy would be cut
        dyc, xc, yc = nhwc(dy), nhwc(x), nhwc(y)
        n, c, h, w = x.shape
        rows = n * h * w
        lib = _hip.lib()
        fold = addend is not None and beta is not None and slope is not None and _BN_BWD_X and addend.is_contiguous(memory_format=CL)
        dx = addend if fold else torch.empty_like(xc, memory_format=CL)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(lib.srhip_bn_workspace(rows, c) // 4, device=x.device, dtype=torch.float32)
        if fold:
            _hip.check(lib.srhip_bn_train_bwd_acc_xa(_p(dyc), _p(xc), _p(gamma.detach().contiguous()), _p(beta.detach().contiguous()), _p(mean),
                                                     _p(invstd), _p(addend), _p(dx), _p(dgamma), _p(dbeta), _p(acc_gamma), _p(acc_beta), _p(ws),
                                                     ws.numel() * 4, rows, c, float(slope), 1, _stream()), 'bn_train_bwd_xa')
        elif beta is not None and slope is not None and _BN_BWD_X:
            # the LeakyReLU mask from the recomputed pre-activation instead of a read of y (srhip_bn_train_bwd_acc_x): the same bits
            _hip.check(lib.srhip_bn_train_bwd_acc_x(_p(dyc), _p(xc), _p(gamma.detach().contiguous()), _p(beta.detach().contiguous()), _p(mean),
                                                    _p(invstd), _p(dx), _p(dgamma), _p(dbeta), _p(acc_gamma), _p(acc_beta), _p(ws),
                                                    ws.numel() * 4, rows, c, float(slope), 1, _stream()), 'bn_train_bwd_x')
        else:
            _hip.check(lib.srhip_bn_train_bwd_acc(_p(dyc), _p(xc), _p(yc), _p(gamma.detach().contiguous()), _p(mean), _p(invstd),
                                                  _p(dx), _p(dgamma), _p(dbeta), _p(acc_gamma), _p(acc_beta), _p(ws), ws.numel() * 4,
                                                  rows, c, float(slope or 0.0), int(slope is not None), _stream()), 'bn_train_bwd')
        if addend is not None and not fold:
            dx.add_(addend)
        ctx.eps, ctx.slope = eps, slope
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(dy, x, y, gamma, mean, invstd, beta)
        return dx, dgamma, dbeta

    @staticmethod
    def backward(ctx, ddx, ddgamma, ddbeta):
        dy, x, y, gamma, mean, invstd, beta = ctx.saved_tensors
        if ddx is None and ddgamma is None and ddbeta is None:
            return (None,) * 12
        if ddx is not None and ddgamma is None and ddbeta is None and not torch.is_grad_enabled():
            # the gradient-penalty case: one fused second-order pass (3 launches instead of ~40 ATen ops)
            uc, dyc, xc, yc = nhwc(ddx), nhwc(dy), nhwc(x), nhwc(y)
            n, c, h, w = x.shape
            rows = n * h * w
            lib = _hip.lib()
            g_dy, g_x = torch.empty_like(xc, memory_format=CL), torch.empty_like(xc, memory_format=CL)
            g_gamma = torch.empty_like(gamma)
            ws = torch.empty(lib.srhip_bn_bwd2_workspace(rows, c) // 4, device=x.device, dtype=torch.float32)
            slot = None if _skip_param_grads(gamma) else _grad_slot(gamma)   # direct_param_grads(): straight into the arena, by the kernel
            if beta is not None and ctx.slope is not None and _BN_BWD_X:
                _hip.check(lib.srhip_bn_train_bwd_bwd_acc_x(_p(uc), _p(dyc), _p(xc), _p(gamma.detach().contiguous()), _p(beta.detach().contiguous()),
                                                            _p(mean), _p(invstd), _p(g_dy), _p(g_x), _p(g_gamma), _p(slot), _p(ws),
                                                            ws.numel() * 4, rows, c, float(ctx.slope), 1, _stream()), 'bn_train_bwd_bwd_x')
            else:
                _hip.check(lib.srhip_bn_train_bwd_bwd_acc(_p(uc), _p(dyc), _p(xc), _p(yc), _p(gamma.detach().contiguous()),
                                                          _p(mean), _p(invstd), _p(g_dy), _p(g_x), _p(g_gamma), _p(slot), _p(ws),
                                                          ws.numel() * 4, rows, c, float(ctx.slope or 0.0),
                                                          int(ctx.slope is not None), _stream()), 'bn_train_bwd_bwd')
            if slot is not None:
                g_gamma = None
            fold = getattr(_state, 'bn_fold', None)
            if fold is not None and ctx.needs_input_grad[1] and (x.data_ptr(), tuple(x.shape)) not in fold:
                fold[(x.data_ptr(), tuple(x.shape))] = g_x     # (bn_fold_second_order) the forward node of this layer adds it to its own
                g_x = None
            return g_dy, g_x, None, g_gamma, None, None, None, None, None, None, None, None
        with torch.enable_grad():
            dy_, x_, g_ = (t.detach().requires_grad_(True) for t in (dy, x, gamma))
            outs = _bn_reference_bwd(dy_, x_, y, g_, ctx.eps, ctx.slope)
            pairs = [(o, d) for o, d in zip(outs, (ddx, ddgamma, ddbeta)) if d is not None]
            gdy, gx, gg = torch.autograd.grad([o for o, _ in pairs], [dy_, x_, g_], [d for _, d in pairs],
                                              allow_unused=True)
        return gdy, gx, None, gg, None, None, None, None, None, None, None, None


class _BNTrainFwd(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, slope):
        _require_gpu(x, 'batch_norm')
        xc = nhwc(x)
        n, c, h, w = x.shape
        rows = n * h * w
        lib = _hip.lib()
        y = torch.empty_like(xc, memory_format=CL)
        mean, invstd = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(lib.srhip_bn_workspace(rows, c) // 4, device=x.device, dtype=torch.float32)
        _hip.check(lib.srhip_bn_train_fwd(_p(xc), _p(gamma.detach().contiguous()), _p(beta.detach().contiguous()),
                                          _p(running_mean), _p(running_var), _p(y), _p(mean), _p(invstd), _p(ws),
                                          ws.numel() * 4, rows, c, float(eps), float(momentum), float(slope or 0.0),
                                          int(slope is not None), _stream()), 'bn_train_fwd')
        ctx.eps, ctx.slope = eps, slope
        ctx.save_for_backward(x, y, gamma, mean, invstd, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, invstd, beta = ctx.saved_tensors
        fold = getattr(_state, 'bn_fold', None)
        held = fold.pop((x.data_ptr(), tuple(x.shape)), None) if (fold and not torch.is_grad_enabled()) else None
        if not torch.is_grad_enabled() and not _skip_param_grads(gamma, beta):
            # direct_param_grads(): add into the arena slots here, on the stream this node runs on, instead of handing the
            # gradients to autograd -- its AccumulateGrad nodes run on the stream the parameter was FIRST used on in this
            # iteration (the main stream, D(gen_hr)), which would make the main stream wait for the D stream's backward
            gg, gb = _grad_slot(gamma), _grad_slot(beta)
            if gg is not None and gb is not None:
                dx, _, _ = _BNTrainBwd.apply(dy, x, y, gamma, mean, invstd, ctx.eps, ctx.slope, gg, gb, beta, held)
                return dx, None, None, None, None, None, None, None
        dx, dgamma, dbeta = _BNTrainBwd.apply(dy, x, y, gamma, mean, invstd, ctx.eps, ctx.slope, None, None, beta, held)
        return dx, dgamma, dbeta, None, None, None, None, None


def batch_norm_act(x, bn, slope=None):
    """Train-mode BatchNorm2d + LeakyReLU (sradsgan.py:478-479) on the fused HIP kernels; updates the
    running statistics like nn.BatchNorm2d (momentum 0.1, unbiased running_var).  Twice differentiable."""
    if not bn.training:
        # eval(): running statistics, a per-channel affine (SRGAN's generator at validation time; the SRADSGAN
        # discriminator is never put in eval()).  Inference only: srhip_bn_eval_fwd has no backward.
        _require_gpu(x, 'batch_norm')
        if torch.is_grad_enabled() and (x.requires_grad or bn.weight.requires_grad):
            raise NotImplementedError('batch_norm_act: eval()-mode BatchNorm is built for inference only (wrap the call in '
                                      'torch.no_grad()); training through frozen statistics is not on the SRADSGAN path')
        xc = nhwc(x)
        n, c, h, w = xc.shape
        y = torch.empty_like(xc, memory_format=CL)
        _hip.check(_hip.lib().srhip_bn_eval_fwd(_p(xc), _p(bn.weight.detach().contiguous()), _p(bn.bias.detach().contiguous()),
                                                _p(bn.running_mean), _p(bn.running_var), _p(y), n * h * w, c, float(bn.eps),
                                                float(slope or 0.0), int(slope is not None), _stream()), 'bn_eval_fwd')
        return y
    y = _BNTrainFwd.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, slope)
    with torch.no_grad():
        bn.num_batches_tracked += 1
    stash = getattr(bn, '_stat_stash', None)
    if stash is not None and y.grad_fn is not None:
        # (batch mean, invstd, count) of this call, so the caller can replay the running-statistics update
        # of a forward pass it does not recompute (TrainStep: D(fake) == D(gen_hr))
        x_, y_, g_, mean, invstd, b_ = y.grad_fn.saved_tensors
        stash.append((bn, mean, invstd, x.numel() // x.shape[1]))
    return y


def replay_bn_update(stash):
    """Applies nn.BatchNorm2d's running-statistics update once more for every stashed (bn, mean, invstd, n)."""
    if not stash:
        return
    with torch.no_grad():
        if len({(bn.momentum, bn.eps) for bn, _, _, _ in stash}) == 1 and len(stash) > 1:
            # one launch per arithmetic step for all layers (the same element-wise operations as the per-layer form below)
            bn0 = stash[0][0]
            m, eps = bn0.momentum, bn0.eps
            var = torch._foreach_pow([t[2] for t in stash], -2)
            torch._foreach_sub_(var, eps)
            torch._foreach_mul_(var, [n / max(n - 1, 1) for _, _, _, n in stash])
            rm, rv = [t[0].running_mean for t in stash], [t[0].running_var for t in stash]
            torch._foreach_mul_(rm, 1 - m)
            torch._foreach_add_(rm, [t[1] for t in stash], alpha=m)
            torch._foreach_mul_(rv, 1 - m)
            torch._foreach_add_(rv, var, alpha=m)
            torch._foreach_add_([t[0].num_batches_tracked for t in stash], 1)
            return
        for bn, mean, invstd, n in stash:
            var = invstd.pow(-2) - bn.eps
            bn.running_mean.mul_(1 - bn.momentum).add_(mean, alpha=bn.momentum)
            bn.running_var.mul_(1 - bn.momentum).add_(var * (n / max(n - 1, 1)), alpha=bn.momentum)
            bn.num_batches_tracked += 1


def max_pool2x2_raw(x):
    _require_gpu(x, 'max_pool2x2')
    x = nhwc(x)
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, h // 2, w // 2, x)
    _hip.check(_hip.lib().srhip_maxpool2x2_fwd(_p(x), _p(y), n, h, w, c, _stream()), 'maxpool2x2_fwd')
    return y


_POOL_IDX = os.environ.get('SRHIP_POOL_IDX', '1') == '1'        # A/B knob: 0 = the max-pool backward re-reads the pool's input


def max_pool2x2_idx_raw(x):
    """(y, rec): the pool and its 2-byte records per 4 outputs (arg-max position + "maximum > 0"), which max_pool2x2_bwd_idx_raw reads
    instead of x."""
    _require_gpu(x, 'max_pool2x2')
    x = nhwc(x)
    n, c, h, w = x.shape
    y = empty_nhwc(n, c, h // 2, w // 2, x)
    rec = torch.empty(n * (h // 2) * (w // 2) * (c // 4), device=x.device, dtype=torch.int16)
    _hip.check(_hip.lib().srhip_maxpool2x2_fwd_idx(_p(x), _p(y), _p(rec), n, h, w, c, _stream()), 'maxpool2x2_fwd_idx')
    return y, rec


def max_pool2x2_bwd_idx_raw(dy, rec, x_shape, relu_input):
    dy = nhwc(dy)
    n, c, h, w = x_shape
    dx = torch.empty(n, h, w, c, device=dy.device, dtype=torch.float32).permute(0, 3, 1, 2)
    _hip.check(_hip.lib().srhip_maxpool2x2_bwd_idx(_p(dy), _p(rec), _p(dx), n, h, w, c, int(relu_input), _stream()), 'maxpool2x2_bwd_idx')
    return dx


def max_pool2x2_bwd_raw(dy, x, relu_input):
    dy, x = nhwc(dy), nhwc(x)
    n, c, h, w = x.shape
    dx = torch.empty_like(x, memory_format=CL)
    _hip.check(_hip.lib().srhip_maxpool2x2_bwd(_p(dy), _p(x), _p(dx), n, h, w, c, int(relu_input), _stream()),
               'maxpool2x2_bwd')
    return dx


class _MaxPool2x2(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return max_pool2x2_raw(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return max_pool2x2_bwd_raw(dy, x, False)


def max_pool2x2(x):
    if x.shape[1] % 4 == 0 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
        return _MaxPool2x2.apply(x)
    raise NotImplementedError('max_pool2x2: the HIP kernel takes C %% 4 == 0 and even H, W (VGG on 216 x 216 tiles); got %s -- '
                              'there is no ATen fallback' % (tuple(x.shape),))


class _VggFeatures(Function):
    """vgg19.features[:12] with frozen weights (sradsgan.py:88-99, 836-838) as ONE autograd node whose
    backward is the data-gradient chain only: every ReLU backward rides in the epilogue of the dgrad
    (or in the max-pool backward) that produces its input gradient -- no separate mask passes."""

    @staticmethod
    def forward(ctx, x, *wb):
        ws, bs = wb[0::2], wb[1::2]
        x = nhwc(x)
        y1 = conv2d_fwd_raw(x, ws[0], bs[0], 1, 1, 0.0)
        y2 = conv2d_fwd_raw(y1, ws[1], bs[1], 1, 1, 0.0)
        idx = _POOL_IDX and ctx.needs_input_grad[0]      # a backward will follow: the pools leave 2-byte records instead of keeping y2 / y4
        p1, r1 = max_pool2x2_idx_raw(y2) if idx else (max_pool2x2_raw(y2), None)
        y3 = conv2d_fwd_raw(p1, ws[2], bs[2], 1, 1, 0.0)
        y4 = conv2d_fwd_raw(y3, ws[3], bs[3], 1, 1, 0.0)
        p2, r2 = max_pool2x2_idx_raw(y4) if idx else (max_pool2x2_raw(y4), None)
        y5 = conv2d_fwd_raw(p2, ws[4], bs[4], 1, 1, 0.0)
        if idx:
            ctx.shapes = (tuple(y2.shape), tuple(y4.shape))
            ctx.save_for_backward(y1, r1, p1, y3, r2, p2, y5, *ws)
        else:
            ctx.shapes = None
            ctx.save_for_backward(y1, y2, p1, y3, y4, p2, y5, *ws)
        ctx.x_shape = tuple(x.shape)
        return y5

    @staticmethod
    def backward(ctx, g):
        y1, y2, p1, y3, y4, p2, y5, w1, w2, w3, w4, w5 = ctx.saved_tensors
        g = lrelu_bwd_raw(g, y5, 0.0)                                            # ReLU after conv3_1 (54x54, small)
        g = conv2d_dgrad_raw(g, w5, tuple(p2.shape), 1, 1)
        g = max_pool2x2_bwd_idx_raw(g, y4, ctx.shapes[1], True) if ctx.shapes else max_pool2x2_bwd_raw(g, y4, True)   # pool2 + ReLU(conv2_2) (y4 = its records)
        g = conv2d_dgrad_raw(g, w4, tuple(y3.shape), 1, 1, None, y3, 0.0)         # + ReLU(conv2_1)
        g = conv2d_dgrad_raw(g, w3, tuple(p1.shape), 1, 1)
        g = max_pool2x2_bwd_idx_raw(g, y2, ctx.shapes[0], True) if ctx.shapes else max_pool2x2_bwd_raw(g, y2, True)   # pool1 + ReLU(conv1_2) (y2 = its records)
        g = conv2d_dgrad_raw(g, w2, tuple(y1.shape), 1, 1, None, y1, 0.0)         # + ReLU(conv1_1)
        g = conv2d_dgrad_raw(g, w1, ctx.x_shape, 1, 1)
        return (g,) + (None,) * 10


def vgg_features(x, weights_and_biases):
    return _VggFeatures.apply(x, *weights_and_biases)


class _L1Mean(Function):
    @staticmethod
    def forward(ctx, a, b):
        _require_gpu(a, 'l1_mean')
        a = nhwc(a) if a.dim() == 4 else a.contiguous()
        b = b.contiguous(memory_format=CL) if (b.dim() == 4) else b.contiguous()
        if a.shape != b.shape:
            raise ValueError('l1_mean: shapes differ: %s vs %s' % (tuple(a.shape), tuple(b.shape)))
        lib = _hip.lib()
        out = torch.empty((), device=a.device, dtype=torch.float32)
        ws = _ws(lib.srhip_reduce_workspace(), a)
        _hip.check(lib.srhip_l1_mean_fwd(_p(a), _p(b), _p(out), _p(ws), ws.numel() * 4, a.numel(), _stream()), 'l1_mean_fwd')
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        need_a, need_b = ctx.needs_input_grad
        da = torch.empty_like(a, memory_format=CL) if a.dim() == 4 else torch.empty_like(a)
        db = (torch.empty_like(a, memory_format=CL) if a.dim() == 4 else torch.empty_like(a)) if need_b else None
        _hip.check(_hip.lib().srhip_l1_mean_bwd(_p(a), _p(b), _p(gout.contiguous()), _p(da), _p(db), a.numel(), _stream()),
                   'l1_mean_bwd')
        return (da if need_a else None), db


def l1_mean(a, b):
    """nn.L1Loss() (sradsgan.py:686,834,838)."""
    return _L1Mean.apply(a, b)


class _Mean(Function):
    @staticmethod
    def forward(ctx, x):
        _require_gpu(x, 'mean')
        xc = x.contiguous()
        lib = _hip.lib()
        out = torch.empty((), device=x.device, dtype=torch.float32)
        ws = _ws(lib.srhip_reduce_workspace(), x)
        _hip.check(lib.srhip_mean_fwd(_p(xc), _p(out), _p(ws), ws.numel() * 4, xc.numel(), _stream()), 'mean_fwd')
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, gout):
        dx = torch.empty(ctx.shape, device=gout.device, dtype=torch.float32)
        _hip.check(_hip.lib().srhip_mean_bwd(_p(gout.contiguous()), _p(dx), dx.numel(), _stream()), 'mean_bwd')
        return dx


def mean(x):
    """The critic means of GANLoss('wgan-gp') (sradsgan.py:61-66)."""
    return _Mean.apply(x)


class _GpPenalty(Function):
    @staticmethod
    def forward(ctx, grads):
        _require_gpu(grads, 'gp_penalty')
        g = nhwc(grads)
        n, c, h, w = g.shape
        lib = _hip.lib()
        out = torch.empty((), device=g.device, dtype=torch.float32)
        ws = _ws(lib.srhip_reduce_workspace(), g)
        _hip.check(lib.srhip_gp_norm_penalty_fwd(_p(g), _p(out), _p(ws), ws.numel() * 4, n * h * w, c, _stream()),
                   'gp_norm_penalty_fwd')
        ctx.save_for_backward(g)
        return out

    @staticmethod
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        n, c, h, w = g.shape
        dg = torch.empty_like(g, memory_format=CL)
        _hip.check(_hip.lib().srhip_gp_norm_penalty_bwd(_p(g), _p(gout.contiguous()), _p(dg), n * h * w, c, _stream()),
                   'gp_norm_penalty_bwd')
        return dg


def gp_penalty(grads):
    """sradsgan.py:630-637: L2 norm over the channel dim (per pixel), LS penalty, mean."""
    return _GpPenalty.apply(grads)
