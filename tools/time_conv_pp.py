#!/usr/bin/env python3
"""Isolated timing of the RAB's four patch-kernel launches with fp32 tensors and with padded planes (B = 32 and 16, 54 x 54)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0'); lib = _hip.lib()
def t(fn, iters=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
cl = lambda x: x.contiguous(memory_format=torch.channels_last)
for B in (32, 16):
    x = cl(torch.randn(B, 64, 54, 54, device=dev)); du = cl(torch.randn(B, 64, 54, 54, device=dev)); g = cl(torch.randn(B, 64, 54, 54, device=dev))
    w1 = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05); b1 = torch.randn(256, device=dev) * 0.1
    w2 = torch.nn.Parameter(torch.randn(64, 256, 3, 3, device=dev) * 0.05); b2 = torch.randn(64, device=dev) * 0.1
    tt = ops.conv2d_fwd_raw(x, w1, b1, 1, 1, 0.2)
    t_pp = ops.pp_from_f32(tt); dt_pp = ops.pp_empty(B, 256, 54, 54, dev)
    dt = ops.conv2d_dgrad_raw(du, w2, tuple(tt.shape), 1, 1, None, tt, 0.2)
    ops.pp_from_f32(dt, out=dt_pp)
    out_pp = ops.pp_empty(B, 256, 54, 54, dev)
    rows = [
        ('conv1 fprop 64->256 bias+lrelu', lambda: ops.conv2d_fwd_raw(x, w1, b1, 1, 1, 0.2), lambda: ops.conv2d_fwd_pp_raw(x, w1, b1, 0.2, out_pp=out_pp)),
        ('conv2 fprop 256->64 bias', lambda: ops.conv2d_fwd_raw(tt, w2, b2, 1, 1), lambda: ops.conv2d_fwd_pp_raw(t_pp, w2, b2)),
        ('conv2 fprop + pool partials', lambda: ops.conv2d_fwd_pool_raw(tt, w2, b2), lambda: ops.conv2d_fwd_pp_raw(t_pp, w2, b2, pool=True)),
        ('conv2 dgrad 64->256 actmask', lambda: ops.conv2d_dgrad_raw(du, w2, tuple(tt.shape), 1, 1, None, tt, 0.2), lambda: ops.conv2d_dgrad_pp_raw(du, w2, actmask=t_pp, slope=0.2, out_pp=out_pp)),
        ('conv1 dgrad 256->64 residual', lambda: ops.conv2d_dgrad_raw(dt, w1, tuple(x.shape), 1, 1, g), lambda: ops.conv2d_dgrad_pp_raw(dt_pp, w1, residual=g)),
    ]
    x_pp = ops.pp_from_f32(x); du_pp = ops.pp_from_f32(du)
    rows += [
        ('conv1 fprop, x ALSO planes', lambda: ops.conv2d_fwd_pp_raw(x, w1, b1, 0.2, out_pp=out_pp), lambda: ops.conv2d_fwd_pp_raw(x_pp, w1, b1, 0.2, out_pp=out_pp)),
        ('conv2 dgrad, du ALSO planes', lambda: ops.conv2d_dgrad_pp_raw(du, w2, actmask=t_pp, slope=0.2, out_pp=out_pp), lambda: ops.conv2d_dgrad_pp_raw(du_pp, w2, actmask=t_pp, slope=0.2, out_pp=out_pp)),
    ]
    for name, f32, pp in rows:
        print('B=%2d %-32s fp32 tensors %6.1f us   planes %6.1f us' % (B, name, t(f32), t(pp)))
