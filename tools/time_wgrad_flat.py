#!/usr/bin/env python3
"""A/B of the flat weight-gradient kernel (conv_wgrad_flat.hip) against wgrad_rowtap_kernel at the bench shapes:
RAB conv1 (64 -> 256) and conv2 (256 -> 64) at [32, ., 54, 54], pair launches, HIP events over back-to-back launches."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sradsgan_amd import ops, _hip

dev = torch.device('cuda:0')
lib = _hip.lib()
B = int(os.environ.get('B', '32'))


def t(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for cin, cout in ((64, 256), (256, 64)):
    xs = [torch.randn(B, cin, 54, 54, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    dys = [torch.randn(B, cout, 54, 54, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    gw = [torch.zeros(cout, cin, 3, 3, device=dev) for _ in range(2)]
    gb = [torch.zeros(cout, device=dev) for _ in range(2)]
    flops = 2.0 * B * 54 * 54 * cin * cout * 9
    old_items = [(xs[i], dys[i], gw[i], gb[i], 1, 1) for i in range(2)]
    us = t(lambda: ops.conv2d_wgrad_multi_raw(old_items))
    print('%3d -> %3d  rowtap pair      %7.1f us  frac %.3f' % (cin, cout, us, 2 * flops / us / 1e6 / 833.3))
    mask = lib.srhip_conv2d_wgrad_pp_ok(B, 54, 54, cin, cout)
    ppx = [ops.pp_from_f32(x) for x in xs]
    ppy = [ops.pp_from_f32(d) for d in dys]
    for fmt in ('one', 'both'):
        if fmt == 'one':
            items = [(xs[i], ppy[i], gw[i], gb[i]) if mask & 1 else (ppx[i], dys[i], gw[i], gb[i]) for i in range(2)]
            blist = (768, 256)
        else:
            items = [(ppx[i], ppy[i], gw[i], gb[i]) for i in range(2)]
            blist = (256, 255, 240, 512)
        for blocks in blist:
            lib.srhip_debug_set(12, blocks)
            for k in (2, 1):
                us = t(lambda: ops.conv2d_wgrad_pp_raw(items[:k]))
                print('%3d -> %3d  flat %-4s x%d blocks %4d %7.1f us  frac %.3f' % (cin, cout, fmt, k, blocks, us, k * flops / us / 1e6 / 833.3))
            nb = [it[:3] + (None,) for it in items]
            us = t(lambda: ops.conv2d_wgrad_pp_raw(nb))
            print('%3d -> %3d  flat %-4s x2 blocks %4d %7.1f us  frac %.3f  (no bias gradient)' % (cin, cout, fmt, blocks, us, 2 * flops / us / 1e6 / 833.3))
    lib.srhip_debug_set(12, 768)
    us = t(lambda: ops.pp_from_f32(dys[0] if cout > cin else xs[0], out=ppy[0] if cout > cin else ppx[0]))
    print('%3d -> %3d  stand-alone pp_from_f32 of the 256-channel operand %7.1f us' % (cin, cout, us))
    us = t(lambda: ops.pp_from_f32(xs[0] if cout > cin else dys[0], out=ppx[0] if cout > cin else ppy[0]))
    print('%3d -> %3d  stand-alone pp_from_f32 of the  64-channel operand %7.1f us' % (cin, cout, us))
