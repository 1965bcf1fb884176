#!/usr/bin/env python3
"""Timing-only ablations of wgrad_flat_kernel (srhip_debug_set(13, bits)) at the bench shapes, pair launches without bias gradient:
what the data movement, the fragment reads, the MFMAs, the in-place split and the partial stores each cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0'); lib = _hip.lib(); B = 32
def t(fn, iters=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
NAMES = {0: 'full', 1: 'no MFMAs', 2: 'no DMAs', 4: 'no fragment reads', 8: 'no partial stores', 16: 'no split', 3: 'no MFMAs, no DMAs',
         5: 'no MFMAs, no fragment reads', 6: 'no DMAs, no fragment reads', 7: 'no MFMA / DMA / fragment reads', 31: 'nothing (loop skeleton + barriers)', 23: 'only partial stores'}
for cin, cout in ((64, 256), (256, 64)):
    xs = [torch.randn(B, cin, 54, 54, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    dys = [torch.randn(B, cout, 54, 54, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(2)]
    gw = [torch.zeros(cout, cin, 3, 3, device=dev) for _ in range(2)]
    cfg = lib.srhip_conv2d_wgrad_pp_ok(B, 54, 54, cin, cout)
    items = [(xs[i], ops.pp_from_f32(dys[i]), gw[i], None) if cfg == 1 else (ops.pp_from_f32(xs[i]), dys[i], gw[i], None) for i in range(2)]
    for abl in (0, 1, 2, 4, 8, 16, 3, 5, 6, 7, 31, 23):
        lib.srhip_debug_set(13, abl)
        print('%3d -> %3d  abl %2d  %7.1f us   %s' % (cin, cout, abl, t(lambda: ops.conv2d_wgrad_pp_raw(items)), NAMES[abl]))
    lib.srhip_debug_set(13, 0)
