#!/usr/bin/env python3
"""Timing-only ablations of conv_patch8_kernel (srhip_debug_set(16, bits)) on RAB conv1 fprop (64 -> 256 @ 54x54, B = 32, planes out)."""
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
x = torch.randn(B, 64, 54, 54, device=dev).contiguous(memory_format=torch.channels_last)
w1 = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05); b1 = torch.randn(256, device=dev) * 0.1
out = ops.pp_empty(B, 256, 54, 54, dev)
NAMES = {0: 'full', 1: 'no MFMAs', 2: 'no DMAs', 4: 'no split', 8: 'no epilogue', 16: 'no fragment reads', 17: 'no MFMAs, no fragment reads', 3: 'no MFMAs, no DMAs',
         31: 'skeleton (barriers, waits, address code)', 23: 'only the epilogue', 30: 'only MFMAs', 14: 'MFMAs + fragment reads'}
lib.srhip_debug_set(15, 0)
print('4-wave persistent kernel            %6.1f us' % t(lambda: ops.conv2d_fwd_pp_raw(x, w1, b1, 0.2, out_pp=out)))
lib.srhip_debug_set(15, 1)
for abl in (0, 1, 2, 4, 8, 16, 17, 3, 31, 23, 30, 14):
    lib.srhip_debug_set(16, abl)
    print('abl %2d %-42s %6.1f us' % (abl, NAMES[abl], t(lambda: ops.conv2d_fwd_pp_raw(x, w1, b1, 0.2, out_pp=out))))
lib.srhip_debug_set(16, 0)
