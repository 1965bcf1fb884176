"""Round 4: the attention tail's 1x1 convs (64 -> 64 at 54 x 54) on their own: forward with bias + residual + both scales, data
gradient; achieved bandwidth against the bytes they must move.  (A streaming rebuild of these convs -- resident blocks, weights
as MFMA fragments in registers -- measured no faster and was dropped: profiles/r04_streaming_1x1_rejected.txt.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
lib = _hip.lib()
dev = torch.device('cuda:0')
CL = torch.channels_last


def t(fn, nit=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(nit): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / nit * 1e3


with ops.conv_math('bf16x3'):
    for B in (32, 16):
        u = torch.randn(B, 64, 54, 54, device=dev).contiguous(memory_format=CL)
        skip = torch.randn(B, 64, 54, 54, device=dev).contiguous(memory_format=CL)
        g = torch.randn(B, 64, 54, 54, device=dev).contiguous(memory_format=CL)
        wc = torch.nn.Parameter(torch.randn(64, 64, 1, 1, device=dev) * 0.1)
        bc = torch.randn(64, device=dev)
        m = torch.rand(B * 54 * 54, device=dev)
        s = torch.rand(B, 64, device=dev)
        mb = B * 64 * 54 * 54 * 4 / 1e6
        for rnd in range(2):
            a = t(lambda: ops.conv2d_fwd_raw(u, wc, bc, 1, 0, None, skip, m, s))
            b = t(lambda: ops.conv2d_dgrad_raw(g, wc, tuple(u.shape), 1, 0))
            print('B=%d 1x1 fwd (bias, residual, scales): %.1f us = %.2f TB/s of %d MB | 1x1 dgrad: %.1f us = %.2f TB/s of %d MB' % (
                B, a, 3 * mb / a, 3 * mb, b, 2 * mb / b, 2 * mb), flush=True)
