"""Round 4: what the pooling epilogue costs RAB conv2 (256 -> 64) and what the stand-alone pooling pass it replaces costs."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()


def t(fn, nit=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(nit): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / nit * 1e3


with ops.conv_math('bf16x3'):
    for B in (16, 32):
        x = torch.randn(B, 256, 54, 54, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter(torch.randn(64, 256, 3, 3, device=dev) * 0.05)
        b = torch.randn(64, device=dev)
        u = ops.conv2d_fwd_raw(x, w, b, 1, 1)
        sec = B * 64 * 64 * 4
        pool = torch.empty(3 * sec // 4, device=dev)
        P = lambda tt: ctypes.c_void_p(tt.data_ptr())
        for rnd in range(2):
            a = t(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1))
            c = t(lambda: ops.conv2d_fwd_pool_raw(x, w, b))
            p = t(lambda: lib.srhip_clam_pool_partial(P(u), P(pool), sec, B, 54, 54, 64, ops._stream()))
            print('B=%d conv2 %.1f us | conv2 + pooling epilogue %.1f us (+%.1f) | stand-alone pooling pass %.1f us' % (B, a, c, c - a, p), flush=True)
