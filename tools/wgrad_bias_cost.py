"""Round 4: what the bias column sum inside wgrad_rowtap_kernel costs (pair launch with / without the bias gradients)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
CL = torch.channels_last
mk = lambda n, c, h, w: torch.randn(n, c, h, w, device=dev).contiguous(memory_format=CL)
B = 32
x64 = [mk(B, 64, 54, 54) for _ in range(2)]
t256 = [mk(B, 256, 54, 54) for _ in range(2)]


def items(xs, dys, cout, cin, bias):
    return [(x, dy, torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev) if bias else None, 1, 1) for x, dy in zip(xs, dys)]


def t(fn, nit=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(nit): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / nit * 1e3


with ops.conv_math('bf16x3'):
    for name, xs, dys, cout, cin in (('conv1 64->256', x64, t256, 256, 64), ('conv2 256->64', t256, x64, 64, 256)):
        ib, inb = items(xs, dys, cout, cin, True), items(xs, dys, cout, cin, False)
        for rnd in range(3):
            print('%s pair: with bias %.1f us   without %.1f us' % (name, t(lambda: ops.conv2d_wgrad_multi_raw(ib)), t(lambda: ops.conv2d_wgrad_multi_raw(inb))), flush=True)
