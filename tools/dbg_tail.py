import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops
DEV = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
n, c, h, w = 2, 64, 54, 54
u = torch.randn(n, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
skip = torch.randn(n, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
def P(t): return torch.nn.Parameter(t.to(DEV))
fc1 = P(torch.randn(4, c, 1, 1, generator=g) * 0.3); fc2 = P(torch.randn(c, 4, 1, 1, generator=g) * 0.3)
w7 = P(torch.randn(1, 2, 7, 7, generator=g) * 0.2); wc = P(torch.randn(c, c, 1, 1, generator=g) * 0.1); bc = P(torch.randn(c, generator=g))
z = lambda t: P(torch.zeros_like(t))
from sradsgan_amd import _hip
_hip.lib().srhip_debug_set(0, -1)
with ops.conv_math('bf16x3'):
    for name, (a1, a2, a7, sk, bb) in {'all': (fc1, fc2, w7, skip, bc), 'fc2=0,w7=0': (fc1, z(fc2), z(w7), skip, bc), 'fc2=0': (fc1, z(fc2), w7, skip, bc),
                                   'w7=0': (fc1, fc2, z(w7), skip, bc), 'fc2=0,w7=0,skip=0,b=None': (fc1, z(fc2), z(w7), torch.zeros_like(skip), None)}.items():
        ref, saved = ops._tail_forward(u, sk, a1, a2, a7, wc, bb)
        with torch.no_grad():
            got = ops.attention_tail(u, sk, a1, a2, a7, wc, bb)
        d = (ref - got).abs()
        print('%-28s max diff %.3e  n_diff %d of %d  max|ref| %.3f' % (name, float(d.max()), int((d > 0).sum()), d.numel(), float(ref.abs().max())))
        if float(d.max()) > 0:
            idx = (d == d.max()).nonzero()[0].tolist()
            print('    worst at', idx, float(ref[tuple(idx)]), float(got[tuple(idx)]))
