"""Round 4: the inference tail -- training-mode launches (pooling partials, MLP, SLAM pool, 7x7 conv + the 1x1 conv) against
srhip_attn_tail_eval (pooling partials + one fused kernel) at the inference batch; back-to-back launches, HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops
DEV = torch.device('cuda:0')
B = int(os.environ.get('B', '16'))
g = torch.Generator().manual_seed(1)
c, h, w = 64, 54, 54
u = torch.randn(B, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
skip = torch.randn(B, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
P = lambda t: torch.nn.Parameter(t.to(DEV))
fc1, fc2 = P(torch.randn(4, c, 1, 1, generator=g) * 0.3), P(torch.randn(c, 4, 1, 1, generator=g) * 0.3)
w7, wc, bc = P(torch.randn(1, 2, 7, 7, generator=g) * 0.2), P(torch.randn(c, c, 1, 1, generator=g) * 0.1), P(torch.randn(c, generator=g))
fns = {'training-mode launches': lambda: ops._tail_forward(u, skip, fc1, fc2, w7, wc, bc), 'fused inference tail': lambda: ops._tail_forward_eval(u, skip, fc1, fc2, w7, wc, bc)}
from sradsgan_amd import _hip
DBG = [int(v) for v in os.environ.get('DBG', '0').split(',')]
with ops.conv_math('bf16x3'), torch.no_grad():
  for dbg in DBG:
    _hip.lib().srhip_debug_set(7, dbg)
    print('dbg', dbg)
    for rnd in range(2):
        for name, fn in fns.items():
            for _ in range(20): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(True), torch.cuda.Event(True)
            s.record()
            for _ in range(200): fn()
            e.record(); torch.cuda.synchronize()
            print('B=%d %-24s %.1f us per tail' % (B, name, s.elapsed_time(e) / 200 * 1e3), flush=True)
_hip.lib().srhip_debug_set(7, 0)
