"""SGAM forward + backward at the x4 (54 x 54, B = 32) and x2 (108 x 108, B = 8) tiles: split-bf16 kernels against the exact-fp32 ones
(srhip_debug_set(4, 1)), HIP-event times of the op through the autograd wrappers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import _hip, ops
dev = torch.device('cuda:0')
for b, hw in ((32, 54), (8, 108)):
    g = torch.Generator().manual_seed(1)
    x, v, dy = (torch.randn(b, 64, hw, hw, generator=g).to(dev).contiguous(memory_format=torch.channels_last) for _ in range(3))
    q, k = (torch.randn(b, 8, hw, hw, generator=g).to(dev).contiguous(memory_format=torch.channels_last) * 1.5 for _ in range(2))
    gamma = torch.nn.Parameter(torch.tensor([0.5], device=dev))
    for exact in (1, 0, 1, 0):
        _hip.lib().srhip_debug_set(4, exact)
        xs = [t.clone().requires_grad_(True) for t in (x, q, k, v)]
        def run():
            y = ops.sgam(xs[0], xs[1], xs[2], xs[3], gamma)
            y.backward(dy)
        for _ in range(3): run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize()
        print('B=%d %dx%d  %s: %.3f ms fwd+bwd' % (b, hw, hw, 'exact fp32 MFMA' if exact else 'split-bf16     ', s.elapsed_time(e) / 10))
_hip.lib().srhip_debug_set(4, 0)
