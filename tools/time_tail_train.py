"""Round 4: forward + backward of one attention tail (training mode) at the bench batch, for a per-kernel profile
(rocprofv3 --kernel-trace --stats -- python3 tools/time_tail_train.py; tools/kstats.py prints the table)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops
DEV = torch.device('cuda:0')
B = int(os.environ.get('B', '32'))
g = torch.Generator().manual_seed(1)
c, h, w = 64, 54, 54
u = torch.randn(B, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
skip = torch.randn(B, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
gout = torch.randn(B, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
P = lambda t: torch.nn.Parameter(t.to(DEV))
fc1, fc2 = P(torch.randn(4, c, 1, 1, generator=g) * 0.3), P(torch.randn(c, 4, 1, 1, generator=g) * 0.3)
w7, wc, bc = P(torch.randn(1, 2, 7, 7, generator=g) * 0.2), P(torch.randn(c, c, 1, 1, generator=g) * 0.1), P(torch.randn(c, generator=g))
with ops.conv_math('bf16x3'):
    for it in range(60):
        out, saved = ops._tail_forward(u, skip, fc1, fc2, w7, wc, bc)
        ops._tail_backward(gout, u, fc1, fc2, w7, wc, bc, saved, True)
    torch.cuda.synchronize()
