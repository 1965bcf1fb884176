"""Runs the dominant kernel (RAB conv1/conv2 fwd, dgrad, wgrad at the bench shape) a few times; meant
to be run under rocprofv3 (--kernel-trace --stats, or --pmc ...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = torch.randn(B, 64, 54, 54, device=dev).contiguous(memory_format=torch.channels_last)
w1 = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.02)
b1 = torch.randn(256, device=dev) * 0.01
w2 = torch.nn.Parameter(torch.randn(64, 256, 3, 3, device=dev) * 0.02)
b2 = torch.randn(64, device=dev) * 0.01
for _ in range(iters):
    y = ops.conv2d_fwd_raw(x, w1, b1, 1, 1, 0.2)
    z = ops.conv2d_fwd_raw(y, w2, b2, 1, 1)
    dy = ops.conv2d_dgrad_raw(z, w2, tuple(y.shape), 1, 1)
    dw, db = ops.conv2d_wgrad_raw(x, y, tuple(w1.shape), 1, 1, True)
torch.cuda.synchronize()
print('done')
