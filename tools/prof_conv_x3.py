"""bf16x3 conv fwd on two shapes, for rocprofv3 --pmc / --kernel-trace runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops
dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for cin, cout in ((256, 256), (64, 256)):
    x = torch.randn(32, cin, 54, 54, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.02)
    b = torch.randn(cout, device=dev) * 0.01
    dy = torch.randn(32, cout, 54, 54, device=dev).contiguous(memory_format=torch.channels_last)
    for _ in range(iters):
        y = ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)
        dw, db = ops.conv2d_wgrad_raw(x, dy, tuple(w.shape), 1, 1, True)
torch.cuda.synchronize()
print('done')
