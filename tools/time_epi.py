import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0'); lib = _hip.lib()
def timeit(fn, iters=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
B, h = 32, 54
x64 = torch.randn(B, 64, h, h, device=dev).contiguous(memory_format=torch.channels_last)
x256 = torch.randn(B, 256, h, h, device=dev).contiguous(memory_format=torch.channels_last)
w1 = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05); b1 = torch.randn(256, device=dev)
w2 = torch.nn.Parameter(torch.randn(64, 256, 3, 3, device=dev) * 0.05); b2 = torch.randn(64, device=dev)
fl = 2.0 * B * h * h * 256 * 64 * 9
for name, fn in (('conv1 fwd bias+lrelu        (64->256)', lambda: ops.conv2d_fwd_raw(x64, w1, b1, 1, 1, 0.2)),
                 ('conv1 fwd plain             (64->256)', lambda: ops.conv2d_fwd_raw(x64, w1, None, 1, 1)),
                 ('conv2 dgrad plain           (64->256)', lambda: ops.conv2d_dgrad_raw(x64, w2, tuple(x256.shape), 1, 1)),
                 ('conv2 dgrad +actmask        (64->256)', lambda: ops.conv2d_dgrad_raw(x64, w2, tuple(x256.shape), 1, 1, None, x256, 0.2)),
                 ('conv2 fwd bias              (256->64)', lambda: ops.conv2d_fwd_raw(x256, w2, b2, 1, 1)),
                 ('conv1 dgrad plain           (256->64)', lambda: ops.conv2d_dgrad_raw(x256, w1, tuple(x64.shape), 1, 1)),
                 ('conv1 dgrad +residual       (256->64)', lambda: ops.conv2d_dgrad_raw(x256, w1, tuple(x64.shape), 1, 1, x64))):
    t = timeit(fn)
    print('%-42s %.3f ms %6.1f TF-eq' % (name, t, fl / t / 1e9), flush=True)
