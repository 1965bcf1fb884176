"""Micro-benchmark of the implicit-GEMM conv kernels at the BASELINE shapes (x4, LR 54x54)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = [  # name, cin, h, w, cout, k, stride, pad
    ('rab.conv1 64->256', 64, 54, 54, 256, 3, 1, 1),
    ('rab.conv2 256->64', 256, 54, 54, 64, 3, 1, 1),
    ('1x1 64->64', 64, 54, 54, 64, 1, 1, 0),
    ('up 64->256 @108', 64, 108, 108, 256, 3, 1, 1),
    ('tail 64->3 @216', 64, 216, 216, 3, 3, 1, 1),
    ('D c64s2 @216', 64, 216, 216, 64, 3, 2, 1),
    ('D c128s1 @108', 64, 108, 108, 128, 3, 1, 1),
    ('D c512s2 @27', 512, 27, 27, 512, 3, 2, 1),
    ('vgg 64->64 @216', 64, 216, 216, 64, 3, 1, 1),
    ('head 3->64 @216', 3, 216, 216, 64, 3, 1, 1),
]
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, cin, h, w, cout, k, st, p in shapes:
    x = torch.randn(B, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.nn.Parameter(torch.randn(cout, cin, k, k, device=dev) * 0.05)
    b = torch.randn(cout, device=dev)
    y = ops.conv2d_fwd_raw(x, wt, b, st, p, 0.2)
    dy = torch.randn_like(y)
    fl = 2.0 * y.numel() * cin * k * k
    tf = timeit(lambda: ops.conv2d_fwd_raw(x, wt, b, st, p, 0.2))
    td = timeit(lambda: ops.conv2d_dgrad_raw(dy, wt, tuple(x.shape), st, p))
    tw = timeit(lambda: ops.conv2d_wgrad_raw(x, dy, tuple(wt.shape), st, p, True))
    print('%-20s B=%d  %.2f GF | fwd %.3f ms %.1f TF | dgrad %.3f ms %.1f TF | wgrad %.3f ms %.1f TF' % (
        name, B, fl / 1e9, tf, fl / tf / 1e9, td, fl / td / 1e9, tw, fl / tw / 1e9), flush=True)
