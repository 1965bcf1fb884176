import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, B, cin, h, cout in (('conv1@54', 32, 64, 54, 256), ('conv1@54 B=128', 128, 64, 54, 256), ('vgg 64->64@216', 32, 64, 216, 64)):
    for fill in ('randn', 'zeros'):
        x = (torch.randn if fill == 'randn' else torch.zeros)(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
        w = torch.nn.Parameter((torch.randn if fill == 'randn' else torch.zeros)(cout, cin, 3, 3, device=dev) * 0.05)
        b = torch.zeros(cout, device=dev)
        t = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2))
        fl = 2.0 * B * h * h * cout * cin * 9
        print('%-18s %-6s %.3f ms %.1f TF' % (name, fill, t, fl / t / 1e9), flush=True)
