import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0'); lib = _hip.lib()
def timeit(fn, iters=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
lib.srhip_set_conv_math(1)
for name, B, cin, h, cout in (('conv1@54 B32', 32, 64, 54, 256), ('conv2@54 B32', 32, 256, 54, 64), ('vgg256@54', 32, 256, 54, 256), ('vgg64@216', 32, 64, 216, 64)):
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (cin * 9)) ** 0.5)
    b = torch.randn(cout, device=dev) * 0.1
    fl = 2.0 * B * h * h * cout * cin * 9
    for abl, label in ((0, 'full'), (0x1000, 'no B DMA'), (0x2000, 'no vmcnt wait'), (0x3000, 'no B DMA, no wait'), (0x4000, 'no barrier'), (0x7000, 'none of them')):
        lib.srhip_debug_set(3, abl)
        t = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2))
        print('%-14s %-20s %.3f ms %6.1f TF-equiv' % (name, label, t, fl / t / 1e9), flush=True)
lib.srhip_debug_set(3, 0)
