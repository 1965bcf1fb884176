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
for name, B, cin, h, cout in (('conv1@54 B128', 128, 64, 54, 256), ('conv2@54 B128', 128, 256, 54, 64)):
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    b = torch.zeros(cout, device=dev)
    fl = 2.0 * B * h * h * cout * cin * 9
    for dyn in (0, 50 * 1024):
        for abl, label in ((0, 'full'), (0x100, 'no loads'), (0x200, 'no barrier'), (0x300, 'no loads, no barrier')):
            lib.srhip_debug_set(2, dyn); lib.srhip_debug_set(3, abl)
            t = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2))
            print('%-14s %s %-22s %.3f ms %6.1f TF' % (name, '1blk/CU' if dyn else 'max occ', label, t, fl / t / 1e9), flush=True)
lib.srhip_debug_set(2, 0); lib.srhip_debug_set(3, 0)
