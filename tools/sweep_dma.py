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
for name, B, cin, h, cout in (('conv1@54 B32', 32, 64, 54, 256), ('conv1@54 B128', 128, 64, 54, 256), ('conv2@54 B32', 32, 256, 54, 64), ('conv2@54 B128', 128, 256, 54, 64),
                              ('vgg64@216', 32, 64, 216, 64), ('D128->256@54', 32, 128, 54, 256), ('D256->512@27', 32, 256, 27, 512)):
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    b = torch.randn(cout, device=dev)
    fl = 2.0 * B * h * h * cout * cin * 9
    lib.srhip_debug_set(0, 20)
    ref = ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)
    for cfg, label in ((20, 'reg-staged'), (0, 'lds-dma'), (20, 'reg-staged'), (0, 'lds-dma')):
        lib.srhip_debug_set(0, cfg)
        y = ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)
        err = float((y - ref).abs().max())
        t = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2))
        print('%-14s %-10s %.3f ms %6.1f TF  maxdiff %.1e' % (name, label, t, fl / t / 1e9, err), flush=True)
lib.srhip_debug_set(0, 0)
