import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0'); lib = _hip.lib(); B = 32
def timeit(fn, iters=20):
    for _ in range(8): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
lib.srhip_set_conv_math(1)
for name, cin, h, cout in (('conv1 64->256 @54', 64, 54, 256), ('conv2 256->64 @54', 256, 54, 64), ('up 64->256 @108', 64, 108, 256), ('vgg 256->256 @54', 256, 54, 256), ('D 256->512 @27', 256, 27, 512)):
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * h * h * cout * cin * 9
    for target in (0, 256, 384, 512, 768, 1024):
        lib.srhip_debug_set(1, target)
        t = timeit(lambda: ops.conv2d_wgrad_raw(x, dy, (cout, cin, 3, 3), 1, 1, True))
        print('%-20s target %4d  %.3f ms %6.1f TF-eq' % (name, target, t, fl / t / 1e9), flush=True)
lib.srhip_debug_set(1, 0)
