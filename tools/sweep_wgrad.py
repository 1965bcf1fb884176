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
shapes = [('conv1 64->256 @54', 64, 54, 256, 3), ('conv2 256->64 @54', 256, 54, 64, 3), ('1x1 64->64 @54', 64, 54, 64, 1), ('up 64->256 @108', 64, 108, 256, 3),
          ('vgg 64->64 @216', 64, 216, 64, 3), ('D 128->256 @54', 128, 54, 256, 3), ('D 256->512 @27', 256, 27, 512, 3)]
for name, cin, h, cout, k in shapes:
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * h * h * cout * cin * k * k
    ref = None
    for cfg in (0, 5, 0, 5):
        lib.srhip_debug_set(1, cfg)
        dw, db = ops.conv2d_wgrad_raw(x, dy, (cout, cin, k, k), 1, k // 2, True)
        if ref is None: ref = dw
        err = float((dw - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: ops.conv2d_wgrad_raw(x, dy, (cout, cin, k, k), 1, k // 2, True))
        print('%-20s wgrad cfg=%d %-9s %.3f ms %6.1f TF  rel diff %.1e' % (name, cfg, {0: 'default', 5: '256-wide'}[cfg], t, fl / t / 1e9, err), flush=True)
lib.srhip_debug_set(1, 0)
