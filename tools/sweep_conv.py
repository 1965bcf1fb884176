"""Tile-configuration sweep of the fast conv kernel (srhip_debug_set key 0) at the bench shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
B = 32
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
shapes = [('conv1 64->256 @54', 64, 54, 256, 1), ('conv2 256->64 @54', 256, 54, 64, 1), ('up 64->256 @108', 64, 108, 256, 1),
          ('vgg 64->64 @216', 64, 216, 64, 1), ('D 128->256 @54', 128, 54, 256, 1), ('D 256->512 @27', 256, 27, 512, 1)]
names = {0: 'heuristic', 1: '128x128 bk32', 2: '64x128 bk16', 3: '64x128 bk32', 4: '256x128 8w bk16', 5: '128x128 bk16',
         6: '128x64 bk32', 7: '64x64 bk16', 8: '128x64(4x1) bk32'}
for name, cin, h, cout, st in shapes:
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    b = torch.zeros(cout, device=dev)
    fl = 2.0 * B * h * h * cout * cin * 9
    ref = None
    for cfg in range(9):
        if cout >= 128 and cfg in (6, 8): continue
        if cout < 128 and cfg in (1, 2, 3, 4, 5): continue
        lib.srhip_debug_set(0, cfg)
        y = ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)
        if ref is None: ref = y
        err = float((y - ref).abs().max())
        t = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2))
        print('%-20s cfg %d %-18s %.3f ms %6.1f TF  maxdiff %.1e' % (name, cfg, names[cfg], t, fl / t / 1e9, err), flush=True)
lib.srhip_debug_set(0, 0)
