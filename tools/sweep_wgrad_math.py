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
shapes = [('conv1 64->256 @54', 64, 54, 256, 3, 1), ('conv2 256->64 @54', 256, 54, 64, 3, 1), ('1x1 64->64 @54', 64, 54, 64, 1, 1), ('up 64->256 @108', 64, 108, 256, 3, 1),
          ('vgg 64->64 @216', 64, 216, 64, 3, 1), ('vgg 256->256 @54', 256, 54, 256, 3, 1), ('D 128->256 @54', 128, 54, 256, 3, 1), ('D 256->512 @27', 256, 27, 512, 3, 1),
          ('D s2 64->64 @216', 64, 216, 64, 3, 2), ('D s2 512->512 @14', 512, 14, 512, 3, 2)]
for name, cin, h, cout, k, stride in shapes:
    ho = (h + 2 * (k // 2) - k) // stride + 1
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, ho, ho, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * ho * ho * cout * cin * k * k
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dy.double(), stride=stride, padding=k // 2)
    refb = dy.double().sum((0, 2, 3))
    for mode in (0, 1):
        lib.srhip_set_conv_math(mode)
        dw, db = ops.conv2d_wgrad_raw(x, dy, (cout, cin, k, k), stride, k // 2, True)
        err = float((dw.double() - ref).abs().max() / ref.abs().max())
        errb = float((db.double() - refb).abs().max() / refb.abs().max())
        t = timeit(lambda: ops.conv2d_wgrad_raw(x, dy, (cout, cin, k, k), stride, k // 2, True))
        print('%-20s wgrad %-7s %.3f ms %6.1f TF-equiv  err vs fp64 %.2e  bias err %.1e' % (name, ('fp32', 'bf16x3')[mode], t, fl / t / 1e9, err, errb), flush=True)
lib.srhip_set_conv_math(0)
