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
for name, B, cin, h, cout in (('conv1@54 B32', 32, 64, 54, 256), ('conv2@54 B32', 32, 256, 54, 64),
                              ('vgg64@216', 32, 64, 216, 64), ('vgg128@108', 32, 128, 108, 128), ('vgg256@54', 32, 256, 54, 256),
                              ('D128->256@54', 32, 128, 54, 256), ('D256->512@27', 32, 256, 27, 512), ('up64->256@108', 32, 64, 108, 256)):
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (cin * 9)) ** 0.5)
    b = torch.randn(cout, device=dev) * 0.1
    fl = 2.0 * B * h * h * cout * cin * 9
    ref64 = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref64 = torch.nn.functional.leaky_relu(ref64, 0.2)
    scale = float(ref64.abs().max())
    for mode, label in ((0, 'fp32-mfma'), (1, 'bf16x3')):
        lib.srhip_set_conv_math(mode)
        y = ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)
        err = float((y.double() - ref64).abs().max()) / scale
        t = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2))
        print('%-14s %-16s %.3f ms %6.1f TF-equiv  max|err|/max|y| vs fp64 %.2e' % (name, label, t, fl / t / 1e9, err), flush=True)
    dy = torch.randn(B, cout, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    refd = torch.nn.grad.conv2d_input(x.shape, w.double(), dy.double(), padding=1)
    sc = float(refd.abs().max())
    for mode, label in ((0, 'fp32-mfma'), (1, 'bf16x3')):
        lib.srhip_set_conv_math(mode)
        dx = ops.conv2d_dgrad_raw(dy, w, x.shape, 1, 1)
        err = float((dx.double() - refd).abs().max()) / sc
        t = timeit(lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, 1, 1))
        print('%-14s dgrad %-10s %.3f ms %6.1f TF-equiv  err %.2e' % (name, label, t, fl / t / 1e9, err), flush=True)
lib.srhip_set_conv_math(0)
