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
torch.manual_seed(0)
# correctness: patch kernel vs LDS-DMA kernel, bitwise
for name, B, cin, h, w_, cout in (('small ragged', 2, 64, 11, 13, 128), ('small 64', 3, 128, 9, 20, 64), ('conv1@54', 4, 64, 54, 54, 256), ('conv2@54', 4, 256, 54, 54, 64), ('27x27', 2, 256, 27, 27, 512), ('216', 1, 64, 216, 216, 64)):
    x = torch.randn(B, cin, h, w_, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * (2.0 / (cin * 9)) ** 0.5)
    b = torch.randn(cout, device=dev) * 0.1
    dy = torch.randn(B, cout, h, w_, device=dev).contiguous(memory_format=torch.channels_last)
    r = torch.randn(B, cin, h, w_, device=dev).contiguous(memory_format=torch.channels_last)
    lib.srhip_debug_set(0, -1)       # force the DMA kernel (patch off: cfg -1 != -2 and tiles < 512)
    y0 = ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2); d0 = ops.conv2d_dgrad_raw(dy, w, tuple(x.shape), 1, 1, r, x, 0.2)
    lib.srhip_debug_set(0, -2)       # force the patch kernel
    y1 = ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2); d1 = ops.conv2d_dgrad_raw(dy, w, tuple(x.shape), 1, 1, r, x, 0.2)
    ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1), 0.2)
    print('%-14s fwd max|patch-dma| %.3e  dgrad %.3e   patch vs fp64 %.2e' % (name, float((y1 - y0).abs().max()), float((d1 - d0).abs().max()),
          float((y1.double() - ref).abs().max() / ref.abs().max())), flush=True)
# timing
for name, B, cin, h, cout in (('conv1@54 B32', 32, 64, 54, 256), ('conv2@54 B32', 32, 256, 54, 64), ('vgg64@216', 32, 64, 216, 64), ('vgg128@108', 32, 128, 108, 128),
                              ('vgg256@54', 32, 256, 54, 256), ('D256->512@27', 32, 256, 27, 512), ('up64->256@108', 32, 64, 108, 256)):
    x = torch.randn(B, cin, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
    b = torch.randn(cout, device=dev)
    dy = torch.randn(B, cout, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * h * h * cout * cin * 9
    for cfg, label in ((21, 'lds-dma'), (0, 'patch'), (21, 'lds-dma'), (0, 'patch')):
        lib.srhip_debug_set(0, cfg)
        t = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2))
        t2 = timeit(lambda: ops.conv2d_dgrad_raw(dy, w, tuple(x.shape), 1, 1))
        print('%-14s %-8s fwd %.3f ms %6.1f TF-eq   dgrad %.3f ms %6.1f TF-eq' % (name, label, t, fl / t / 1e9, t2, fl / t2 / 1e9), flush=True)
lib.srhip_debug_set(0, 0)
