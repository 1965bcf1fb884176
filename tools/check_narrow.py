import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0'); lib = _hip.lib()
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
torch.manual_seed(0)
B, h = 32, 216
x = torch.randn(B, 64, h, h, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(3, 64, 3, 3, device=dev) * 0.05)
b = torch.randn(3, device=dev)
w0 = torch.nn.Parameter(torch.randn(64, 3, 3, 3, device=dev) * 0.05)      # D conv0: dgrad has 64 source channels, 3 destination
dy = torch.randn(B, 64, h, h, device=dev).contiguous(memory_format=torch.channels_last)
ref = torch.nn.functional.conv2d(x[:2].double(), w.double(), b.double(), padding=1)
refd = torch.nn.grad.conv2d_input((2, 3, h, h), w0.double(), dy[:2].double(), padding=1)
for mode in (0, 1):
    lib.srhip_set_conv_math(mode)
    y = ops.conv2d_fwd_raw(x, w, b, 1, 1)
    d = ops.conv2d_dgrad_raw(dy, w0, (B, 3, h, h), 1, 1)
    e1 = float((y[:2].double() - ref).abs().max() / ref.abs().max()); e2 = float((d[:2].double() - refd).abs().max() / refd.abs().max())
    t1 = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1)); t2 = timeit(lambda: ops.conv2d_dgrad_raw(dy, w0, (B, 3, h, h), 1, 1))
    print('mode %d  64->3 fwd %.3f ms err %.2e   3<-64 dgrad %.3f ms err %.2e' % (mode, t1, e1, t2, e2), flush=True)
