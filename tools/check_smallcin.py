import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
torch.manual_seed(0)
for B, h, w in ((32, 216, 216), (3, 70, 131)):
    x = torch.randn(B, 3, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, 64, h, w, device=dev).contiguous(memory_format=torch.channels_last)
    ref = torch.nn.grad.conv2d_weight(x[:3].double(), (64, 3, 3, 3), dy[:3].double(), padding=1) if B > 3 else torch.nn.grad.conv2d_weight(x.double(), (64, 3, 3, 3), dy.double(), padding=1)
    dw, db = ops.conv2d_wgrad_raw(x if B <= 3 else x[:3].contiguous(memory_format=torch.channels_last), dy if B <= 3 else dy[:3].contiguous(memory_format=torch.channels_last), (64, 3, 3, 3), 1, 1, True)
    print('B=%d %dx%d  3->64 wgrad err %.2e' % (min(B, 3), h, w, float((dw.double() - ref).abs().max() / ref.abs().max())))
    t = timeit(lambda: ops.conv2d_wgrad_raw(x, dy, (64, 3, 3, 3), 1, 1, True))
    print('   B=%d time %.3f ms' % (B, t), flush=True)
# 3 -> 64 forward (head conv) at full size
x = torch.randn(32, 3, 216, 216, device=dev).contiguous(memory_format=torch.channels_last)
w0 = torch.nn.Parameter(torch.randn(64, 3, 3, 3, device=dev) * 0.1); b0 = torch.randn(64, device=dev)
ref = torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x[:2].double(), w0.double(), b0.double(), padding=1), 0.2)
y = ops.conv2d_fwd_raw(x, w0, b0, 1, 1, 0.2)
print('3->64 fwd err %.2e  time %.3f ms' % (float((y[:2].double() - ref).abs().max() / ref.abs().max()), timeit(lambda: ops.conv2d_fwd_raw(x, w0, b0, 1, 1, 0.2))))
xr = torch.randn(2, 3, 259, 257, device=dev).contiguous(memory_format=torch.channels_last)
ref = torch.nn.functional.conv2d(xr.double(), w0.double(), None, padding=1)
print('ragged err %.2e' % float((ops.conv2d_fwd_raw(xr, w0, None, 1, 1).double() - ref).abs().max() / ref.abs().max()))
