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
