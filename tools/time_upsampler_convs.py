import os, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from sradsgan_amd import ops
dev = torch.device('cuda:0')
def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for h in (54, 108):
    x = torch.randn(32, 64, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(32, 256, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05); b = torch.zeros(256, device=dev)
    gf = 2.0 * 32 * h * h * 64 * 256 * 9 / 1e9
    tf = timed(lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1)); td = timed(lambda: ops.conv2d_dgrad_raw(dy, w, (32, 64, h, h), 1, 1)); tw = timed(lambda: ops.conv2d_wgrad_raw(x, dy, (256, 64, 3, 3), 1, 1, True))
    xp, dp = ops.pp_from_f32(x), ops.pp_from_f32(dy)
    gw, gb = torch.zeros_like(w), torch.zeros(256, device=dev)
    tp = timed(lambda: ops.conv2d_wgrad_pp_raw([(xp, dp, gw, gb)]))
    tc = timed(lambda: ops.pp_from_f32(dy, out=dp))
    print('64 -> 256 @%d (%.0f GF): fwd %.1f us (%.0f TF/s)  dgrad %.1f (%.0f)  wgrad row-tap %.1f (%.0f)  wgrad flat on planes %.1f (%.0f)  pp_from_f32(dy) %.1f' % (h, gf, tf, gf/tf*1e3, td, gf/td*1e3, tw, gf/tw*1e3, tp, gf/tp*1e3, tc))
