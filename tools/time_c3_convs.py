"""The 3-channel convs at B = 32, 216 x 216 (D's head conv 3 -> 64 and the generator's last conv 64 -> 3): forward, data gradient, weight
gradient, isolated back-to-back launches; the HBM floor of each is one pass over the 382 MB 64-channel tensor (~80 us at 4.8 TB/s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops
dev = torch.device('cuda:0')
B, S = 32, 216


def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


x3 = torch.randn(B, 3, S, S, device=dev).contiguous(memory_format=torch.channels_last)
x64 = torch.randn(B, 64, S, S, device=dev).contiguous(memory_format=torch.channels_last)
w_head = torch.nn.Parameter(torch.randn(64, 3, 3, 3, device=dev) * 0.1); b_head = torch.zeros(64, device=dev)
w_tail = torch.nn.Parameter(torch.randn(3, 64, 3, 3, device=dev) * 0.05); b_tail = torch.zeros(3, device=dev)
print('3 -> 64 fwd (+bias+lrelu)   %6.1f us' % timed(lambda: ops.conv2d_fwd_raw(x3, w_head, b_head, 1, 1, 0.2)))
print('3 -> 64 dgrad (64 -> 3)     %6.1f us' % timed(lambda: ops.conv2d_dgrad_raw(x64, w_head, (B, 3, S, S), 1, 1)))
print('3 -> 64 wgrad (+bias)       %6.1f us' % timed(lambda: ops.conv2d_wgrad_raw(x3, x64, (64, 3, 3, 3), 1, 1, True)))
print('64 -> 3 fwd (+bias)         %6.1f us' % timed(lambda: ops.conv2d_fwd_raw(x64, w_tail, b_tail, 1, 1)))
print('64 -> 3 dgrad (3 -> 64)     %6.1f us' % timed(lambda: ops.conv2d_dgrad_raw(x3, w_tail, (B, 64, S, S), 1, 1)))
print('64 -> 3 wgrad (+bias)       %6.1f us' % timed(lambda: ops.conv2d_wgrad_raw(x64, x3, (3, 64, 3, 3), 1, 1, True)))
