"""Standalone times of the attention-tail pieces of one RAB at the bench shape (B=32, 64 ch, 54x54)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
def timeit(fn, iters=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
B, h = 32, 54
u = torch.randn(B, 64, h, h, device=dev).contiguous(memory_format=torch.channels_last)
skip = torch.randn_like(u).contiguous(memory_format=torch.channels_last)
g = torch.randn_like(u).contiguous(memory_format=torch.channels_last)
fc1 = torch.nn.Parameter(torch.randn(4, 64, 1, 1, device=dev) * 0.1); fc2 = torch.nn.Parameter(torch.randn(64, 4, 1, 1, device=dev) * 0.1)
w7 = torch.nn.Parameter(torch.randn(1, 2, 7, 7, device=dev) * 0.1); wc = torch.nn.Parameter(torch.randn(64, 64, 1, 1, device=dev) * 0.1); bc = torch.randn(64, device=dev)
out, saved = ops._tail_forward(u, skip, fc1, fc2, w7, wc, bc)
print('tail forward (pool, mlp, slam pool, conv7, 1x1 conv)   %.1f us' % timeit(lambda: ops._tail_forward(u, skip, fc1, fc2, w7, wc, bc)))
print('tail backward, params skipped                           %.1f us' % timeit(lambda: ops._tail_backward(g, u, fc1, fc2, w7, wc, bc, saved, True, True)))
print('tail backward, with parameter gradients                 %.1f us' % timeit(lambda: ops._tail_backward(g, u, fc1, fc2, w7, wc, bc, saved, True, False)))
print('1x1 dgrad alone                                         %.1f us' % timeit(lambda: ops.conv2d_dgrad_raw(g, wc, tuple(u.shape), 1, 0)))
print('1x1 fwd +bias alone                                     %.1f us' % timeit(lambda: ops.conv2d_fwd_raw(u, wc, bc, 1, 0)))
print('1x1 wgrad (scaled operand) alone                        %.1f us' % timeit(lambda: ops.conv2d_wgrad_raw(u, g, (64, 64, 1, 1), 1, 0, True, saved[6], saved[3])))
print('1x1 wgrad plain alone                                   %.1f us' % timeit(lambda: ops.conv2d_wgrad_raw(u, g, (64, 64, 1, 1), 1, 0, True)))
