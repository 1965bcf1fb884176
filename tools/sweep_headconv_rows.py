import os, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0'); lib = _hip.lib()
x3 = torch.randn(32, 3, 216, 216, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(64, 3, 3, 3, device=dev) * 0.1); b = torch.zeros(64, device=dev)
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
ref = None
for rows in (1, 4, 1, 4, 2, 3, 6, 1, 4, 1, 4):
    lib.srhip_debug_set(18, rows)
    y = ops.conv2d_fwd_raw(x3, w, b, 1, 1, 0.2)
    if ref is None: ref = y.clone()
    print('rows per block %2d: %6.1f us  identical %s' % (rows, timed(lambda: ops.conv2d_fwd_raw(x3, w, b, 1, 1, 0.2)), torch.equal(y, ref)), flush=True)
