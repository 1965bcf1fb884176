"""Experiment variants of conv_patch_kernel<128, bias+lrelu> (srhip_debug_set(3, bits << 12)) on RAB conv1 at the bench shape:
interleaved timing rounds (median of per-round means) and bit-exactness against the shipped variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
B = int(os.environ.get('B', '32'))
x = torch.randn(B, 64, 54, 54, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05)
b = torch.randn(256, device=dev) * 0.01
fl = 2.0 * B * 54 * 54 * 256 * 64 * 9
names = {16: 'prefetched fragments (2 blocks per CU)', 20: 'prefetched fragments + buffer DMA', 0: 'shipped', 1: 'setprio', 2: 'no convert (timing only)', 4: 'buffer DMA', 5: 'buffer DMA + setprio',
         6: 'buffer DMA, no convert (timing only)', 8: 'no stores (timing only)', 15: 'all'}
variants = [int(v) for v in os.environ.get('VARS', '0,1,2,4,5,6,8').split(',')]
fn = lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)
lib.srhip_debug_set(3, 0)
ref = fn().clone()
for v in variants:
    lib.srhip_debug_set(3, v << 12)
    y = fn()
    print('var %2d %-40s max |diff| vs shipped %.3e' % (v, names.get(v, '?'), float((y - ref).abs().max())), flush=True)
times = {v: [] for v in variants}
for rnd in range(7):
    for v in variants:
        lib.srhip_debug_set(3, v << 12)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(30): fn()
        e.record(); torch.cuda.synchronize()
        times[v].append(s.elapsed_time(e) / 30)
lib.srhip_debug_set(3, 0)
for v in variants:
    t = sorted(times[v])
    print('var %2d %-40s median %.1f us  min %.1f us  %.0f TFLOP/s-equivalent' % (v, names.get(v, '?'), t[len(t) // 2] * 1e3, t[0] * 1e3, fl / t[len(t) // 2] / 1e9))
