"""RAB conv1 fprop (3x3, 64 -> 256 @54x54, bias + LeakyReLU) at growing batch = growing number of block rounds on the 768 block
slots of the chip, with 128-wide (shipping) and 64-wide N tiles (srhip_debug_set(0, 24)): does the kernel's efficiency depend
on how many rounds a launch has?  Sustained loop of ~0.5 s per point."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import _hip, ops
dev = torch.device('cuda:0')
w = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.02)
b = torch.randn(256, device=dev) * 0.01
w2 = torch.nn.Parameter(torch.randn(64, 256, 3, 3, device=dev) * 0.02)
for cin, cout, wt in ((64, 256, w), (256, 64, w2)):
    for B in (16, 32, 64, 128):
        x = torch.randn(B, cin, 54, 54, device=dev).contiguous(memory_format=torch.channels_last)
        for cfg in (0, 24):
            if cout == 64 and cfg == 24:
                continue
            _hip.lib().srhip_debug_set(0, cfg)
            fn = lambda: ops.conv2d_fwd_raw(x, wt, b[:cout].contiguous(), 1, 1, 0.2)
            for _ in range(20): fn()
            torch.cuda.synchronize()
            n = 0; s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            t_end = time.perf_counter() + 0.5
            s.record()
            while time.perf_counter() < t_end:
                for _ in range(50): fn()
                n += 50
                torch.cuda.synchronize()
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / n * 1e3
            fl = 2.0 * B * 54 * 54 * cout * cin * 9
            print('%3d->%3d  B=%3d  N tile %3d: %7.1f us  %6.1f TFLOP/s-equivalent  (%.3f of 833)' % (cin, cout, B, 64 if (cfg == 24 or cout == 64) else 128, us, fl / us / 1e6, fl / us / 1e6 / 833.3))
_hip.lib().srhip_debug_set(0, 0)
