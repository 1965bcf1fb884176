"""Round 4: timing-only ablations of conv_patch_pers_kernel<128, bias+lrelu> on RAB conv1 (64 -> 256 @ 54x54, B = 32) with board
power and shader clock sampled while each variant loops (srhip_debug_set(6, bits); results of bits != 0 are wrong on purpose).
Energy per launch = mean W x us per launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
B = int(os.environ.get('B', '32'))
x = torch.randn(B, 64, 54, 54, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05)
b = torch.randn(256, device=dev) * 0.01
fl = 2.0 * B * 54 * 54 * 256 * 64 * 9
names = {64: 'direct epilogue (operand roles swapped, 32-byte stores)', 128: 'B DMA issued twice', 256: 'A DMA issued twice', 0: 'full kernel', 1: 'stores dropped (OOB)', 2: 'no conversion', 4: 'no MFMA', 12: 'no MFMA, no fragment reads', 16: 'no B DMA',
         32: 'no epilogue', 30: 'skeleton: A DMA + barriers + epilogue', 26: 'MFMA + A DMA + barriers + epilogue', 63: 'A DMA + barriers only',
         -1: 'one-tile kernel (rounds 1-3)'}
variants = [int(v) for v in os.environ.get('VARS', '0,-1,1,32,2,16,4,12,26,30,63,0').split(',')]
DUR = float(os.environ.get('DUR', '1.5'))
fn = lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)
with ops.conv_math('bf16x3'):
    for v in variants:
        lib.srhip_debug_set(5, -1 if v == -1 else 0)
        lib.srhip_debug_set(6, max(v, 0))
        for _ in range(200): fn()
        torch.cuda.synchronize()
        ps = bench.PowerSampler(0)
        ps.__enter__()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < DUR:
            for _ in range(500): fn()
            torch.cuda.synchronize(); n += 500
        dt = time.perf_counter() - t0
        ps.__exit__()
        sm = ps.summary() or {}
        us = dt / n * 1e6
        wm = sm.get('watts_mean') or 0.0
        print('abl %3d %-40s %6.1f us  %5.0f W  %5.0f MHz  %6.1f mJ/launch  (%.0f TFLOP/s-eq if it were the conv)' % (
            v, names.get(v, '?'), us, wm, sm.get('sclk_mhz_mean') or 0.0, wm * us * 1e-3, fl / us / 1e6), flush=True)
lib.srhip_debug_set(5, 0)
lib.srhip_debug_set(6, 0)
