"""Round 4: board power and shader clock while the RAB weight-gradient pair launch loops (default, per-lane addressing, pipelined),
with the persistent fprop kernel beside it for scale.  Energy per launch = mean W x us."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
B = 32
CL = torch.channels_last
mk = lambda n, c, h, w: torch.randn(n, c, h, w, device=dev).contiguous(memory_format=CL)
x64 = [mk(B, 64, 54, 54) for _ in range(2)]
t256 = [mk(B, 256, 54, 54) for _ in range(2)]
w1 = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.05)
b1 = torch.randn(256, device=dev) * 0.01
items = [(x, dy, torch.zeros(256, 64, 3, 3, device=dev), torch.zeros(256, device=dev), 1, 1) for x, dy in zip(x64, t256)]
fl = 2.0 * B * 54 * 54 * 256 * 64 * 9
DUR = float(os.environ.get('DUR', '1.5'))


def loop(name, fn, flops, setup):
    setup()
    for _ in range(100): fn()
    torch.cuda.synchronize()
    ps = bench.PowerSampler(0)
    ps.__enter__()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < DUR:
        for _ in range(300): fn()
        torch.cuda.synchronize(); n += 300
    dt = time.perf_counter() - t0
    ps.__exit__()
    sm = ps.summary() or {}
    us = dt / n * 1e6
    wm = sm.get('watts_mean') or 0.0
    print('%-52s %6.1f us  %5.0f W  %5.0f MHz  %6.1f mJ/launch  %.2f pJ/flop-eq  frac %.3f' % (
        name, us, wm, sm.get('sclk_mhz_mean') or 0.0, wm * us * 1e-3, wm * us * 1e-6 / flops * 1e12, flops / us / 1e6 / 833.3), flush=True)


def cfg(addr, pipe, target):
    def f():
        lib.srhip_debug_set(8, addr); lib.srhip_debug_set(9, pipe); lib.srhip_debug_set(1, target)
    return f


with ops.conv_math('bf16x3'):
    loop('fprop 64->256 persistent', lambda: ops.conv2d_fwd_raw(x64[0], w1, b1, 1, 1, 0.2), fl, cfg(1, 0, 0))
    loop('wgrad pair 64->256 (default: scalar offsets)', lambda: ops.conv2d_wgrad_multi_raw(items), 2 * fl, cfg(1, 0, 0))
    loop('wgrad pair, per-lane offsets (rounds 1-3)', lambda: ops.conv2d_wgrad_multi_raw(items), 2 * fl, cfg(0, 0, 0))
    loop('wgrad pair, pipelined, 480 blocks', lambda: ops.conv2d_wgrad_multi_raw(items), 2 * fl, cfg(1, 1, 480))
    loop('wgrad single 64->256', lambda: ops.conv2d_wgrad_raw(x64[0], t256[0], (256, 64, 3, 3), 1, 1, True), fl, cfg(1, 0, 0))
cfg(1, 0, 0)()
