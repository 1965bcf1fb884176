"""A/B of srhip_debug_set(0, cfg) kernel selections on one conv shape: bit-exactness against cfg 0 and interleaved timing rounds.
usage: sweep_cfg.py <cfgs comma separated> [cin cout hw batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
cfgs = [int(v) for v in sys.argv[1].split(',')]
cin, cout, hw, B = (int(v) for v in (sys.argv[2:6] if len(sys.argv) >= 6 else (64, 256, 54, 32)))
x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.nn.Parameter(torch.randn(cout, cin, 3, 3, device=dev) * 0.05)
b = torch.randn(cout, device=dev) * 0.01
fl = 2.0 * B * hw * hw * cout * cin * 9
fn = lambda: ops.conv2d_fwd_raw(x, w, b, 1, 1, 0.2)
lib.srhip_debug_set(0, 0)
ref = fn().clone()
for c in cfgs:
    lib.srhip_debug_set(0, c)
    for rep in range(3):
        y = fn()
        print('cfg %3d rep %d max |diff| vs cfg 0: %.3e  finite %s' % (c, rep, float((y - ref).abs().max()), bool(torch.isfinite(y).all())), flush=True)
times = {c: [] for c in cfgs}
for rnd in range(7):
    for c in cfgs:
        lib.srhip_debug_set(0, c)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(30): fn()
        e.record(); torch.cuda.synchronize()
        times[c].append(s.elapsed_time(e) / 30)
lib.srhip_debug_set(0, 0)
for c in cfgs:
    t = sorted(times[c])
    print('cfg %3d median %.1f us  min %.1f us  %.0f TFLOP/s-equivalent' % (c, t[len(t) // 2] * 1e3, t[0] * 1e3, fl / t[len(t) // 2] / 1e9))
