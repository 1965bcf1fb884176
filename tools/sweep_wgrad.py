"""A/B of srhip_debug_set(1, cfg) wgrad kernel selections on one conv shape: error against fp64 and interleaved timing rounds.
usage: sweep_wgrad.py <cfgs> [cin cout hw batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
cfgs = [int(v) for v in sys.argv[1].split(',')]
cin, cout, hw, B = (int(v) for v in (sys.argv[2:6] if len(sys.argv) >= 6 else (64, 256, 54, 32)))
x = torch.randn(B, cin, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
dy = torch.randn(B, cout, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)
fl = 2.0 * B * hw * hw * cout * cin * 9
fn = lambda: ops.conv2d_wgrad_raw(x, dy, (cout, cin, 3, 3), 1, 1, True)
xs, dys = x[:4].double(), dy[:4].double()
ref_w = torch.nn.grad.conv2d_weight(xs, (cout, cin, 3, 3), dys, padding=1)
for c in cfgs:
    lib.srhip_debug_set(1, c)
    dw, db = ops.conv2d_wgrad_raw(x[:4].contiguous(memory_format=torch.channels_last), dy[:4].contiguous(memory_format=torch.channels_last), (cout, cin, 3, 3), 1, 1, True)
    e = float((dw.double() - ref_w).abs().max() / ref_w.abs().max())
    eb = float((db.double() - dys.sum((0, 2, 3))).abs().max() / dys.sum((0, 2, 3)).abs().max())
    dwf, dbf = fn()
    dwf2, dbf2 = fn()
    print('cfg %3d small-batch rel err dw %.2e db %.2e; full batch finite %s, repeat-identical %s' % (c, e, eb, bool(torch.isfinite(dwf).all()), bool(torch.equal(dwf, dwf2) and torch.equal(dbf, dbf2))), flush=True)
lib.srhip_debug_set(1, 0)
full0 = fn()[0].clone()
for c in cfgs:
    lib.srhip_debug_set(1, c)
    print('cfg %3d full batch max rel diff vs cfg 0: %.2e' % (c, float((fn()[0] - full0).abs().max() / full0.abs().max())))
times = {c: [] for c in cfgs}
for rnd in range(7):
    for c in cfgs:
        lib.srhip_debug_set(1, c)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for _ in range(30): fn()
        e.record(); torch.cuda.synchronize()
        times[c].append(s.elapsed_time(e) / 30)
lib.srhip_debug_set(1, 0)
for c in cfgs:
    t = sorted(times[c])
    print('cfg %3d median %.1f us  min %.1f us  %.0f TFLOP/s-equivalent' % (c, t[len(t) // 2] * 1e3, t[0] * 1e3, fl / t[len(t) // 2] / 1e9))
