"""RAB conv1 fprop (3x3 64 -> 256 @ 54x54, B = 32, bias + LeakyReLU), sustained loops of ~1 s each with HIP events around batches of 100:
fp32 tensors / padded planes, one operand set (warm: x sits in the Infinity Cache) / three sets in rotation (cold)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd import ops
dev = torch.device('cuda:0')
B, S = 32, 54
w = torch.nn.Parameter(torch.randn(256, 64, 3, 3, device=dev) * 0.02); b = torch.randn(256, device=dev) * 0.01
xs = [torch.randn(B, 64, S, S, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(3)]
xpp = [ops.pp_from_f32(x) for x in xs]
opp = [ops.pp_empty(B, 256, S, S, dev) for _ in range(3)]
flops = 2.0 * B * S * S * 256 * 64 * 9


def sustained(fn, secs=1.0):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    with bench.PowerSampler(0) as ps:
        t_end = time.perf_counter() + secs
        n, ms, pairs = 0, 0.0, []
        while time.perf_counter() < t_end:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(100): fn()
            e.record(); pairs.append((s, e)); n += 100
            if len(pairs) >= 8:
                pairs[0][1].synchronize(); ms += pairs[0][0].elapsed_time(pairs[0][1]); pairs.pop(0)
        torch.cuda.synchronize()
        for s, e in pairs: ms += s.elapsed_time(e)
    pw = ps.summary() or {}
    return ms / n * 1e3, pw.get('sclk_mhz_mean'), pw.get('watts_mean')


rot = [0]
def nxt():
    rot[0] = (rot[0] + 1) % 3
    return rot[0]
forms = [
    ('fp32 -> fp32, one set', lambda: ops.conv2d_fwd_raw(xs[0], w, b, 1, 1, 0.2)),
    ('fp32 -> fp32, three sets', lambda: ops.conv2d_fwd_raw(xs[nxt()], w, b, 1, 1, 0.2)),
    ('planes -> planes, one set', lambda: ops.conv2d_fwd_pp_raw(xpp[0], w, b, 0.2, out_pp=opp[0])),
    ('planes -> planes, three sets', lambda: ops.conv2d_fwd_pp_raw(xpp[nxt()], w, b, 0.2, out_pp=opp[rot[0]])),
    ('fp32 -> planes, one set', lambda: ops.conv2d_fwd_pp_raw(xs[0], w, b, 0.2, out_pp=opp[0])),
    ('planes -> fp32 (no planes out), one set', lambda: ops.conv2d_fwd_pp_raw(xpp[0], w, b, 0.2)),
]
for rep in range(2):
    for name, fn in forms:
        us, clk, wt = sustained(fn)
        print('%-42s %6.1f us  %5.3f of 833 TF/s   sclk %s MHz  %s W' % (name, us, flops / us / 1e6 / 833.3, clk, wt), flush=True)
