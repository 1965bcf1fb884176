"""Round 4: software-pipelined row-tap weight gradient (srhip_debug_set(9, 1): conversion of chunk k+1 beside the MFMAs of chunk
k, 2 blocks / CU, 5-deep ring) against the default, for several block targets (srhip_debug_set(1, target >= 100)).
  python tools/sweep_wgrad_pipe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
CL = torch.channels_last
torch.manual_seed(0)
ROUNDS, NIT = int(os.environ.get('ROUNDS', '5')), int(os.environ.get('NIT', '100'))
B = int(os.environ.get('B', '32'))
mk = lambda n, c, h, w: torch.randn(n, c, h, w, device=dev).contiguous(memory_format=CL)


def pair(xs, dys, cout, cin):
    items = [(x, dy, torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev), 1, 1) for x, dy in zip(xs, dys)]
    ops.conv2d_wgrad_multi_raw(items)
    return [it[2] for it in items] + [it[3] for it in items]


def single(x, dy, cout, cin):
    return list(ops.conv2d_wgrad_raw(x, dy, (cout, cin, 3, 3), 1, 1, True))


VARS = [(0, 0)] + [(1, t) for t in (int(v) for v in os.environ.get('TARGETS', '480,512,384,768').split(','))]


def setv(v):
    lib.srhip_debug_set(9, v[0]); lib.srhip_debug_set(1, v[1])


def bench(name, fn, flops):
    setv(VARS[0]); ref = fn()
    for v in VARS[1:]:
        setv(v); got = fn()
        err = max(float((a - b).abs().max() / a.abs().max()) for a, b in zip(ref, got))
        print('%-40s pipe %d target %4d  max rel diff vs default %.2e' % (name, v[0], v[1], err), flush=True)
    t = {v: [] for v in VARS}
    for _ in range(ROUNDS):
        for v in VARS:
            setv(v)
            for _ in range(10): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(True), torch.cuda.Event(True)
            s.record()
            for _ in range(NIT): fn()
            e.record(); torch.cuda.synchronize()
            t[v].append(s.elapsed_time(e) / NIT)
    for v in VARS:
        tt = sorted(t[v]); med = tt[len(tt) // 2]
        print('%-40s pipe %d target %4d  median %.1f us  min %.1f  (frac %.3f)' % (name, v[0], v[1], med * 1e3, tt[0] * 1e3, flops / med / 1e9 / 833.3), flush=True)


with ops.conv_math('bf16x3'):
    x64 = [mk(B, 64, 54, 54) for _ in range(2)]
    t256 = [mk(B, 256, 54, 54) for _ in range(2)]
    fl = 2.0 * B * 54 * 54 * 256 * 64 * 9
    bench('pair conv1 64->256', lambda: pair(x64, t256, 256, 64), 2 * fl)
    bench('pair conv2 256->64', lambda: pair(t256, x64, 64, 256), 2 * fl)
    bench('single conv1 64->256', lambda: single(x64[0], t256[0], 256, 64), fl)
    # small ragged shapes: exactness of the pipelined loop's prologue / epilogue (1, 2, 3 chunks per split ...)
    for (n, cin, cout, h, w) in [(1, 64, 128, 5, 19), (2, 64, 256, 13, 8), (1, 128, 64, 16, 16), (3, 64, 192, 9, 70), (2, 256, 64, 27, 27)]:
        x, dy = mk(n, cin, h, w), mk(n, cout, h, w)
        setv((0, 0)); ref = single(x, dy, cout, cin)
        setv((1, 0)); got = single(x, dy, cout, cin)
        print('small n%d %d->%d %dx%d: bit-identical %s' % (n, cin, cout, h, w, all(torch.equal(a, b) for a, b in zip(ref, got))), flush=True)
setv((0, 0))
