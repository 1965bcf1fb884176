"""Round 4: wgrad_rowtap_kernel with the chunk position as the DMA's scalar offset (srhip_debug_set(8, 1), default) against the
per-lane offset arithmetic of rounds 1-3 (srhip_debug_set(8, 0)): bit-exactness and interleaved timing, single and pair
launches of the two RAB convs, a 216 x 216 D shape (no paired tails) and ragged small shapes.
  python tools/sweep_wgrad_addr.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import ops, _hip
dev = torch.device('cuda:0')
lib = _hip.lib()
CL = torch.channels_last
torch.manual_seed(0)
ROUNDS, NIT = int(os.environ.get('ROUNDS', '5')), int(os.environ.get('NIT', '100'))


def mk(n, c, h, w):
    return torch.randn(n, c, h, w, device=dev).contiguous(memory_format=CL)


def single(x, dy, cout, cin):
    return ops.conv2d_wgrad_raw(x, dy, (cout, cin, 3, 3), 1, 1, True)


def pair(xs, dys, cout, cin):
    items = [(x, dy, torch.zeros(cout, cin, 3, 3, device=dev), torch.zeros(cout, device=dev), 1, 1) for x, dy in zip(xs, dys)]
    ops.conv2d_wgrad_multi_raw(items)
    return [it[2] for it in items] + [it[3] for it in items]


def check(name, fn):
    lib.srhip_debug_set(8, 0)
    ref = fn()
    lib.srhip_debug_set(8, 1)
    got = fn()
    ref = ref if isinstance(ref, (list, tuple)) else [ref]
    got = got if isinstance(got, (list, tuple)) else [got]
    same = all(torch.equal(a, b) for a, b in zip(ref, got) if a is not None)
    print('%-44s bit-identical: %s' % (name, same), flush=True)
    return same


def timeit(name, fn, flops):
    t = {0: [], 1: []}
    for _ in range(ROUNDS):
        for v in (0, 1):
            lib.srhip_debug_set(8, v)
            for _ in range(10): fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(True), torch.cuda.Event(True)
            s.record()
            for _ in range(NIT): fn()
            e.record(); torch.cuda.synchronize()
            t[v].append(s.elapsed_time(e) / NIT)
    for v in (0, 1):
        tt = sorted(t[v]); med = tt[len(tt) // 2]
        print('%-44s addr %d median %.1f us  min %.1f  (%.0f TFLOP/s-eq, frac %.3f)' % (name, v, med * 1e3, tt[0] * 1e3, flops / med / 1e9, flops / med / 1e9 / 833.3), flush=True)


with ops.conv_math('bf16x3'):
    ok = True
    for (n, cin, cout, h, w) in [(2, 64, 256, 54, 54), (3, 256, 64, 27, 27), (1, 64, 128, 24, 40), (2, 128, 128, 17, 23), (1, 64, 256, 9, 70),
                                 (2, 128, 64, 16, 16), (1, 64, 192, 5, 19), (2, 64, 256, 13, 8)]:
        x, dy = mk(n, cin, h, w), mk(n, cout, h, w)
        ok &= check('single n%d %d->%d %dx%d' % (n, cin, cout, h, w), lambda: single(x, dy, cout, cin))
    B = int(os.environ.get('B', '32'))
    x64 = [mk(B, 64, 54, 54) for _ in range(2)]
    t256 = [mk(B, 256, 54, 54) for _ in range(2)]
    ok &= check('pair conv1 64->256 B%d' % B, lambda: pair(x64, t256, 256, 64))
    ok &= check('pair conv2 256->64 B%d' % B, lambda: pair(t256, x64, 64, 256))
    print('ALL BIT-IDENTICAL' if ok else 'MISMATCH', flush=True)
    fl = 2.0 * B * 54 * 54 * 256 * 64 * 9
    timeit('pair conv1 64->256 (2 convolutions + reduce)', lambda: pair(x64, t256, 256, 64), 2 * fl)
    timeit('pair conv2 256->64 (2 convolutions + reduce)', lambda: pair(t256, x64, 64, 256), 2 * fl)
    timeit('single conv1 64->256', lambda: single(x64[0], t256[0], 256, 64), fl)
    xd, dyd = mk(8, 64, 216, 216), mk(8, 128, 216, 216)
    ok &= check('single D-like 64->128 216x216 B8', lambda: single(xd, dyd, 128, 64))
    timeit('single D-like 64->128 216x216 B8', lambda: single(xd, dyd, 128, 64), 2.0 * 8 * 216 * 216 * 128 * 64 * 9)
lib.srhip_debug_set(8, 1)
