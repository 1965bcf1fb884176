"""Census of the conv calls of one training step: every conv2d_{fwd,dgrad,wgrad}_raw call is timed on its own (device
synchronised around it), aggregated by (kind, shape).  Standalone times: no overlap between streams while measuring."""
import collections
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench  # noqa: E402
from sradsgan_amd import ops  # noqa: E402
from sradsgan_amd.train_step import TrainStep  # noqa: E402

dev = torch.device('cuda:0')
G, D, F = bench.build_networks(dev, 20240)
step = TrainStep(G, D, F)
B = 32
gen = torch.Generator().manual_seed(1)
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev)
lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev)
alpha = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
for _ in range(2):
    step(lr, hr, alpha)
torch.cuda.synchronize()
stats = collections.defaultdict(lambda: [0, 0.0])


def wrap(name, key_fn):
    orig = getattr(ops, name)

    def f(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = orig(*a, **k)
        torch.cuda.synchronize()
        s = stats[(name[7:-4].replace('_multi','').replace('_pp',''),) + key_fn(*a, **k)]
        s[0] += 1
        s[1] += (time.perf_counter() - t0) * 1e6
        return out
    setattr(ops, name, f)


wrap("conv2d_fwd_raw", lambda x, w, bias, stride, pad, slope=None, residual=None, rowscale=None, chanscale=None, graddata=False, **kw:
     (tuple(x.shape), tuple(w.shape), stride, 'lrelu' if slope is not None else '', 'res' if residual is not None else '', 'scale' if rowscale is not None else ''))
wrap('conv2d_dgrad_raw', lambda dy, w, x_shape, stride, pad, residual=None, actmask=None, slope=0.0, **kw:
     (tuple(dy.shape), tuple(w.shape), stride, 'mask' if actmask is not None else '', 'res' if residual is not None else '', ''))
wrap('conv2d_wgrad_raw', lambda x, dy, w_shape, stride, pad, with_bias=False, xrowscale=None, xchanscale=None, out=None, **kw:
     (tuple(x.shape), tuple(w_shape), stride, 'bias' if with_bias else '', '', 'scale' if xrowscale is not None else ''))

def _shape(t):
    return tuple(t.shape)


wrap('conv2d_wgrad_multi_raw', lambda items, on_stream=None, **kw:
     (_shape(items[0][0]), _shape(items[0][2]), items[0][4], 'x%d' % len(items), '', 'multi'))
wrap('conv2d_fwd_pp_raw', lambda x, w, bias, slope=None, out_pp=None, pool=False, **kw:
     (_shape(x), _shape(w), 1, 'lrelu' if slope is not None else '', 'pool' if pool else '', 'pp'))
wrap('conv2d_dgrad_pp_raw', lambda dy, w, residual=None, actmask=None, slope=0.0, out_pp=None, **kw:
     (_shape(dy), _shape(w), 1, 'mask' if actmask is not None else '', 'res' if residual is not None else '', 'pp'))
wrap('conv2d_wgrad_pp_raw', lambda items, accumulate=True, on_stream=None, **kw:
     (_shape(items[0][0]), _shape(items[0][2]), 1, 'x%d' % len(items), '', 'pp'))
step(lr, hr, alpha)
torch.cuda.synchronize()
stats.clear()          # the first wrapped step pays one-time allocations under the per-call synchronisation
step(lr, hr, alpha)
torch.cuda.synchronize()
rows = sorted(stats.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print('total conv time (standalone, incl. ~10 us launch+sync each): %.1f ms over %d calls' % (tot / 1e3, sum(v[0] for _, v in rows)))
for k, (n, t) in rows[:int(__import__("os").environ.get("TOP", "45"))]:
    kind, xs, ws, stride = k[0], k[1], k[2], k[3]
    cout, cin, kh, kw = ws
    n_, _, h, w = xs
    if kind == 'fwd':
        px = n_ * ((h + stride - 1) // stride) * ((w + stride - 1) // stride)
    elif kind == 'dgrad':
        px = n_ * h * w
    else:
        px = n_ * ((h + stride - 1) // stride) * ((w + stride - 1) // stride)
    gf = 2.0 * px * cout * cin * kh * kw / 1e9 * (int(k[4][1:]) if k[4].startswith('x') else 1)
    print('%-6s x%-20s w%-18s s%d %-5s %-4s %-5s calls %3d  avg %7.1f us  total %6.2f ms  %6.1f TF/s' % (kind, xs, ws, stride, k[4], k[5], k[6], n, t / n, t / 1e3, gf / (t / n) * 1e6 / 1e3))
