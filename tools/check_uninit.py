"""Hunt for reads of uninitialised device memory: poison the caching allocator's free blocks with NaN patterns (fresh hipMalloc
pages come up zeroed, which hides such reads in the first model of a process), then run pieces of the step and report which
results are non-finite."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd import ops
DEV = torch.device('cuda:0')


def poison():
    torch.cuda.synchronize()
    keep = []
    for nbytes, cnt in ((1 << 30, 24), (256 << 20, 16), (32 << 20, 32), (2 << 20, 64), (64 << 10, 256), (4 << 10, 512), (512, 1024)):
        for _ in range(cnt):
            keep.append(torch.full((nbytes // 4,), float('nan'), device=DEV))
    torch.cuda.synchronize()
    del keep


B = int(os.environ.get('B', '32'))
(hg, hd, hf), _ = build_pair(12, 3, 4, DEV)
lr = O.det_fill('bench_b12.lr.0', (B, 3, 54, 54), 0.5, 0.5).to(DEV)
hr = O.det_fill('bench_b12.hr.0', (B, 3, 216, 216), 0.5, 0.5).to(DEV)
for p in hf.parameters():
    p.requires_grad_(False)
poison()
gen = hg(lr)
print('generator forward finite:', bool(torch.isfinite(gen).all()), flush=True)
gen_d = gen.detach().requires_grad_(True)
poison()
l1 = ops.l1_mean(gen_d, hr)
(g1,) = torch.autograd.grad(l1, gen_d)
print('L1: loss finite', bool(torch.isfinite(l1)), 'grad finite', bool(torch.isfinite(g1).all()), flush=True)
poison()
with torch.no_grad():
    rf = hf(hr)
ff = hf(gen_d)
content = ops.l1_mean(ff, rf)
poison()
(g2,) = torch.autograd.grad(content, gen_d)
print('VGG content: loss finite', bool(torch.isfinite(content)), 'grad finite', bool(torch.isfinite(g2).all()), flush=True)
poison()
dg = hd(gen_d)
lg = -ops.mean(dg)
print('D(gen) forward finite:', bool(torch.isfinite(dg).all()), flush=True)
poison()
with ops.backward_scope(skip_params=list(hd.parameters())):
    (g3,) = torch.autograd.grad(lg, gen_d)
print('D(gen): grad wrt gen_hr finite', bool(torch.isfinite(g3).all()), flush=True)
poison()
gen.backward(g1 + g2 + g3)
bad = [k for k, p in hg.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
print('generator backward: %d non-finite gradients' % len(bad), bad[:5], flush=True)
