"""Is the B=12 full-depth step bit-reproducible run to run?  (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd.train_step import TrainStep
DEV = torch.device('cuda:0')
B = int(os.environ.get('B', '12'))
def run():
    (hg, hd, hf), _ = build_pair(12, 3, 4, DEV)
    step = TrainStep(hg, hd, hf)
    res = []
    for it in range(2):
        lr = O.det_fill('bench_b12.lr.%d' % it, (B, 3, 54, 54), 0.5, 0.5).to(DEV)
        hr = O.det_fill('bench_b12.hr.%d' % it, (B, 3, 216, 216), 0.5, 0.5).to(DEV)
        al = O.det_fill('bench_b12.alpha.%d' % it, (B, 1, 1, 1), 0.5, 0.5).to(DEV)
        out = step(lr, hr, al)
        torch.cuda.synchronize()
        res.append(({k: float(out[k]) for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')},
                    {('G.' + k): p.grad.clone() for k, p in hg.named_parameters()} | {('D.' + k): p.grad.clone() for k, p in hd.named_parameters()}))
    return res
a = run()
for trial in range(3):
    b = run()
    for it in range(2):
        ds = max(abs(a[it][0][k] - b[it][0][k]) for k in a[it][0])
        worst = sorted(((float((a[it][1][k] - b[it][1][k]).abs().max() / max(float(a[it][1][k].abs().max()), 1e-30)), k) for k in a[it][1]), reverse=True)[:4]
        print('trial %d it %d: scalar diff %.3e; worst grad rel diffs %s' % (trial, it, ds, [(('%.1e' % v), k) for v, k in worst]), flush=True)
