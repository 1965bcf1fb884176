"""Which parameter gradients are non-finite after one full-depth step at batch B (parity-test initialisation)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd.train_step import TrainStep
DEV = torch.device('cuda:0')
for B in [int(v) for v in os.environ.get('BS', '12,24,32').split(',')]:
    (hg, hd, hf), _ = build_pair(12, 3, 4, DEV)
    step = TrainStep(hg, hd, hf)
    lr = O.det_fill('bench_b12.lr.0', (B, 3, 54, 54), 0.5, 0.5).to(DEV)
    hr = O.det_fill('bench_b12.hr.0', (B, 3, 216, 216), 0.5, 0.5).to(DEV)
    al = O.det_fill('bench_b12.alpha.0', (B, 1, 1, 1), 0.5, 0.5).to(DEV)
    out = step(lr, hr, al)
    torch.cuda.synchronize()
    bad = [('G.' + k) for k, p in hg.named_parameters() if not torch.isfinite(p.grad).all()] + [('D.' + k) for k, p in hd.named_parameters() if not torch.isfinite(p.grad).all()]
    tot = len(list(hg.parameters())) + len(list(hd.parameters()))
    print('B=%d: scalars %s; %d of %d gradients non-finite; first: %s ... last: %s' % (B, {k: round(float(out[k]), 5) for k in ('loss_G', 'loss_D', 'gp')}, len(bad), tot, bad[:3], bad[-3:]), flush=True)
