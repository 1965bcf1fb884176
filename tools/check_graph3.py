import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd.train_step import TrainStep
DEV = torch.device('cuda:0')
def run(sync_every=False, **kw):
    (hg, hd, hf), _ = build_pair(2, 1, 4, DEV)
    step = TrainStep(hg, hd, hf, overlap_wgrad=False, overlap_d_step=False, **kw)
    snaps = []
    for it in range(2):
        out = step(O.det_fill('graph.lr.%d' % it, (4, 3, 24, 24), 0.5, 0.5).to(DEV), O.det_fill('graph.hr.%d' % it, (4, 3, 96, 96), 0.5, 0.5).to(DEV),
                   O.det_fill('graph.alpha.%d' % it, (4, 1, 1, 1), 0.5, 0.5).to(DEV))
        g = {('G.' + k): p.grad.detach().clone() for k, p in hg.named_parameters()} | {('D.' + k): p.grad.detach().clone() for k, p in hd.named_parameters()}
        if sync_every: torch.cuda.synchronize()
        snaps.append((float(out['loss_gan']), g))
    torch.cuda.synchronize()
    return snaps
for label, kw in (('eager-noov', {}), ('graph', dict(use_graph=True))):
    base = run(**kw)
    for trial in range(3):
        other = run(sync_every=(trial == 2), **kw)
        for it in range(2):
            dg = sorted(((float((base[it][1][k] - other[it][1][k]).abs().max()), k) for k in base[it][1]), reverse=True)[:2]
            print(label, 'trial', trial, 'it', it, 'loss_gan diff %.2e' % (base[it][0] - other[it][0]), 'grad diffs', dg, flush=True)
