import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd.train_step import TrainStep
DEV = torch.device('cuda:0')
def run(**kw):
    (hg, hd, hf), _ = build_pair(2, 1, 4, DEV)
    step = TrainStep(hg, hd, hf, **kw)
    snaps = []
    for it in range(3):
        out = step(O.det_fill('graph.lr.%d' % it, (4, 3, 24, 24), 0.5, 0.5).to(DEV), O.det_fill('graph.hr.%d' % it, (4, 3, 96, 96), 0.5, 0.5).to(DEV),
                   O.det_fill('graph.alpha.%d' % it, (4, 1, 1, 1), 0.5, 0.5).to(DEV))
        torch.cuda.synchronize()
        snaps.append(({k: float(out[k]) for k in ('loss_G', 'loss_D', 'gp')},
                      {('G.' + k): p.detach().clone() for k, p in hg.named_parameters()} | {('D.' + k): p.detach().clone() for k, p in hd.named_parameters()},
                      {('G.' + k): p.grad.detach().clone() for k, p in hg.named_parameters()} | {('D.' + k): p.grad.detach().clone() for k, p in hd.named_parameters()},
                      {k: b.detach().clone() for k, b in hd.named_buffers()}))
    return snaps
A = run()
for name, kw in (('eager, no overlap', dict(overlap_wgrad=False, overlap_d_step=False)), ('graph', dict(use_graph=True))):
    Bv = run(**kw)
    for it in range(3):
        dw = sorted(((float((A[it][1][k] - Bv[it][1][k]).abs().max()), k) for k in A[it][1]), reverse=True)[:3]
        dg = sorted(((float((A[it][2][k] - Bv[it][2][k]).abs().max() / max(float(A[it][2][k].abs().max()), 1e-30)), k) for k in A[it][2]), reverse=True)[:3]
        db = sorted(((float((A[it][3][k].double() - Bv[it][3][k].double()).abs().max()), k) for k in A[it][3]), reverse=True)[:2]
        print(name, 'it', it, 'scalars', [A[it][0][k] - Bv[it][0][k] for k in A[it][0]], '\n   weights', dw, '\n   grads', dg, '\n   buffers', db, flush=True)
