import os, sys
os.environ['SRHIP_STEP_DEBUG'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd.train_step import TrainStep
DEV = torch.device('cuda:0')
def run(**kw):
    (hg, hd, hf), _ = build_pair(2, 1, 4, DEV)
    step = TrainStep(hg, hd, hf, overlap_wgrad=False, overlap_d_step=False, **kw)
    snaps = []
    for it in range(3):
        out = step(O.det_fill('graph.lr.%d' % it, (4, 3, 24, 24), 0.5, 0.5).to(DEV), O.det_fill('graph.hr.%d' % it, (4, 3, 96, 96), 0.5, 0.5).to(DEV),
                   O.det_fill('graph.alpha.%d' % it, (4, 1, 1, 1), 0.5, 0.5).to(DEV))
        torch.cuda.synchronize()
        snaps.append(({k: out[k].clone() for k in out},
                      {('G.' + k): p.detach().clone() for k, p in hg.named_parameters()} | {('D.' + k): p.detach().clone() for k, p in hd.named_parameters()},
                      {k: b.detach().clone() for k, b in hd.named_buffers()}))
    return snaps
A, B = run(), run(use_graph=True)
for it in range(3):
    print('it', it, {k: float((A[it][0][k].double() - B[it][0][k].double()).abs().max()) for k in A[it][0]})
    dw = sorted(((float((A[it][1][k] - B[it][1][k]).abs().max()), k) for k in A[it][1]), reverse=True)[:3]
    db = sorted(((float((A[it][2][k].double() - B[it][2][k].double()).abs().max()), k) for k in A[it][2]), reverse=True)[:3]
    print('   weights', dw, 'buffers', db)
    if 'd_gen' in A[it][0]:
        d = (A[it][0]['d_gen'] - B[it][0]['d_gen']).flatten()
        print('   d_gen diff nonzero elements:', int((d != 0).sum()), 'of', d.numel(), 'values', A[it][0]['d_gen'].flatten()[:4].tolist(), B[it][0]['d_gen'].flatten()[:4].tolist())
