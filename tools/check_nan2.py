"""One full-depth step at B = 32 on a poisoned allocator (see check_uninit.py), with the stream overlap flags from the environment."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd.train_step import TrainStep
DEV = torch.device('cuda:0')


def poison():
    torch.cuda.synchronize()
    keep = []
    for nbytes, cnt in ((1 << 30, 24), (256 << 20, 16), (32 << 20, 32), (2 << 20, 64), (64 << 10, 256), (4 << 10, 512), (512, 1024)):
        for _ in range(cnt):
            keep.append(torch.full((nbytes // 4,), float('nan'), device=DEV))
    torch.cuda.synchronize()
    del keep


B = 32
(hg, hd, hf), _ = build_pair(12, 3, 4, DEV)
step = TrainStep(hg, hd, hf, overlap_wgrad=os.environ.get('OW', '1') == '1', overlap_d_step=os.environ.get('OD', '1') == '1')
lr = O.det_fill('bench_b12.lr.0', (B, 3, 54, 54), 0.5, 0.5).to(DEV)
hr = O.det_fill('bench_b12.hr.0', (B, 3, 216, 216), 0.5, 0.5).to(DEV)
al = O.det_fill('bench_b12.alpha.0', (B, 1, 1, 1), 0.5, 0.5).to(DEV)
if os.environ.get('POISON', '1') == '1':
    poison()
out = step(lr, hr, al)
torch.cuda.synchronize()
badg = [k for k, p in hg.named_parameters() if not torch.isfinite(p.grad).all()]
badd = [k for k, p in hd.named_parameters() if not torch.isfinite(p.grad).all()]
print('OW=%s OD=%s: scalars %s; non-finite gradients: G %d %s, D %d %s' % (os.environ.get('OW', '1'), os.environ.get('OD', '1'), {k: round(float(out[k]), 5) for k in ('loss_G', 'loss_D', 'gp')},
                                                                             len(badg), badg[-2:], len(badd), badd[:2]), flush=True)
