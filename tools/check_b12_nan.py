"""Which debug configuration makes the B=12 step go NaN, and where (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd import _hip
from sradsgan_amd.train_step import TrainStep
DEV = torch.device('cuda:0')
lib = _hip.lib()
B = 12
for name, dbg in (('cfg21', ((0, 21),)), ('cfg20', ((0, 20),)), ('cfg23+w7', ((0, 23), (1, 7)))):
    for k, v in dbg: lib.srhip_debug_set(k, v)
    (hg, hd, hf), _ = build_pair(12, 3, 4, DEV)
    step = TrainStep(hg, hd, hf)
    for it in range(2):
        lr = O.det_fill('bench_b12.lr.%d' % it, (B, 3, 54, 54), 0.5, 0.5).to(DEV)
        hr = O.det_fill('bench_b12.hr.%d' % it, (B, 3, 216, 216), 0.5, 0.5).to(DEV)
        al = O.det_fill('bench_b12.alpha.%d' % it, (B, 1, 1, 1), 0.5, 0.5).to(DEV)
        out = step(lr, hr, al)
        vals = {k: float(out[k]) for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')}
        bad = [k for n_, net in (('G', hg), ('D', hd)) for k, p in net.named_parameters() if not torch.isfinite(p.grad).all()]
        badw = [k for n_, net in (('G', hg), ('D', hd)) for k, p in net.named_parameters() if not torch.isfinite(p).all()]
        print(name, 'it', it, vals, 'non-finite grads:', bad[:6], len(bad), 'non-finite weights:', badw[:6], len(badw), flush=True)
    for k, v in dbg: lib.srhip_debug_set(k, 0)
    del step, hg, hd, hf
