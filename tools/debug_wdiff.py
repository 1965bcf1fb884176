import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair, rel_err
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
(hg, hd, hf), (og, od, of) = build_pair(2, 1, 4, dev)
step = TrainStep(hg, hd, hf)
oG = torch.optim.Adam(og.parameters(), lr=2e-4, betas=(0.9, 0.999))
oD = torch.optim.Adam(od.parameters(), lr=2e-4, betas=(0.9, 0.999))
for it in range(2):
    lr_img = O.det_fill('dbg.lr.%d' % it, (2, 3, 8, 8), 0.5, 0.5)
    hr_img = O.det_fill('dbg.hr.%d' % it, (2, 3, 32, 32), 0.5, 0.5)
    alpha = O.det_fill('dbg.alpha.%d' % it, (2, 1, 1, 1), 0.5, 0.5)
    # snapshot grads: run oracle
    want = O.train_step(og, od, of, oG, oD, lr_img, hr_img, alpha)
    got = step(lr_img.to(dev), hr_img.to(dev), alpha.to(dev))
    print(it, {k: (float(got[k]), want[k]) for k in ('loss_G', 'loss_D', 'gp')})
    rows = []
    for net_h, net_o, tag in ((hg, og, 'G'), (hd, od, 'D')):
        ho = dict(net_o.named_parameters())
        for k, p in net_h.named_parameters():
            e = rel_err(p, ho[k])
            ge = rel_err(p.grad, ho[k].grad) if ho[k].grad is not None else -1
            rows.append((e, ge, tag + '.' + k, float(ho[k].grad.abs().max()) if ho[k].grad is not None else 0))
    rows.sort(reverse=True)
    for r in rows[:12]:
        print('  wdiff %.3e graddiff %.3e %s |grad|max %.3e' % r)
