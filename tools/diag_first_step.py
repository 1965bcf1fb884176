import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.parity_util import build_pair
from oracle import sradsgan_ref as O
from sradsgan_amd.train_step import TrainStep
from sradsgan_amd import ops
DEV = torch.device('cuda:0')
scale, B, blocks, groups = 4, 32, 12, 3
side = 216 // scale
lr = O.det_fill('first.lr', (B, 3, side, side), 0.5, 0.5).to(DEV)
hr = O.det_fill('first.hr', (B, 3, side * scale, side * scale), 0.5, 0.5).to(DEV)
al = O.det_fill('first.alpha', (B, 1, 1, 1), 0.5, 0.5).to(DEV)
def run():
    (hg, hd, hf), _ = build_pair(blocks, groups, scale, DEV)
    step = TrainStep(hg, hd, hf)
    out = step(lr, hr, al)
    torch.cuda.synchronize()
    names = [k for k, _ in hg.named_parameters()] + ['D.' + k for k, _ in hd.named_parameters()]
    grads = [p.grad.detach().clone() for p in list(hg.parameters()) + list(hd.parameters())]
    return names, grads
for trial in range(2):
    n1, g1 = run()
    keep = [torch.full(((256 << 20) // 4,), float('nan'), device=DEV) for _ in range(24)] + [torch.full(((2 << 20) // 4,), float('nan'), device=DEV) for _ in range(64)]
    torch.cuda.synchronize(); del keep
    n2, g2 = run()
    bad = [(n, float((a - b).abs().max()), float(a.abs().max())) for n, a, b in zip(n1, g1, g2) if not torch.equal(a, b)]
    print('trial', trial, 'tensors that differ:', len(bad), 'of', len(g1))
    for r in bad[:12]: print('   ', r)
