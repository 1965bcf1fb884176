import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd.train_step import TrainStep
DEV = torch.device('cuda:0')
if os.environ.get('PRE', '1') == '1':
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    xx = torch.zeros(1 << 22, device='cuda:0')
    with torch.cuda.stream(a): xx.add_(1.0)
    with torch.cuda.stream(b): xx.add_(1.0)
    torch.cuda.synchronize()
scale, B, blocks, groups = [int(v) for v in os.environ.get('CFG', '4,32,12,3').split(',')]
side = 216 // scale
lr = O.det_fill('first.lr', (B, 3, side, side), 0.5, 0.5).to(DEV)
hr = O.det_fill('first.hr', (B, 3, side * scale, side * scale), 0.5, 0.5).to(DEV)
al = O.det_fill('first.alpha', (B, 1, 1, 1), 0.5, 0.5).to(DEV)
def run():
    (hg, hd, hf), _ = build_pair(blocks, groups, scale, DEV)
    step = TrainStep(hg, hd, hf)
    out = step(lr, hr, al)
    torch.cuda.synchronize()
    names = ['G.' + k for k, _ in hg.named_parameters()] + ['D.' + k for k, _ in hd.named_parameters()]
    grads = [p.grad.detach().clone() for p in list(hg.parameters()) + list(hd.parameters())]
    return names, grads
n1, g1 = run()
if os.environ.get('POISON', '1') == '1':
    keep = [torch.full(((256 << 20) // 4,), float('nan'), device=DEV) for _ in range(24)] + [torch.full(((2 << 20) // 4,), float('nan'), device=DEV) for _ in range(64)]
    torch.cuda.synchronize(); del keep
n2, g2 = run()
bad = [(n, float((a - b).abs().max()), float(a.abs().max())) for n, a, b in zip(n1, g1, g2) if not torch.equal(a, b)]
print(len(bad), 'tensors differ of', len(g1))
for b in bad[:25]: print(b)
