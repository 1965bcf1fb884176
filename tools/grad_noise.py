"""How accurate are the D / G gradients of one full-size training iteration (B=2)?  Compares, per network, the fp32
CPU oracle, the HIP path in fp32 mode and the HIP path in bf16x3 mode against the fp64 oracle (grad_score metric of
tests/parity_util.py)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from oracle import sradsgan_ref as O
import parity_util as P
from sradsgan_amd import ops
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
tag, batch, lr_side, scale = 'train_full', 2, 54, 4
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'train_full.npz'))
lr_img = O.det_fill('%s.lr.0' % tag, (batch, 3, lr_side, lr_side), 0.5, 0.5)
hr_img = O.det_fill('%s.hr.0' % tag, (batch, 3, lr_side * scale, lr_side * scale), 0.5, 0.5)
alpha = torch.from_numpy(g['alpha0'])
def oracle(dtype):
    og = O.GeneratorResNet(O.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=4); od, of = O.Discriminator(), O.FeatureExtractor()
    O.det_init_(og, prefix='G.'), O.det_init_(od, prefix='D.'), O.det_init_(of, prefix='F.')
    og, od, of = og.to(dtype), od.to(dtype), of.to(dtype)
    oG = torch.optim.Adam(og.parameters(), lr=2e-4); oD = torch.optim.Adam(od.parameters(), lr=2e-4)
    t = time.time(); O.train_step(og, od, of, oG, oD, lr_img.to(dtype), hr_img.to(dtype), alpha.to(dtype)); print('oracle', dtype, '%.1f s' % (time.time() - t), flush=True)
    return og, od
g64, d64 = oracle(torch.float64)
g32, d32 = oracle(torch.float32)
print('fp32 oracle vs fp64:  G %.3e (%s)   D %.3e (%s)' % (P.grad_score((g32,), (g64,)) + P.grad_score((d32,), (d64,))))
for mode in ('fp32', 'bf16x3'):
    with ops.conv_math(mode):
        (hg, hd, hf), _ = P.build_pair(12, 3, 4, dev)
        step = TrainStep(hg, hd, hf)
        step(lr_img.to(dev), hr_img.to(dev), alpha.to(dev))
        torch.cuda.synchronize()
        print('HIP %-6s vs fp64:     G %.3e (%s)   D %.3e (%s)' % ((mode,) + P.grad_score((hg,), (g64,)) + P.grad_score((hd,), (d64,))))
        print('HIP %-6s vs fp32 ora: G %.3e (%s)   D %.3e (%s)' % ((mode,) + P.grad_score((hg,), (g32,)) + P.grad_score((hd,), (d32,))), flush=True)
