"""Which gradients differ between graph replay and eager launches once other autograd work ran earlier in the process
(the situation of tests/test_graph_gpu.py inside the whole suite)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytest
import torch
pre = sys.argv[1:] or ['tests/test_conv_gpu.py']
if pre != ['none']:
    pytest.main(['-q', '-m', 'gpu', '-x'] + pre)
from oracle import sradsgan_ref as O
from tests.parity_util import build_pair
from sradsgan_amd.train_step import TrainStep
DEV = torch.device('cuda:0')


def run(iters, **kw):
    (hg, hd, hf), _ = build_pair(2, 1, 4, DEV)
    step = TrainStep(hg, hd, hf, overlap_wgrad=False, overlap_d_step=False, **kw)
    snaps = []
    for it in range(iters):
        out = step(O.det_fill('graph.lr.%d' % it, (4, 3, 24, 24), 0.5, 0.5).to(DEV), O.det_fill('graph.hr.%d' % it, (4, 3, 96, 96), 0.5, 0.5).to(DEV),
                   O.det_fill('graph.alpha.%d' % it, (4, 1, 1, 1), 0.5, 0.5).to(DEV))
        g = {('G.' + k): p.grad.detach().clone() for k, p in hg.named_parameters()} | {('D.' + k): p.grad.detach().clone() for k, p in hd.named_parameters()}
        snaps.append(({k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}, g))
    torch.cuda.synchronize()
    return snaps


run(2)
graph = run(4, use_graph=True)
for trial in range(2):
    eager = run(4)
    for it in range(4):
        dg = [(float((graph[it][1][k] - eager[it][1][k]).abs().max() / graph[it][1][k].abs().max().clamp_min(1e-30)), k) for k in graph[it][1]]
        bad = sorted([d for d in dg if d[0] > 0], reverse=True)
        ds = {k: graph[it][0][k] - eager[it][0][k] for k in graph[it][0] if graph[it][0][k] != eager[it][0][k]}
        print('trial', trial, 'it', it, 'scalars differing', ds, '| %d grads differ' % len(bad), bad[:4], flush=True)
