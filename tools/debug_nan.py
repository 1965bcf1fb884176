import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
B = 32
def run(use_graph, n, sync_each, tag):
    G, D, F = bench.build_networks(dev, seed=20240)
    step = TrainStep(G, D, F, use_graph=use_graph)
    gen = torch.Generator().manual_seed(1234)
    hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev)
    lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev)
    alpha = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
    hist = []
    for i in range(n):
        o = step(lr, hr, alpha)
        if sync_each:
            torch.cuda.synchronize()
        hist.append({k: o[k].clone() for k in ('loss_G', 'loss_D', 'gp')})
    torch.cuda.synchronize()
    print(tag, ' '.join('%d:%.4g/%.4g' % (i, float(h['loss_G']), float(h['loss_D'])) for i, h in enumerate(hist)), flush=True)
    pn = sum(int(torch.isnan(p).sum()) for p in list(G.parameters()) + list(D.parameters()))
    print(tag, 'nan params:', pn, flush=True)
which = sys.argv[1] if len(sys.argv) > 1 else 'all'
if which in ('all', 'a'): run(False, 9, False, 'eager-nosync')
if which in ('all', 'b'): run(True, 9, True, 'graph-sync ')
if which in ('all', 'c'): run(True, 9, False, 'graph-nosync')
