import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
def run(use_graph):
    G, D, F = bench.build_networks(dev, seed=20240)
    step = TrainStep(G, D, F, use_graph=use_graph)
    gen = torch.Generator().manual_seed(1234)
    hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev)
    lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev)
    alpha = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
    out = []
    for i in range(N):
        o = step(lr, hr, alpha)
        out.append([float(o[k]) for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')])
    return out
a = run(False)
b = run(True)
for i, (x, y) in enumerate(zip(a, b)):
    print(i, 'eager', ['%.5g' % v for v in x])
    print(i, 'graph', ['%.5g' % v for v in y])
