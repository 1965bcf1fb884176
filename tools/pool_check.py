import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from sradsgan_amd import ops
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
G, D, F = bench.build_networks(dev, 20240)
step = TrainStep(G, D, F)
g = torch.Generator().manual_seed(1)
hr = torch.rand(32, 3, 216, 216, generator=g).to(dev); lr = torch.rand(32, 3, 54, 54, generator=g).to(dev); al = torch.rand(32, 1, 1, 1, generator=g).to(dev)
for i in range(12):
    step(lr, hr, al)
    print(i, 'plane buffers created so far', ops.plane_pool.created, 'free lists', {k[1]: len(v) for k, v in ops.plane_pool.free.items()}, 'alloc GB %.1f' % (torch.cuda.memory_allocated() / 1e9), flush=True)
torch.cuda.synchronize()
