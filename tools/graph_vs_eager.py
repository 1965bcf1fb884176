"""Step time of the four launch / stream combinations at the bench shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
gen = torch.Generator().manual_seed(1)
B = 32
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev); lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev); al = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
for name, kw in (('eager, three streams', {}), ('eager, one stream', dict(overlap_wgrad=False, overlap_d_step=False)), ('hipGraph, one stream', dict(use_graph=True))):
    G, D, F = bench.build_networks(dev, 20240)
    step = TrainStep(G, D, F, **kw)
    for _ in range(6):
        step(lr, hr, al)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step(lr, hr, al)
    host = (time.perf_counter() - t0) / 10 * 1e3
    torch.cuda.synchronize()
    print('%-24s %.2f ms per step (host %.1f ms)' % (name, (time.perf_counter() - t0) / 10 * 1e3, host), flush=True)
    del step, G, D, F
