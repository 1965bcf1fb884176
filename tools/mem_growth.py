"""Does the caching allocator's reserved memory plateau over a long run of the three-stream step (record_stream defers reuse)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
G, D, F = bench.build_networks(dev, 20240)
step = TrainStep(G, D, F)
gen = torch.Generator().manual_seed(1)
B = 32
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev); lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev); al = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
import time
t0 = None
for it in range(1, 401):
    if it == 101:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    step(lr, hr, al)
    if it in (1, 2, 5, 10, 25, 50, 100, 200, 300, 400):
        torch.cuda.synchronize()
        if it == 400:
            print('steps 101-400: %.2f ms per step' % ((time.perf_counter() - t0) / 300 * 1e3))
        print('step %3d: allocated %.1f GB, peak %.1f GB, reserved %.1f GB' % (it, torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.max_memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30), flush=True)
