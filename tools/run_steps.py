"""N plain training steps at the bench shape and nothing else (a clean target for profilers).  usage: run_steps.py [N=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
B = 32
G, D, F = bench.build_networks(dev, 20240)
step = TrainStep(G, D, F)
gen = torch.Generator().manual_seed(1)
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev); lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev); al = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3): step(lr, hr, al)
torch.cuda.synchronize()
