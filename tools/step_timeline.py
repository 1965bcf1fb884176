"""Per-stream milestones of the training step measured with HIP events (no profiler: the host runs at full speed), in ms from
the step's first kernel; mean over the last steps.  SRHIP_STEP_TIMELINE=1 switches the markers in TrainStep on."""
import os, sys
os.environ['SRHIP_STEP_TIMELINE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
G, D, F = bench.build_networks(dev, seed=20240)
step = TrainStep(G, D, F)
B = 32
gen = torch.Generator().manual_seed(1234)
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev)
lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev)
alpha = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
import time
for _ in range(6):
    step(lr, hr, alpha)
torch.cuda.synchronize()
step.timeline.clear()
t0 = time.perf_counter()
N = 10
for _ in range(N):
    step(lr, hr, alpha)
host = (time.perf_counter() - t0) / N * 1e3
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
by = {}
for call, name, ev in step.timeline:
    by.setdefault(call, {})[name] = ev
calls = sorted(by)[2:]
names = list(by[calls[0]].keys())
print('host enqueue %.1f ms per step, wall %.1f ms per step' % (host, wall))
for n in names:
    v = [by[c]['start'].elapsed_time(by[c][n]) for c in calls]
    print('%-48s +%6.2f ms' % (n, sum(v) / len(v)))
v = [by[a]['start'].elapsed_time(by[b]['start']) for a, b in zip(calls[:-1], calls[1:])]
print('%-48s  %6.2f ms' % ('step to step', sum(v) / len(v)))
