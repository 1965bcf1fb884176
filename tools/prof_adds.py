"""Round 4: which ATen kernels (adds, muls, copies, fills ...) the default training step still launches next to the srhip_* ones,
with shapes and counts (torch.profiler over two steps at the bench shape)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
B = 32
G, D, F = bench.build_networks(dev, 20240)
step = TrainStep(G, D, F)
gen = torch.Generator().manual_seed(1)
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev); lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev); al = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
for _ in range(8): step(lr, hr, al)
torch.cuda.synchronize()
NS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(NS): step(lr, hr, al)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0:
        continue
    shapes = str([s for s in (ev.input_shapes or []) if s])[:90]
    k = (ev.name, shapes)
    agg[k][0] += 1
    agg[k][1] += ev.self_device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print('ATen ops with device time, per step (%d steps profiled): %.2f ms of kernel time per step' % (NS, tot / NS / 1e3))
for (name, shapes), (n, t) in rows[:45]:
    print('%6.1f calls/step %8.1f us/step  %-22s %s' % (n / NS, t / NS, name, shapes))
