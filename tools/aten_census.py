"""Which ATen kernels are left in one training step, with input shapes (torch profiler, CPU-side op records): the element-wise passes
autograd or the step's Python adds around the library's launches.  usage: aten_census.py"""
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
for _ in range(3): step(lr, hr, al)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    step(lr, hr, al)
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.name in ('aten::empty', 'aten::empty_like', 'aten::empty_strided', 'aten::view', 'aten::as_strided', 'aten::permute',
                                                        'aten::detach', 'aten::alias', 'aten::reshape', 'aten::contiguous', 'aten::select', 'aten::slice',
                                                        'aten::_unsafe_view', 'aten::unsqueeze', 'aten::squeeze', 'aten::expand', 'aten::t', 'aten::transpose',
                                                        'aten::to', 'aten::lift_fresh', 'aten::is_nonzero', 'aten::item', 'aten::resize_', 'aten::narrow', 'aten::view_as'):
        continue
    key = (ev.name, str([s for s in ev.input_shapes if s]))
    rows[key] = rows.get(key, 0) + 1
for (name, shp), cnt in sorted(rows.items(), key=lambda kv: -kv[1]):
    print('%4d  %-28s %s' % (cnt, name, shp[:150]))
