"""Round 4: what ONE conv geometry gets inside the training step (srhip_probe_*: HIP events around every matching call on its own
launch stream while the other two streams of the step share the chip).
  KIND=2 CIN=256 COUT=64 python tools/probe_step.py     # conv2's dgrad (64 -> 256 data gradient with the activation mask)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sradsgan_amd import _hip
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
B = 32
G, D, F = bench.build_networks(dev, 20240)
step = TrainStep(G, D, F)
gen = torch.Generator().manual_seed(1)
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev); lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev); al = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
for _ in range(15): step(lr, hr, al)
torch.cuda.synchronize()
lib = _hip.lib()
for kind, cin, cout in [tuple(int(v) for v in k.split(':')) for k in os.environ.get('PROBES', '1:64:256,2:256:64,1:256:64,2:64:256,3:64:256,3:256:64').split(',')]:
    lib.srhip_probe_config(kind, B, 54, 54, cin, cout, 1024)
    for _ in range(4): step(lr, hr, al)
    torch.cuda.synchronize()
    ms = (ctypes.c_float * 1024)(); un = (ctypes.c_int * 1024)()
    n = lib.srhip_probe_read(ms, un, 1024)
    lib.srhip_probe_config(0, 0, 0, 0, 0, 0, 0)
    convs = sum(un[i] for i in range(n))
    print('kind %d (1 fprop, 2 dgrad, 3 wgrad) conv %3d -> %3d: %4d calls, %.1f us per convolution in the step' % (kind, cin, cout, n, sum(ms[i] for i in range(n)) / max(convs, 1) * 1e3), flush=True)
