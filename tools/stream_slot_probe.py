"""Does the step time depend on WHICH pool streams TrainStep receives (torch hands out its 32 pool streams round robin; ROCm maps
streams onto a few hardware queues)?  Burn k streams first, then time the x3 step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import model as M
from sradsgan_amd.train_step import TrainStep
from sradsgan_amd.trainer import weights_init_normal
dev = torch.device('cuda:0')
burn = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sc = int(sys.argv[2]) if len(sys.argv) > 2 else 3
keep = [torch.cuda.Stream(device=dev) for _ in range(burn)]
G = M.GeneratorResNet(M.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=sc)
D, Fx = M.Discriminator(), M.FeatureExtractor()
G.apply(weights_init_normal), D.apply(weights_init_normal)
for m in (G, D, Fx):
    m.to(dev)
step = TrainStep(G, D, Fx)
print('burned %d; wgrad stream %#x, D stream %#x' % (burn, step._wgrad_stream.cuda_stream, step._d_stream.cuda_stream))
B, side = 32, 216 // sc
hr = torch.rand(B, 3, side * sc, side * sc, device=dev)
lr = torch.rand(B, 3, side, side, device=dev)
al = torch.rand(B, 1, 1, 1, device=dev)
for _ in range(4):
    step(lr, hr, al)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    step(lr, hr, al)
torch.cuda.synchronize()
print('x%d: %.2f ms per step' % (sc, (time.perf_counter() - t0) / 8 * 1e3))
