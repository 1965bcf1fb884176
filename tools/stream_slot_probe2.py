"""Step time for every (wgrad stream, D stream) pair out of the first six torch pool streams, in ONE process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import model as M
from sradsgan_amd.train_step import TrainStep
from sradsgan_amd.trainer import weights_init_normal
dev = torch.device('cuda:0')
sc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NS = int(os.environ.get('NS', '6'))
pool = [torch.cuda.Stream(device=dev) for _ in range(NS)]
G = M.GeneratorResNet(M.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=sc)
D, Fx = M.Discriminator(), M.FeatureExtractor()
G.apply(weights_init_normal), D.apply(weights_init_normal)
for m in (G, D, Fx):
    m.to(dev)
B, side = 32, 216 // sc
hr = torch.rand(B, 3, side * sc, side * sc, device=dev)
lr = torch.rand(B, 3, side, side, device=dev)
al = torch.rand(B, 1, 1, 1, device=dev)
res = {}
for i in range(NS):
    for j in range(NS):
        if i == j:
            continue
        step = TrainStep(G, D, Fx, wgrad_stream=pool[i], d_stream=pool[j])
        for _ in range(3):
            step(lr, hr, al)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step(lr, hr, al)
        torch.cuda.synchronize()
        res[(i, j)] = (time.perf_counter() - t0) / 5 * 1e3
        del step
print('x%d ms per step; rows = wgrad stream slot, columns = D stream slot' % sc)
for i in range(NS):
    print(i, ' '.join(('%7.1f' % res[(i, j)]) if i != j else '      -' for j in range(NS)))
