"""Step time of a sequence of scales in ONE process (fresh model per scale, the same two side streams for all): does a scale's
time depend on what ran before it?  usage: seq_probe.py 2,3  |  3,3  |  4,3 ..."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sradsgan_amd import model as M, ops
from sradsgan_amd.train_step import TrainStep
from sradsgan_amd.trainer import weights_init_normal
dev = torch.device('cuda:0')
pool = [torch.cuda.Stream(device=dev) for _ in range(2)]
for sc in [int(v) for v in sys.argv[1].split(',')]:
    G = M.GeneratorResNet(M.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=sc)
    D, Fx = M.Discriminator(), M.FeatureExtractor()
    G.apply(weights_init_normal), D.apply(weights_init_normal)
    for m in (G, D, Fx):
        m.to(dev)
    step = TrainStep(G, D, Fx, wgrad_stream=pool[0], d_stream=pool[1])
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
    host = (time.perf_counter() - t0) / 8 * 1e3
    torch.cuda.synchronize()
    print('x%d: %.2f ms per step (host enqueue %.1f ms), registry entries %d, reserved %.1f GB' % (
        sc, (time.perf_counter() - t0) / 8 * 1e3, host, len(ops._registry.entries), torch.cuda.memory_reserved() / 2 ** 30), flush=True)
    del step, G, D, Fx, hr, lr, al
    gc.collect()
    if os.environ.get('EMPTY') == '1':
        torch.cuda.empty_cache()
