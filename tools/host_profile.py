"""Where the host spends its time in one eager training step (cProfile over 6 steps at the bench shape)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
B = int(os.environ.get('B', '32'))
G, D, F = bench.build_networks(dev, 20240)
step = TrainStep(G, D, F, use_graph=os.environ.get('GRAPH') == '1')
step.max_run_ahead = 0          # measure the host alone: no waiting for the GPU inside the step
gen = torch.Generator().manual_seed(1)
hr = torch.rand(B, 3, 216, 216, generator=gen).to(dev); lr = torch.rand(B, 3, 54, 54, generator=gen).to(dev); al = torch.rand(B, 1, 1, 1, generator=gen).to(dev)
for _ in range(8): step(lr, hr, al)
torch.cuda.synchronize()
# host-only enqueue time: how long does the host need to enqueue one step when the GPU is not the limit?
t0 = time.perf_counter()
for _ in range(6): step(lr, hr, al)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('6 steps: host enqueue %.1f ms/step, until GPU idle %.1f ms/step' % ((t1 - t0) / 6 * 1e3, (t2 - t0) / 6 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(6): step(lr, hr, al)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(28)
