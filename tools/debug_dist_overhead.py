import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench
from sradsgan_amd.train_step import TrainStep
from sradsgan_amd import dp
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29545')
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
mode = sys.argv[1]
overlap = os.environ.get('OVERLAP', '1') == '1'
late = os.environ.get('LATE_INIT', '0') == '1'
sync = None
def init():
    if os.environ.get('NO_DEVICE_ID') == '1':
        dist.init_process_group('nccl', rank=0, world_size=1)
        os.environ['X']='1'
    else:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
if mode != 'plain' and not late:
    init()
    if mode in ('allreduce', 'tiny'):
        sync = dp.GradSync(1, force=True)
        if mode == 'tiny':
            sync.buckets = lambda flat: [flat[:16]]
G, D, F = bench.build_networks(dev, seed=1)
step = TrainStep(G, D, F, grad_sync=sync, overlap_wgrad=overlap)
hr = torch.rand(32, 3, 216, 216, device=dev); lr = torch.rand(32, 3, 54, 54, device=dev); alpha = torch.rand(32, 1, 1, 1, device=dev)
for _ in range(3): step(lr, hr, alpha)
if mode != 'plain' and late:
    init()
    step(lr, hr, alpha)
torch.cuda.synchronize()
cpu = []
t0 = time.perf_counter()
for _ in range(6):
    c0 = time.perf_counter(); step(lr, hr, alpha); cpu.append(time.perf_counter() - c0)
    if os.environ.get('SYNC_EACH') == '1':
        torch.cuda.synchronize()
torch.cuda.synchronize()
print('%-10s overlap=%d late=%d wall %.1f ms/step, CPU time inside step(): %s ms' % (mode, overlap, late, (time.perf_counter() - t0) / 6 * 1e3, ' '.join('%.0f' % (c * 1e3) for c in cpu)), flush=True)
