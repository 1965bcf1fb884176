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
sync = None
if os.environ.get('BENCH_FORCE_DIST') == '1':       # the exchange on a single-rank RCCL communicator: when does each part of the arenas leave / arrive?
    from sradsgan_amd import dp
    sync = dp.GradSync(1, force=True)
    sync.timing = True
step = TrainStep(G, D, F, grad_sync=sync)
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
if sync is not None:
    sync.done_log.clear()
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

if sync is not None:
    # done_log holds (tag, part, event) in issue order: N steps x parts per step
    per = len(sync.done_log) // N
    print('exchange (single-rank RCCL communicator, %s): completion of each part on the comm stream' % ('enqueue thread' if sync._enqueuer is not None else 'issued by the caller'))
    for j in range(per):
        tag, part, _ = sync.done_log[j]
        v = [by[c]['start'].elapsed_time(sync.done_log[i * per + j][2]) for i, c in enumerate(sorted(by)) if c in calls]
        lo_n = [(lo, n) for t, p, lo, n in sync.parts if t == tag and p == part]
        mb = lo_n[0][1] * 4 / 1e6 if lo_n else (step.arena_D.numel * 4 / 1e6 if tag == 'D' else float('nan'))
        print('%-48s +%6.2f ms   (%.1f MB)' % ('%s part %s exchanged' % (tag, part), sum(v) / len(v), mb))
