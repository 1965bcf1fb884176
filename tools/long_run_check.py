"""Long-run reproducibility of the default step (three streams, C-side forks, grouped weight gradients): two independent runs of 150
iterations at the bench shape (B = 32, fresh random batch every step from a seeded generator) must give bit-identical loss
trajectories and final weights -- a race between streams (an event ring wrapping too early, a grouped launch forked behind the
wrong producer) would show up as a difference somewhere along 150 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
import bench
from sradsgan_amd.train_step import TrainStep
dev = torch.device('cuda:0')
N = int(os.environ.get('STEPS', '150'))


def run(sync=None):
    G, D, F = bench.build_networks(dev, 20240)
    step = TrainStep(G, D, F, grad_sync=sync)
    gen = torch.Generator().manual_seed(7)
    traj = []
    for it in range(N):
        hr = torch.rand(32, 3, 216, 216, generator=gen).to(dev)
        lr = torch.rand(32, 3, 54, 54, generator=gen).to(dev)
        al = torch.rand(32, 1, 1, 1, generator=gen).to(dev)
        out = step(lr, hr, al)
        traj.append(torch.stack([out[k].double() for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')]))
    torch.cuda.synchronize()
    w = torch.cat([step.arena_G.flat_p, step.arena_D.flat_p]).clone()
    return torch.stack(traj).cpu(), w.cpu()


a_t, a_w = run()
b_t, b_w = run()
bad = (a_t != b_t).any(dim=1).nonzero().flatten().tolist()
print('steps: %d, finite: %s, first differing step: %s, weights identical: %s' % (N, bool(torch.isfinite(a_t).all()), bad[:1] or None, bool(torch.equal(a_w, b_w))))
print('last losses', a_t[-1].tolist())
if os.environ.get('RCCL') == '1':       # round 6: the same 150 iterations with the gradient exchange on a single-rank RCCL communicator (the generator's
    from sradsgan_amd import dp         # arena in parts from inside the backward, enqueue thread): a one-rank sum is the identity, so the trajectory must not move
    sync = dp.GradSync(1, force=True)
    c_t, c_w = run(sync)
    sync.close()
    badc = (a_t != c_t).any(dim=1).nonzero().flatten().tolist()
    print('forced single-rank RCCL exchange: first differing step: %s, weights identical: %s, parts of the last step: %s'
          % (badc[:1] or None, bool(torch.equal(a_w, c_w)), [(t, p) for t, p, _, _ in sync.parts[-5:]]))
    if badc or not torch.equal(a_w, c_w):
        sys.exit(1)
sys.exit(0 if (not bad and torch.equal(a_w, b_w) and torch.isfinite(a_t).all()) else 1)
