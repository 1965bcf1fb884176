"""Worker of tests/test_dp_world2_gpu.py: rank R of a 2-rank job on GPU R.  Checks the gradient exchange of TrainStep + dp.GradSync
(RCCL through srhip_dp_*, its own stream, event ordering) against an all-reduce of the local gradients done by torch.distributed,
and that replicas stay identical over a few real steps.  Exit code 0 = all checks passed on this rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')


def main():
    import torch
    import torch.distributed as dist
    from oracle import sradsgan_ref as O
    from sradsgan_amd import dp
    from sradsgan_amd.train_step import TrainStep
    from tests.parity_util import build_pair
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=world)

    def batch(it):
        tag = 'dp2.r%d' % rank                                   # disjoint shards
        return (O.det_fill('%s.lr.%d' % (tag, it), (4, 3, 24, 24), 0.5, 0.5).to(dev),
                O.det_fill('%s.hr.%d' % (tag, it), (4, 3, 96, 96), 0.5, 0.5).to(dev),
                O.det_fill('%s.alpha.%d' % (tag, it), (4, 1, 1, 1), 0.5, 0.5).to(dev))

    # (1) one step with frozen weights (lr 0, no clip): the arenas then hold the exchanged gradients = SUM over ranks
    (g1, d1, f1), _ = build_pair(2, 2, 4, dev)
    sync = dp.GradSync(world)
    step = TrainStep(g1, d1, f1, lr=0.0, clip_value=0.0, grad_sync=sync)
    step(*batch(0))
    torch.cuda.synchronize()
    assert sync.rccl_ranks() == world
    got = [step.arena_G.flat_g.clone(), step.arena_D.flat_g.clone()]
    (g2, d2, f2), _ = build_pair(2, 2, 4, dev)                   # the same replica without the exchange: local gradients
    plain = TrainStep(g2, d2, f2, lr=0.0, clip_value=0.0)
    plain(*batch(0))
    torch.cuda.synchronize()
    for name, have, arena in (('G', got[0], plain.arena_G), ('D', got[1], plain.arena_D)):
        want = arena.flat_g.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM)
        err = float((have - want).abs().max() / want.abs().max().clamp_min(1e-30))
        print('rank %d: %s arena after the exchange vs all-reduced local gradients: rel err %.2e' % (rank, name, err), flush=True)
        assert err < 1e-5, (name, err)
        assert float((have - arena.flat_g).abs().max()) > 0.0    # the shards differ, so the sum is not the local gradient
    # (2) real steps: replicas must stay bit-identical (same summed gradients, same Adam arithmetic on every rank)
    (g3, d3, f3), _ = build_pair(2, 2, 4, dev)
    sync3 = dp.GradSync(world)
    step3 = TrainStep(g3, d3, f3, grad_sync=sync3)
    for it in range(4):
        out = step3(*batch(it))
    torch.cuda.synchronize()
    # from the second iteration on the generator's arena leaves in parts: group 1 + up-sampler, group 0, head, MSB + tail conv
    assert [p for t, p, _, _ in sync3.parts if t == 'G'][:4] == [0, 1, 2, 3], sync3.parts
    for arena in (step3.arena_G, step3.arena_D):
        mine = arena.flat_p.clone()
        theirs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(theirs, mine)
        for t in theirs:
            assert torch.equal(t, mine)
    assert all(float(out[k]) == float(out[k]) for k in ('loss_G', 'loss_D'))
    dist.barrier()
    sync.close()
    sync3.close()
    dist.destroy_process_group()
    print('rank %d ok' % rank, flush=True)


if __name__ == '__main__':
    main()
