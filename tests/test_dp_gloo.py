"""world_size-2 CPU (gloo) test of the gradient exchange used by the N>1 path: bucketed in-place
all-reduce of a gradient arena gives the mean over ranks, identical on every rank."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sradsgan_amd.dp import GradSync, ParamArena, broadcast_module
    torch.manual_seed(100 + rank)                          # different replicas on purpose
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 3))
    broadcast_module(net, src=0)
    arena = ParamArena(net)
    x = torch.randn(4, 3, 10, 10)                          # disjoint data shard per rank
    net(x).square().mean().backward()
    local = arena.flat_g.clone()
    sync = GradSync(world, bucket_bytes=1024)              # force several buckets
    assert len(sync.buckets(arena.flat_g)) > 1
    sync(arena.flat_g)
    gathered = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    want = sum(gathered) / world
    ok = torch.allclose(arena.flat_g, want, rtol=1e-6, atol=1e-7) and arena.check_views()
    p0 = [torch.empty_like(arena.flat_p) for _ in range(world)]
    dist.all_gather(p0, arena.flat_p)
    ok = ok and all(torch.equal(p0[0], t) for t in p0)     # broadcast made the replicas identical
    open(os.path.join(out_dir, 'ok%d' % rank), 'w').write('1' if ok else '0')
    dist.destroy_process_group()


def test_gradsync_mean_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(os.path.join(str(tmp_path), 'ok%d' % r)).read() for r in range(2)] == ['1', '1']
