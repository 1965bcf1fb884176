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


def _order_worker(rank, world, port, out_dir):
    """The ordering TrainStep drives (train_step.TrainStep._exchange_start / _update) on the object it uses (dp.GradSync):
    the G arena goes out after the generator's backward and is in flight while the D step computes, the D arena after
    the discriminator's backward; each Adam waits for its own arena only."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sradsgan_amd.dp import GradSync, ParamArena, broadcast_module
    torch.manual_seed(7)
    netG = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(8, 3, 3))
    netD = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 1, 3))
    broadcast_module(netG, 0), broadcast_module(netD, 0)
    aG, aD = ParamArena(netG), ParamArena(netD)
    sync = GradSync(world, bucket_bytes=512)
    torch.manual_seed(50 + rank)
    x = torch.randn(2, 3, 12, 12)
    netG(x).abs().mean().backward()                        # "generator backward"
    localG = aG.flat_g.clone()
    sync.start('G', aG.flat_g)                             # TrainStep._exchange_start('G')
    try:
        sync.start('G', aG.flat_g)                         # a second start before finish is a bug in the caller
        dup = False
    except RuntimeError:
        dup = True
    netD(x).mean().backward()                              # "discriminator step" while G's buckets travel
    localD = aD.flat_g.clone()
    sync.start('D', aD.flat_g)                             # TrainStep._exchange_start('D')
    sync.finish('G')                                       # TrainStep._update: before Adam(G)
    gG = [torch.empty_like(localG) for _ in range(world)]
    dist.all_gather(gG, localG)
    ok = dup and torch.allclose(aG.flat_g * sync.grad_scale, sum(gG) / world, rtol=1e-6, atol=1e-7)
    sync.finish('D')                                       # before Adam(D)
    gD = [torch.empty_like(localD) for _ in range(world)]
    dist.all_gather(gD, localD)
    ok = ok and torch.allclose(aD.flat_g * sync.grad_scale, sum(gD) / world, rtol=1e-6, atol=1e-7)
    ok = ok and sync.trace == [('start', 'G'), ('start', 'G'), ('start', 'D'), ('finish', 'G'), ('finish', 'D')]
    ok = ok and not sync._pending
    open(os.path.join(out_dir, 'ord%d' % rank), 'w').write('1' if ok else '0')
    dist.destroy_process_group()


def test_exchange_ordering_world2(tmp_path):
    port = _free_port()
    mp.spawn(_order_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(os.path.join(str(tmp_path), 'ord%d' % r)).read() for r in range(2)] == ['1', '1']


def _parts_worker(rank, world, port, out_dir):
    """Round 6: the generator's arena leaves in parts, in reverse layer order, driven by TrainStep's own plan and hand-over code
    (train_step.TrainStep._plan_g_parts / _bucket_ready / _exchange_start) on the real generator's parameter layout: the parts are
    disjoint, cover the arena exactly once, go out late layers first, in the same order on every rank, and finish('G') leaves the
    sum over ranks everywhere."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sradsgan_amd import model as M
    from sradsgan_amd import train_step as ts
    from sradsgan_amd.dp import GradSync, ParamArena, broadcast_module
    torch.manual_seed(3)
    G = M.GeneratorResNet(M.ResGroup, n_residual_blocks=6, n_basic_blocks=1, upscale_factor=2)
    broadcast_module(G, 0)
    step = ts.TrainStep.__new__(ts.TrainStep)
    step.G, step.arena_G = G, ParamArena(G)
    step.grad_sync = sync = GradSync(world, bucket_bytes=64 << 10)
    step._capturing, step.overlap_wgrad, step._wgrad_stream, step._timeline_on = False, False, None, False
    step._g_parts, step._g_rest = step._plan_g_parts(3, any_device=True)
    ok = len(step._g_parts) == 3 and [k for k, _ in step._part_groups] == [0, 1, 2]
    names = {id(p): n for n, p in G.named_parameters()}
    first = lambda lo: names[id(step.arena_G.params[step.arena_G.offsets.index(lo)])]
    ok = ok and [first(lo) for lo, _ in step._g_parts] == ['res_groups.4.RG.0.conv1.weight', 'res_groups.2.RG.0.conv1.weight', 'res_groups.0.RG.0.conv1.weight']
    last0 = max(o for o in step.arena_G.offsets if o < step._g_parts[0][1])
    ok = ok and names[id(step.arena_G.params[step.arena_G.offsets.index(last0)])].startswith('GAB_UP.')     # the up-sampler rides with the last groups
    flat = step.arena_G.flat_g
    torch.manual_seed(20 + rank)
    flat.copy_(torch.randn(flat.numel()))                  # "the backward's result", different on every rank
    local = flat.clone()
    step._parts_sent, step._parts_armed = set(), True
    for k in range(3):                                     # the tensor hooks fire in this order: late groups first
        step._bucket_ready(k)
    step._bucket_ready(1)                                  # a hook that fires twice hands nothing over twice
    step._exchange_start('G')                              # the end of the backward: what is left
    sync.finish('G')
    both = [torch.empty_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    ok = ok and torch.allclose(flat, sum(both), rtol=1e-6, atol=1e-6)
    spans = sorted((lo, lo + n) for _, _, lo, n in sync.parts)
    ok = ok and spans[0][0] == 0 and spans[-1][1] == flat.numel() and all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
    ok = ok and [p for _, p, _, _ in sync.parts] == [0, 1, 2, 3, 4]
    ok = ok and [lo for _, _, lo, _ in sync.parts][:3] == sorted([lo for _, _, lo, _ in sync.parts][:3], reverse=True)
    order = [None] * world
    dist.all_gather_object(order, sync.parts)
    ok = ok and order[0] == order[1] and not sync._pending
    open(os.path.join(out_dir, 'parts%d' % rank), 'w').write('1' if ok else '0')
    dist.destroy_process_group()


def test_generator_arena_leaves_in_reverse_layer_order_world2(tmp_path):
    port = _free_port()
    mp.spawn(_parts_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(os.path.join(str(tmp_path), 'parts%d' % r)).read() for r in range(2)] == ['1', '1']


def _id_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sradsgan_amd.dp import share_unique_id
    import time
    ids = []
    for generation in range(2):                              # a second communicator must not pick up the first one's id
        if rank == 0:
            time.sleep(0.3)                                  # the other rank is already blocked in store.get
        ids.append(share_unique_id(lambda: os.urandom(128), rank, world, generation))
    mine = torch.tensor([list(i) for i in ids], dtype=torch.uint8)
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    ok = torch.equal(both[0], both[1]) and ids[0] != ids[1] and all(len(i) == 128 for i in ids)
    open(os.path.join(out_dir, 'id%d' % rank), 'w').write('1' if ok else '0')
    dist.destroy_process_group()


def test_communicator_id_hand_off_world2(tmp_path):
    """dp.share_unique_id: how the RCCL communicator id travels from rank 0 to the others (rendezvous store, generation-keyed) --
    the one part of GradSync.init_rccl that only runs with more than one rank."""
    port = _free_port()
    mp.spawn(_id_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(os.path.join(str(tmp_path), 'id%d' % r)).read() for r in range(2)] == ['1', '1']


def test_train_step_drives_the_exchange_in_that_order():
    """TrainStep's own call sequence, checked without a GPU: _exchange_start / _update call GradSync.start('G'),
    start('D'), finish('G'), finish('D') in this order (a recording stand-in replaces the arenas and kernels)."""
    import types
    from sradsgan_amd import train_step as ts
    from sradsgan_amd.dp import GradSync
    calls = []
    sync = GradSync(2)
    sync.start = lambda tag, flat, after=(), part=None, events=None: calls.append(('start', tag))
    sync.finish = lambda tag: calls.append(('finish', tag))
    step = ts.TrainStep.__new__(ts.TrainStep)
    step.grad_sync, step._capturing, step.use_graph, step._graph = sync, False, False, None
    step.overlap_wgrad, step._wgrad_stream = False, None
    step.arena_G = types.SimpleNamespace(flat_g=torch.zeros(4))
    step.arena_D = types.SimpleNamespace(flat_g=torch.zeros(4))
    step.lr_G = step.lr_D = 1e-4
    step.clip_value = 0.01
    step._adam = lambda arena, lr, clip, scale: calls.append(('adam', 'G' if arena is step.arena_G else 'D', scale))
    orig = ts.ops.bump_weight_epoch
    ts.ops.bump_weight_epoch = lambda: None
    try:
        step._exchange_start('G')
        step._exchange_start('D')
        step._update()
    finally:
        ts.ops.bump_weight_epoch = orig
    assert calls == [('start', 'G'), ('start', 'D'), ('finish', 'G'), ('adam', 'G', 0.5), ('finish', 'D'), ('adam', 'D', 0.5)]


def _control_worker(rank, world, port, out_dir):
    """The trainer's data-parallel control plane (sradsgan_amd/trainer.py under torch.distributed): rank 0's validation
    numbers reach every rank (NaN included), so every rank replays the same PlateauRollback sequence; shards of one
    permutation are disjoint, equal-sized and cover the prefix DataLoader(drop_last) would keep."""
    import math
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sradsgan_amd import dp
    from sradsgan_amd.checkpoint import PlateauRollback
    ok = dp.rank_world() == (rank, world)
    control = PlateauRollback(2e-4)
    epoch, trace = 0, []
    r0 = [20.0, 21.0, 20.5, 20.4, 20.3, 20.2, 20.1, 20.0, 19.9, 22.0]      # psnr, psnr, ssim, ergas improve, then 5 misses
    psnr_by_rank = [r0, [25.0, 19.0, 26.0, 18.0, 27.0, 17.0, 28.0, 16.0, 29.0, 15.0]][rank]
    for i in range(10):
        local = (psnr_by_rank[i], 0.5, 9.0, float('nan'))                 # ranks disagree on purpose; lpips slot is NaN
        val = dp.broadcast_floats(local, 0)
        ok = ok and val[0] == r0[i] and math.isnan(val[3])
        epoch, rolled = control.update(epoch, val[0], val[1], val[2], 10000 if math.isnan(val[3]) else val[3])
        trace.append((epoch, rolled, control.lr))
        dp.barrier()
    gathered = [None] * world
    dist.all_gather_object(gathered, trace)
    ok = ok and gathered[0] == gathered[1] and sum(r for _, r, _ in trace) == 1                 # same decisions, one rollback
    g = torch.Generator().manual_seed(0)
    order = torch.randperm(11, generator=g).tolist()
    mine = dp.shard_indices(order, rank, world)
    shards = [None] * world
    dist.all_gather_object(shards, mine)
    ok = ok and len(shards[0]) == len(shards[1]) == 5 and not set(shards[0]) & set(shards[1])
    ok = ok and sorted(shards[0] + shards[1]) == sorted(order[:10])
    open(os.path.join(out_dir, 'ctl%d' % rank), 'w').write('1' if ok else '0')
    dist.destroy_process_group()


def test_trainer_control_plane_world2(tmp_path):
    port = _free_port()
    mp.spawn(_control_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert [open(os.path.join(str(tmp_path), 'ctl%d' % r)).read() for r in range(2)] == ['1', '1']
