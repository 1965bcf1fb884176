"""Data-parallel plumbing for the training step: flat parameter/gradient arenas and the gradient
all-reduce.  The reference is single-GPU (README.md:91); this is the one exchange
step the sharded path needs: one process per GPU, full replicas, mean of the G and D gradients over
ranks before each Adam step.  Backend is whatever torch.distributed was initialised with -- "nccl"
(= RCCL over xGMI) on MI355X, "gloo" in the CPU tests.

Layout: every network keeps ONE flat fp32 arena each for parameters, gradients and the two Adam
moments (ParamArena).  Parameters and .grad tensors are views into the arenas, so
  * zero_grad is one memset, Adam(+clip) is one elementwise HIP kernel over the arena,
  * the all-reduce runs in place on contiguous arena slices (buckets) -- no flatten/unflatten copies.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): G's 44.3 MB and D's 18.8 MB of gradients are
sent as <= 32 MiB buckets so the second bucket's reduce-scatter overlaps the first one's all-gather.
"""
import torch
import torch.distributed as dist

ALIGN = 64                      # elements; keeps every parameter view 256-byte aligned in the arena


class ParamArena:
    """Flat storage for one network's parameters, gradients and Adam state (sradsgan.py:724-725)."""

    def __init__(self, module):
        seen, params = set(), []
        for p in module.parameters():
            if id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError('ParamArena: module has no parameters')
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise TypeError('ParamArena: fp32 parameters on one device only')
            offs.append(total)
            total += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.params, self.offsets, self.numel = params, offs, total
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        # [step, 1 - b1^step, 1 - b2^step] kept on the device so a captured hipGraph can advance it
        self.step_state = torch.zeros(4, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = self.flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)

    def zero_grad(self):
        self.flat_g.zero_()

    def check_views(self):
        """True while every parameter and gradient still aliases the arena (module.to()/zero_grad(
        set_to_none=True) would silently break that)."""
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != self.flat_p.data_ptr() + 4 * o:
                return False
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * o:
                return False
        return True


class GradSync:
    """Mean of a gradient arena over the ranks: bucketed in-place all-reduce(SUM), asynchronous per
    bucket, then one scale by 1/world (folded into the Adam kernel by the caller when it can)."""

    def __init__(self, world_size=None, bucket_bytes=32 << 20, group=None, force=False):
        self.group = group
        self.force = force           # run the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.bucket_elems = max(1, bucket_bytes // 4)

    def buckets(self, flat):
        n = flat.numel()
        return [flat[i:min(i + self.bucket_elems, n)] for i in range(0, n, self.bucket_elems)]

    def start(self, flat):
        """Launch the all-reduces; returns handles for finish().  `flat` must be the gradient arena
        (contiguous 1-D).  With world == 1 this is a no-op."""
        if self.world <= 1 and not self.force:
            return []
        if not dist.is_initialized():
            raise RuntimeError('GradSync: torch.distributed is not initialised')
        return [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for b in self.buckets(flat)]

    def finish(self, handles):
        for h in handles:
            h.wait()

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def __call__(self, flat):
        """Synchronous convenience form: flat <- mean over ranks."""
        self.finish(self.start(flat))
        if self.world > 1:
            flat.mul_(self.grad_scale)
        return flat


def broadcast_module(module, src=0, group=None):
    """Identical replicas: parameters and buffers (BN running stats) from rank `src`."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)


def rank_world(group=None):
    """(rank, world) of the default process group, (0, 1) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def broadcast_floats(values, src=0, device=None, group=None):
    """The python floats `values` as rank `src` holds them, on every rank (validation metrics / rollback decisions are
    taken from rank 0).  NaN survives.  No-op with one rank."""
    rank, world = rank_world(group)
    vals = [float(v) for v in values]
    if world == 1:
        return vals
    t = torch.tensor(vals, dtype=torch.float64, device=device)
    dist.broadcast(t, src=src, group=group)
    return [float(v) for v in t.cpu()]


def shard_indices(order, rank, world):
    """Rank `rank`'s share of the (already shuffled, identical on all ranks) index list: every world-th index,
    truncated so that all ranks get the same count (the tail is dropped like DataLoader's drop_last)."""
    per = len(order) // world
    return order[rank:per * world:world]


def barrier(group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.barrier(group)
