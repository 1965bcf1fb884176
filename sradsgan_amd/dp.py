"""Data-parallel plumbing for the training step: flat parameter/gradient arenas and the gradient
all-reduce.  The reference is single-GPU (README.md:91); this is the one exchange
step the sharded path needs: one process per GPU, full replicas, mean of the G and D gradients over
ranks before each Adam step.  On MI355X the collectives are RCCL calls over xGMI through the C ABI
(srhip_dp_*, include/sradsgan_hip.h) on their own HIP stream; torch.distributed provides the rendezvous
(and, with "gloo", the whole exchange in the CPU tests).

Layout: every network keeps ONE flat fp32 arena each for parameters, gradients and the two Adam
moments (ParamArena).  Parameters and .grad tensors are views into the arenas, so
  * zero_grad is one memset, Adam(+clip) is one elementwise HIP kernel over the arena,
  * the all-reduce runs in place on contiguous arena slices (buckets) -- no flatten/unflatten copies.
"""
import torch
import torch.distributed as dist

ALIGN = 64                      # elements; keeps every parameter view 256-byte aligned in the arena

# The RCCL communicator behind srhip_dp_* is ONE per process (csrc/dp_rccl.hip): GradSync objects share it.  _COMM_USERS counts
# the objects that hold it (the last close() finalises it), _COMM_GENERATION numbers the communicators this process has
# created so that the unique id of a destroyed communicator is never picked up from the rendezvous store again.
_COMM_USERS = 0
_COMM_GENERATION = 0


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


def share_unique_id(make_id, rank, world, generation):
    """The communicator id as every rank must pass it to srhip_dp_init: rank 0 draws it (make_id() -> bytes) and publishes it
    in torch.distributed's rendezvous store under a key that carries the communicator's generation -- every rank creates its
    communicators in the same order, so the key of a destroyed communicator is never read again --; the other ranks block in
    store.get until it is there.  No collective, no second process group."""
    if world <= 1:
        return make_id()
    store = dist.distributed_c10d._get_default_store()
    key = 'srhip_dp_unique_id/%d' % generation
    if rank == 0:
        raw = make_id()
        store.set(key, raw)
        return raw
    return bytes(store.get(key))


class _Enqueuer:
    """One host thread that issues the collectives in FIFO order (device path).  An RCCL enqueue can hold its calling thread until the
    comm stream's event waits have been satisfied on the GPU (observed with a single-rank communicator in round 3: issued from the main
    thread in the middle of a backward pass it stalled the launch queue behind it) -- from this thread the wait costs nothing: the
    thread that enqueues the step never blocks.  Every rank submits its buckets in the same program order and the thread keeps it."""

    def __init__(self, device):
        import queue
        import threading
        self.q = queue.Queue()
        self.device = device
        self.thread = threading.Thread(target=self._run, name='srhip-dp-enqueue', daemon=True)
        self.thread.start()

    def _run(self):
        torch.cuda.set_device(self.device)
        while True:
            item = self.q.get()
            if item is None:
                return
            fn, ticket = item
            try:
                ticket.result = fn()
            except BaseException as e:                 # surfaces in finish() on the calling thread
                ticket.error = e
            ticket.done.set()

    def submit(self, fn):
        import threading
        ticket = type('Ticket', (), {})()
        ticket.done, ticket.result, ticket.error = threading.Event(), None, None
        self.q.put((fn, ticket))
        return ticket

    def stop(self):
        self.q.put(None)
        self.thread.join(timeout=30.0)


class GradSync:
    """The per-iteration gradient exchange and its ordering: `start(tag, flat, ...)` launches the in-place all-reduce(SUM) of a
    gradient arena -- or, with `part=`, of ONE SLICE of it: an arena may leave in several parts, each as soon as its producers are
    enqueued --, `finish(tag)` makes the consumer (the Adam kernel of that network) wait for every part; the 1/world of the mean is
    folded into the Adam kernel by the caller.

    What TrainStep does with it (round 6, train_step._bucket_ready / _exchange_start): the generator's arena leaves in REVERSE LAYER
    ORDER while its backward is still running -- groups 11..8 + the up-sampler when the backward has passed group 8, groups 7..4,
    groups 3..0, then the head / multi-scale block / tail conv at the end --, each part behind an event of the main stream and one of
    the weight-gradient stream taken at that point; the discriminator's arena after the D stream's backward (the second one,
    sradsgan.py:886; the first, :639, accumulates locally); finish('G') / finish('D') in front of the two Adam launches (:858, :887).
    (Rounds 2-5 sent the generator's arena in one piece after its backward: in the default schedule, where the discriminator's
    passes run BESIDE that backward, nothing was left to hide it under.)

    Device tensors: the collectives are RCCL calls through the C ABI (srhip_dp_allreduce_bucket, include/sradsgan_hip.h)
    on a dedicated HIP stream, issued by a helper thread (_Enqueuer; SRHIP_DP_THREAD=0: by the caller); ordering against the
    compute streams is by events only, the host never synchronises.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a part
    goes out as <= 32 MiB buckets so the ring's reduce-scatter of one bucket overlaps the all-gather of the previous one.
    CPU tensors (the gloo tests): the same object drives torch.distributed's asynchronous all_reduce.
    """

    def __init__(self, world_size=None, bucket_bytes=32 << 20, group=None, force=False):
        self.group = group
        self.force = force           # run the collectives even with one rank (exercises the RCCL path on a 1-GPU box)
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.trace = []              # ('start' | 'finish', tag): the order the step drove the exchange in (tests)
        self._pending = {}           # tag -> event (device) or list of work handles (CPU)
        self._deferred = {}          # tag -> flat arena whose exchange waits for finish() (host_sync mode)
        import os
        # Opt-in fallback until a run with more than one rank exists on hardware: nothing is enqueued ahead of
        # its inputs -- start() only remembers the arena, finish() waits on the host for the producing streams, runs the
        # collective and waits for it.  The known-good r1 behaviour; costs the overlap (the exchange is exposed, ~1 ms).
        self.host_sync = os.environ.get('SRHIP_DP_HOST_SYNC') == '1'
        self._comm_stream = None
        self._rccl_ready = False
        self._whole = set()          # tags whose pending exchange is a whole arena (no further part may join it)
        self._enqueuer = None        # _Enqueuer (device path, unless SRHIP_DP_THREAD=0)
        self.parts = []              # (tag, part, first element, elements) of every start() with part=..., in issue order (tests, tools)
        self.timing = False          # True: the completion events of the parts carry timestamps and are kept in done_log (tools/step_timeline.py)
        self.done_log = []           # (tag, part, event on the comm stream)

    @property
    def active(self):
        return self.world > 1 or self.force

    def buckets(self, flat):
        n = flat.numel()
        return [flat[i:min(i + self.bucket_elems, n)] for i in range(0, n, self.bucket_elems)]

    # ---- RCCL communicator behind the C ABI (device path) ------------------------------------------------------ #
    def init_rccl(self, device=None):
        """Creates (or joins) this process's RCCL communicator (srhip_dp_init).  Rank 0 draws the unique id and hands it to
        the other ranks through torch.distributed's rendezvous store (no collective, no second communicator needed), under a
        key that carries the communicator's generation: after a close() the next communicator never reads a stale id."""
        global _COMM_USERS, _COMM_GENERATION
        if self._rccl_ready:
            return
        if self.group is not None:
            raise NotImplementedError('GradSync: the device (RCCL) path runs over the default process group only; a sub-group '
                                      'would need its own communicator and rendezvous key')
        import ctypes
        import os
        from . import _hip
        lib = _hip.lib()
        if device is not None:
            torch.cuda.set_device(device)
        rank = dist.get_rank() if dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_initialized() else 1
        if lib.srhip_dp_world() == 0:
            nbytes = lib.srhip_dp_id_bytes()

            def make_id():
                buf = ctypes.create_string_buffer(nbytes)
                _hip.check(lib.srhip_dp_unique_id(buf), 'dp_unique_id')
                return bytes(buf.raw)
            raw = share_unique_id(make_id, rank, world, _COMM_GENERATION)
            _hip.check(lib.srhip_dp_init(ctypes.create_string_buffer(raw, nbytes), rank, world), 'dp_init')
            _COMM_GENERATION += 1
        if lib.srhip_dp_world() != world:
            raise RuntimeError('GradSync: RCCL communicator has %d ranks, torch.distributed %d' % (lib.srhip_dp_world(), world))
        # HIGH priority: the exchange is enqueued tens of milliseconds ahead of its inputs, i.e. its stream sits on an
        # event wait for most of the step.  Measured on MI355X / ROCm 7.0 (single rank, profiles/r02_bench_n1_rccl_single_rank.json): parked on
        # a normal-priority queue that wait slows the compute streams' kernels by 12 % (75.1 vs 66.6 ms per step, with or
        # without an RCCL call behind it); on a high-priority queue the step costs 66.9 ms (+0.4 %).
        self._comm_stream = torch.cuda.Stream(priority=0 if os.environ.get('SRHIP_DP_PRIO') == '0' else -1)
        if os.environ.get('SRHIP_DP_THREAD', '1') == '1' and not self.host_sync:
            self._enqueuer = _Enqueuer(torch.cuda.current_device())
        self._rccl_ready = True
        _COMM_USERS += 1

    def rccl_ranks(self):
        from . import _hip
        return _hip.lib().srhip_dp_world() if self._rccl_ready else 0

    def close(self):
        """Releases this object's hold on the process's communicator; the last holder finalises it (srhip_dp_finalize)."""
        global _COMM_USERS
        if self._rccl_ready:
            from . import _hip
            if self._enqueuer is not None:
                self._enqueuer.stop()
                self._enqueuer = None
            torch.cuda.synchronize()
            self._rccl_ready = False
            self._pending.clear()
            _COMM_USERS -= 1
            if _COMM_USERS <= 0:
                _COMM_USERS = 0
                _hip.check(_hip.lib().srhip_dp_finalize(), 'dp_finalize')

    # ---- the exchange -------------------------------------------------------------------------------------------- #
    def start(self, tag, flat, after=(), part=None, events=None):
        """Launch the all-reduce of `flat` under the name `tag`.  `after`: the HIP streams whose already enqueued work produces
        `flat` (default: the current stream); `events`: events already recorded on those streams instead.  `part`: None = `flat` is
        the whole arena (a second start before finish is a caller bug); anything else names one SLICE of a bucketed exchange --
        several starts per tag, finish(tag) waits for all of them.  Returns immediately; with one rank (and not forced) a no-op."""
        self.trace.append(('start', tag) if part is None else ('start', tag, part))
        if not self.active:
            return
        if tag in self._pending and (part is None or tag in self._whole):
            raise RuntimeError('GradSync.start(%r): the previous exchange of this arena was never finished' % (tag,))
        if part is None:
            self._whole.add(tag)
        if part is not None:
            self.parts.append((tag, part, flat.storage_offset(), flat.numel()))
        for log in (self.trace, self.parts, self.done_log):         # diagnostics, not history: bounded over a long run
            if len(log) > 8192:
                del log[:-4096]
        if flat.is_cuda and self.host_sync:
            self.init_rccl(flat.device)
            evs = list(events) if events is not None else [s.record_event() for s in (after or (torch.cuda.current_stream(),))]
            self._deferred.setdefault(tag, []).append((flat, evs))
            self._pending.setdefault(tag, [])
            return
        if flat.is_cuda:
            import ctypes
            import os
            from . import _hip
            self.init_rccl(flat.device)
            lib, comm = _hip.lib(), self._comm_stream
            evs = list(events) if events is not None else [s.record_event() for s in (after or (torch.cuda.current_stream(),))]
            fake = os.environ.get('SRHIP_DP_MODE') == 'fake'        # experiment: same stream / event structure, no RCCL call
            timing = self.timing

            def enqueue():
                for ev in evs:
                    comm.wait_event(ev)                             # no host synchronisation
                if fake:
                    with torch.cuda.stream(comm):
                        flat[:64].mul_(1.0)
                else:
                    for b in self.buckets(flat):
                        _hip.check(lib.srhip_dp_allreduce_bucket(ctypes.c_void_p(b.data_ptr()), b.numel(),
                                                                 ctypes.c_void_p(comm.cuda_stream)), 'dp_allreduce_bucket')
                done = torch.cuda.Event(enable_timing=timing)
                done.record(comm)
                if timing:
                    self.done_log.append((tag, part, done))
                return done
            entry = self._enqueuer.submit(enqueue) if self._enqueuer is not None else enqueue()
            self._pending.setdefault(tag, []).append(entry)
        else:
            if not dist.is_initialized():
                raise RuntimeError('GradSync: torch.distributed is not initialised')
            self._pending.setdefault(tag, []).extend(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                                                     for b in self.buckets(flat))

    def finish(self, tag):
        """Order everything enqueued afterwards on the current stream behind every part of the exchange `tag`."""
        self.trace.append(('finish', tag))
        self._whole.discard(tag)
        if tag in self._deferred:                                   # host_sync mode: produce, exchange, consume -- serially
            import ctypes
            from . import _hip
            self._pending.pop(tag, None)
            comm = self._comm_stream
            for flat, events in self._deferred.pop(tag):
                for ev in events:
                    ev.synchronize()
                for b in self.buckets(flat):
                    _hip.check(_hip.lib().srhip_dp_allreduce_bucket(ctypes.c_void_p(b.data_ptr()), b.numel(),
                                                                    ctypes.c_void_p(comm.cuda_stream)), 'dp_allreduce_bucket')
            comm.synchronize()
            return
        for entry in self._pending.pop(tag, None) or ():
            if hasattr(entry, 'done'):                              # a ticket of the enqueue thread: wait (on the host) until the
                entry.done.wait()                                   # collective has been ISSUED, then order the stream behind it
                if entry.error is not None:
                    raise entry.error
                entry = entry.result
            if isinstance(entry, torch.cuda.Event):
                torch.cuda.current_stream().wait_event(entry)
            else:
                entry.wait()                                        # torch.distributed work handle (CPU tensors)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def __call__(self, flat):
        """Synchronous convenience form: flat <- mean over ranks."""
        self.start('_sync', flat)
        self.finish('_sync')
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
