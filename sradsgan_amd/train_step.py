"""One SRADSGAN training iteration on the HIP path: the body of the reference's batch loop
(SRADSGAN/model/sradsgan.py:829-892) with identical arithmetic.

Structure (MI355X-first, not the reference's call order):
  compute   = zero grads; G forward, losses, backward; D forwards, gradient penalty, backward: ~2600 kernel launches on
              THREE HIP streams -- main (forward passes, data gradients), weight gradients, discriminator passes -- that
              fork and join through events.  Eager by default; with use_graph=True the whole three-stream compute part is
              captured once into ONE hipGraph (the side streams fork from and join the capturing stream, so their kernels
              become parallel branches of the graph) and replayed per iteration: the host enqueues a step in ~15 ms
              instead of 30 - 45 -- but ROCm 7.0's graph executor replays multi-branch graphs slower than the eager
              streams run them (99 vs 63 ms), so eager stays the default.
  exchange  = in-place all-reduce of the G and D gradient arenas over RCCL (only with world_size > 1;
              sradsgan_amd/dp.py GradSync) on its own HIP stream.  G's arena leaves in parts, in REVERSE LAYER ORDER, while
              the generator's backward is still running (round 6): groups 11..8 + the up-sampler as soon as the backward has
              passed group 8 (a tensor hook on that group's input), groups 7..4, groups 3..0, the head / multi-scale block /
              tail conv at the end; each part waits for an event of the main stream and one of the weight-gradient stream
              recorded at that point.  D's arena follows the D stream's backward.  The two Adam launches wait on the parts'
              events.  (In rounds 2-5 the G arena left in one piece after the backward; with the discriminator's passes running
              BESIDE that backward -- the default since round 3 -- nothing was left to hide it under.  Replayed from a graph
              both arenas go out after the replay: the collectives stay outside the capture.)
  update    = one fused Adam kernel per network over its flat arena (srhip_adam_step), the D one
              also applying the weight clip (:891-892).

Differences from the reference that do not change results (DESIGN.md "restructured, same numbers"):
  * Adam(G) runs after the D-step compute instead of before it: the D step reads only
    gen_hr.detach() and D's weights, never G's, so the order is unobservable;
  * during the G step the discriminator / VGG parameters do not require grad, so the weight
    gradients the reference computes and then throws away (:857 -> :865) are never computed;
  * the gradient penalty's double backward runs once with weight (1 + lambda_gp) instead of twice
    (once inside gradient_penalty() :639, once inside loss_D.backward() :886) -- same sum;
  * losses stay on the device; nothing calls .item() inside the step (the reference syncs 4x, :898);
  * D(gen_hr) feeds both losses (-weight_gan * mean in loss_G :847-852, +mean in loss_D :877): its graph is walked ONCE with
    upstream +1, which yields the fake term's share of D's gradients and g = d mean / d gen_hr; the generator's backward takes
    (-weight_gan) * g at gen_hr (_compute_onewalk; differs from two walks by the rounding of one multiplication per element);
  * the discriminator's real and penalty terms are backpropagated on the D stream beside the generator's backward, not after it.
"""
import ctypes
import os

import torch

from . import _hip, ops
from .dp import ParamArena


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


_SIDE_STREAMS = {}


def _side_streams(dev):
    """The two side streams of the step, ONE pair per process and device, created when the first TrainStep is built.  torch
    hands out its pool streams round robin and ROCm binds a stream to a hardware queue when it is created: a second model
    in the same process (chain training, a sweep) used to get two fresh pool streams created late, and its step ran up to
    34 % slower (x3 after x2: 112 instead of 84 ms) -- the same pair for every model keeps the queue assignment of the
    first."""
    key = (dev.type, dev.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _SIDE_STREAMS[key]


def _join_side(stream, side):
    """`stream` waits for the weight-gradient stream -- after every grouped weight gradient that is still waiting for a partner
    has been launched on it: a join that did not flush first would order itself behind "all weight gradients so far" and miss the
    held ones."""
    ops.flush_pending_wgrads()
    stream.wait_stream(side)


class TrainStep:
    _timeline_on = False        # SRHIP_STEP_TIMELINE=1 (set per instance in __init__)
    max_run_ahead = 0

    def __init__(self, generator, discriminator, feature_extractor, lr=2e-4, b1=0.9, b2=0.999,
                 weight_content=1e-2, weight_gan=1e-3, lambda_gp=10.0, clip_value=0.01, use_gp=True,
                 grad_sync=None, use_graph=False, reuse_d_fake=True, overlap_wgrad=True, overlap_d_step=True,
                 wgrad_stream=None, d_stream=None):
        self.G, self.D, self.F = generator, discriminator, feature_extractor
        self.weight_content, self.weight_gan = weight_content, weight_gan
        self.lambda_gp, self.clip_value, self.use_gp = lambda_gp, clip_value, use_gp
        self.lr_G = self.lr_D = lr                       # sradsgan.py:724-725 (halved on plateau, :1021-1027)
        self.b1, self.b2, self.eps = b1, b2, 1e-8
        self.arena_G, self.arena_D = ParamArena(self.G), ParamArena(self.D)
        self.grad_sync = grad_sync                       # dp.GradSync or None
        self.use_graph = use_graph
        self.reuse_d_fake = reuse_d_fake
        self.overlap_wgrad = overlap_wgrad = overlap_wgrad and os.environ.get('SRHIP_OVERLAP_WGRAD', '1') == '1'     # A/B knob
        self.overlap_d_step = overlap_d_step and os.environ.get('SRHIP_OVERLAP_D', '1') == '1'
        if self.overlap_d_step and hasattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
            # D's parameters are used on both streams by design; the engine orders their AccumulateGrad nodes itself
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        dev = self.arena_G.flat_p.device
        shared = _side_streams(dev) if (overlap_wgrad and dev.type == 'cuda') else (None, None)
        self._wgrad_stream = (wgrad_stream or shared[0]) if (overlap_wgrad and dev.type == 'cuda') else None
        d_default = torch.cuda.Stream(device=dev, priority=-1) if os.environ.get('SRHIP_D_PRIO') == '1' else shared[1]   # knob: D stream at high priority
        self._d_stream = (d_stream or d_default) if (overlap_wgrad and dev.type == 'cuda' and os.environ.get('SRHIP_D_STREAM', '1') == '1') else None
        self._bns = [m for m in self.D.modules() if isinstance(m, torch.nn.BatchNorm2d)]
        # weight gradients of one shape launched together (consecutive RABs): 1 = every conv on its own (A/B knob)
        self.wgrad_group = max(1, min(4, int(os.environ.get('SRHIP_WGRAD_GROUP', '2'))))
        self._graph = None
        self._capturing = False
        self._timeline_on = os.environ.get('SRHIP_STEP_TIMELINE') == '1'
        self.timeline = []
        self.max_run_ahead = int(os.environ.get('SRHIP_RUN_AHEAD', '2'))     # steps the host may be ahead of the GPU (0: unbounded)
        self._step_events = []
        self._calls = 0
        self._static = None
        self._d_params = self.arena_D.params
        if dev.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
            ops.plane_pool.release_all()                 # the previous model's padded-plane buffers (other geometries) go back to the allocator
        self._g_parts, self._g_rest = self._plan_g_parts(int(os.environ.get('SRHIP_DP_PARTS', '3')))
        self._parts_armed = False
        self._parts_sent = set()
        self._main = None
        if grad_sync is not None and grad_sync.active and dev.type == 'cuda' and os.environ.get('SRHIP_DP_EAGER_INIT') == '1':
            grad_sync.init_rccl(dev)                     # experiment only: see _compute_onewalk for when the communicator is created
        for k, grp in self._part_groups:
            grp.register_forward_hook(self._make_part_hook(k))
        for p in self.F.parameters():
            p.requires_grad_(False)                      # never in an optimiser (sradsgan.py:724-725)
        ops.mark_static(self.F)

    # ------------------------------------------------------------------------------------------ #
    def _mark(self, name, stream=None):
        """SRHIP_STEP_TIMELINE=1: a timing event on `stream` (default: current), kept in self.timeline for tools/step_timeline.py."""
        if self._timeline_on and not self._capturing:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            self.timeline.append((self._calls, name, ev))

    def _set_d_grad(self, flag):
        for p in self._d_params:
            p.requires_grad_(flag)

    def gradient_penalty(self, real, fake, alpha):
        """sradsgan.py:595-641 ('L2' norm over channels => per-pixel, 'LS' penalty); returns the
        penalty with its double-backward graph attached (the caller backpropagates it)."""
        interp = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
        self._interp = interp                 # the D step's backward does not need d/d(interp) again (backward_scope stop_at)
        d_out = self.D(interp)
        with ops.no_param_grads():
            (grads,) = torch.autograd.grad(d_out, interp, torch.ones_like(d_out), create_graph=True)
        return ops.gp_penalty(grads)

    def _compute(self, imgs_lr, imgs_hr, alpha):
        if (self.reuse_d_fake and self.use_gp and self.overlap_wgrad and self.overlap_d_step and self._wgrad_stream is not None
                and self._d_stream is not None and os.environ.get('SRHIP_D_ONEWALK', '1') == '1'):
            return self._compute_onewalk(imgs_lr, imgs_hr, alpha)
        if self.reuse_d_fake:
            return self._compute_shared(imgs_lr, imgs_hr, alpha)
        G, D, F = self.G, self.D, self.F
        # ------------------ generator (sradsgan.py:829-857) ------------------
        self._set_d_grad(False)
        self.arena_G.zero_grad()
        gen_hr = G(imgs_lr)
        pixel = ops.l1_mean(gen_hr, imgs_hr)
        with torch.no_grad():
            real_feat = F(imgs_hr)
        content = ops.l1_mean(F(gen_hr), real_feat)
        loss_gan = -ops.mean(D(gen_hr))
        loss_G = pixel + self.weight_content * content + self.weight_gan * loss_gan
        loss_G.backward()
        self._exchange_start('G')
        # ---------------- discriminator (sradsgan.py:865-886) ----------------
        self._set_d_grad(True)
        self.arena_D.zero_grad()
        fake = gen_hr.detach()
        terms = [-ops.mean(D(imgs_hr)), ops.mean(D(fake))]
        loss_D = terms[0] + terms[1]
        if self.use_gp:
            gp = self.gradient_penalty(imgs_hr, fake, alpha)
            terms.append((1.0 + self.lambda_gp) * gp)             # :639 + :884-886 => 1 + lambda
            loss_D = loss_D + self.lambda_gp * gp
        else:
            gp = torch.zeros((), device=imgs_hr.device)
        self._backward_terms(terms)
        self._exchange_start('D')
        return dict(loss_G=loss_G.detach(), loss_D=loss_D.detach(), pixel=pixel.detach(),
                    content=content.detach(), loss_gan=loss_gan.detach(), gp=gp.detach(), gen_hr=fake)

    @staticmethod
    def _backward_terms(terms, inputs=None):
        """loss_D.backward() (sradsgan.py:886) as one backward per term of the sum, in program order.  A single call over
        the sum lets the autograd engine interleave the nodes of the real pass, the fake pass and the penalty's double
        backward by their sequence numbers -- and those come from two per-thread counters (forward nodes are numbered by
        the calling thread, the nodes autograd.grad(create_graph=True) creates by the engine's device thread), so which of
        the four contributions to a discriminator weight is added first depended on how much autograd work the process
        had done before (an ulp in D's gradients; tests/test_graph_gpu.py).  Term by term the order is fixed: inside one
        term the nodes of one thread are ordered by creation, and the penalty's two contributions to a weight (through
        the first-order backward's node, then through the forward node) are ordered by data dependence."""
        for t in terms:
            with ops.bn_fold_second_order():        # (the penalty term: a BatchNorm input's two gradients are summed by the BatchNorm backward itself)
                torch.autograd.backward(t, inputs=inputs)

    def _plan_g_parts(self, nparts, any_device=False):
        """The generator's gradient arena as the parts its backward completes, in that order (dp.GradSync, `part=`).

        Arena order = registration order: conv1 | res_groups[0..n) | GAB_UP | MSB | conv3; the backward finishes conv3 and GAB_UP
        first, then the groups n-1 .. 0, then the multi-scale block and the head conv.  Part k (k = 0 .. nparts-1) = the groups
        [b_k, b_(k-1)) with b_k = n (nparts-1-k) / nparts -- part 0 also takes GAB_UP, contiguous behind the last group -- and is
        handed over when the gradient at the INPUT of group b_k exists (a tensor hook registered by a forward hook of that group).
        Everything else -- conv1, MSB, conv3: 0.3 MB -- leaves with the end of the backward.  Returns ([(first, end) per part],
        [(first, end) of the rest]) in elements; self._part_groups = [(k, group module)]."""
        self._part_groups = []
        arena = self.arena_G
        groups = getattr(self.G, 'res_groups', None)
        if nparts < 1 or groups is None or len(groups) < 2 or not (arena.flat_g.is_cuda or any_device):
            return [], [(0, arena.numel)]
        off = {id(p): o for p, o in zip(arena.params, arena.offsets)}
        first_of = []
        for grp in groups:
            ps = list(grp.parameters())
            if not ps or id(ps[0]) not in off:
                return [], [(0, arena.numel)]
            first_of.append(off[id(ps[0])])
        if first_of != sorted(first_of):
            return [], [(0, arena.numel)]
        n = len(groups)
        tail_of_groups = arena.numel                            # first element behind the last group's parameters ...
        up = getattr(self.G, 'GAB_UP', None)
        behind = [o for o in arena.offsets if o > first_of[-1] and not any(o == off[id(p)] for p in groups[-1].parameters())]
        if behind:
            tail_of_groups = min(behind)
        end0 = tail_of_groups                                   # ... part 0 extends over GAB_UP when it sits right there
        if up is not None:
            ups = [off[id(p)] for p in up.parameters() if id(p) in off]
            if ups and min(ups) == tail_of_groups:
                later = [o for o in arena.offsets if o > max(ups)]
                end0 = min(later) if later else arena.numel
        bounds = sorted({n * (nparts - 1 - k) // nparts for k in range(nparts)}, reverse=True)      # group indices b_0 > b_1 > ... >= 0
        parts, hi = [], end0
        for k, b in enumerate(bounds):
            parts.append((first_of[b], hi))
            self._part_groups.append((k, groups[b]))
            hi = first_of[b]
        rest = [(0, hi)] if hi > 0 else []
        if end0 < arena.numel:
            rest.append((end0, arena.numel))
        return parts, rest

    def _make_part_hook(self, k):
        def on_forward(module, inputs, output):
            x = inputs[0] if inputs else None
            if self._parts_armed and torch.is_tensor(x) and x.requires_grad:
                x.register_hook(lambda grad: self._bucket_ready(k))    # fires when the whole group's backward has been enqueued
        return on_forward

    def _bucket_ready(self, k):
        """Part k of the generator's arena is final once everything enqueued so far has run: hand it to the exchange behind an event
        of the main stream and one of the weight-gradient stream.  Runs on autograd's device thread, in the middle of the backward."""
        gs = self.grad_sync
        if not self._parts_armed or k in self._parts_sent or gs is None or not gs.active:
            return
        self._parts_sent.add(k)
        ops.flush_pending_wgrads()              # (group boundaries are pair boundaries of the RAB weight gradients: nothing of theirs is held)
        flat = self.arena_G.flat_g
        evs = []
        if flat.is_cuda:
            evs.append(self._main.record_event())
            if self.overlap_wgrad and self._wgrad_stream is not None:
                evs.append(self._wgrad_stream.record_event())
        lo, hi = self._g_parts[k]
        gs.start('G', flat[lo:hi], events=evs, part=k)
        if flat.is_cuda:
            self._mark('G part %d handed to the exchange (main)' % k, self._main)

    def _exchange_start(self, which, producers=None):
        """Hands a finished gradient arena -- for the generator: what its parts have not taken yet -- to the exchange
        (dp.GradSync.start): called right after the backward that completes it has been ENQUEUED; the collective waits on events
        of the streams that produce the arena."""
        ops.flush_pending_wgrads()              # grouped weight gradients still waiting for a partner go out now
        gs = self.grad_sync
        if gs is None or not gs.active or self._capturing:
            return
        arena = self.arena_G if which == 'G' else self.arena_D
        streams = [torch.cuda.current_stream()] if arena.flat_g.is_cuda else []
        if arena.flat_g.is_cuda and self.overlap_wgrad and self._wgrad_stream is not None:
            streams.append(self._wgrad_stream)                    # the weight-gradient kernels run there
        if producers is not None and arena.flat_g.is_cuda:
            streams = list(producers)                             # the caller knows which streams complete this arena
        if which == 'G' and getattr(self, '_parts_armed', False):
            self._parts_armed = False
            evs = [s.record_event() for s in streams]
            todo = [(k, r) for k, r in enumerate(self._g_parts) if k not in self._parts_sent]       # (a hook that never fired: its part leaves now)
            todo += [(len(self._g_parts) + i, r) for i, r in enumerate(self._g_rest)]
            for k, (lo, hi) in todo:
                gs.start('G', arena.flat_g[lo:hi], events=evs, part=k)
            return
        gs.start(which, arena.flat_g, after=streams)

    def _compute_onewalk(self, imgs_lr, imgs_hr, alpha):
        """The three-stream step with the discriminator's graph on the generated batch walked ONCE (default).

        D(gen_hr) appears in both losses: loss_G carries -weight_gan * mean(D(gen_hr)) (sradsgan.py:847-852) and loss_D
        +mean(D(gen_hr.detach())) (:877).  Both need the same data-gradient chain through D (and through its train-mode
        BatchNorms) with upstream gradients that differ only by the constant factor -weight_gan, so the chain is walked once
        with upstream +1: it yields d mean / d theta_D -- the fake term's contribution to D's gradients -- and
        g = d mean / d gen_hr, which enters the generator's backward as (-weight_gan) * g at gen_hr.  (Walking it twice, as
        _compute_shared does, costs a second data-gradient + BatchNorm-backward pass through D at 216 x 216 on the main
        stream's critical path; the results differ by the rounding of one multiplication by weight_gan per element.)

        Host order = dependency order of the three chains:
          main : G forward, VGG(fake), D(gen), losses | walk of D(gen) [its weight gradients -> wgrad stream] | G backward
          D    : (after the forward) D(real), D(interp) + first-order penalty backward | (after the walk) backward of the
                 real and penalty terms, weight gradients in line -- all of it beside the generator's backward
          wgrad: VGG(real) | fake-term weight gradients of D | the generator's weight gradients
        D's gradient arena receives: fake term (main / wgrad stream), then real + penalty (D stream, after events) --
        a fixed order, so the step stays deterministic."""
        G, D, F = self.G, self.D, self.F
        g_params, d_params = self.arena_G.params, self.arena_D.params
        side, dside = self._wgrad_stream, self._d_stream
        main = self._main = torch.cuda.current_stream()
        gs = self.grad_sync
        self._parts_sent = set()
        # The parts are armed from the SECOND iteration on: the first one hands both arenas over whole, from this thread, after the
        # backward -- that is where the communicator, its rendezvous and the comm stream come into being (GradSync.init_rccl inside
        # start()), AFTER the step's three compute streams have been used and own their hardware queues.  Created before them (round 6
        # tried it in TrainStep.__init__, to keep the rendezvous off autograd's device thread) the high-priority comm stream takes a
        # queue the compute streams then share: 69.6 instead of 48.2 ms per step (profiles/r06_step_ab.txt).
        self._parts_armed = (bool(self._g_parts) and gs is not None and gs.active and not self._capturing and not gs.host_sync
                             and (gs._rccl_ready or not self.arena_G.flat_g.is_cuda))
        self._set_d_grad(True)
        self.arena_G.zero_grad()
        self.arena_D.zero_grad()
        side.wait_stream(main)
        with torch.cuda.stream(side), torch.no_grad():
            real_feat = F(imgs_hr)                               # depends on nothing the generator produces
        imgs_hr.record_stream(side)
        self._mark('start')
        gen_hr = G(imgs_lr)
        self._mark('G fwd done')
        pixel = ops.l1_mean(gen_hr, imgs_hr)
        fake_feat = F(gen_hr)
        gen_in = gen_hr.detach().requires_grad_(True)            # the discriminator's graph hangs off its own leaf
        stash = []
        for bn in self._bns:
            bn._stat_stash = stash
        d_gen = D(gen_in)                                        # running-stat update #1
        for bn in self._bns:
            bn._stat_stash = None
        _join_side(main, side)
        real_feat.record_stream(main)
        content = ops.l1_mean(fake_feat, real_feat)
        fake_term = ops.mean(d_gen)
        loss_gan = -fake_term.detach()
        loss_G_own = pixel + self.weight_content * content       # the part of loss_G whose graph is G + VGG
        loss_G = loss_G_own.detach() + self.weight_gan * loss_gan
        self._mark('fwd done (VGG, D(gen), losses)')
        # ---- D stream: real pass, interpolate pass, first-order backward of the penalty ----
        dside.wait_stream(main)
        self._mark('D passes begin (D stream)', dside)
        with torch.cuda.stream(dside):
            real_term = -ops.mean(D(imgs_hr))                    # update #2 (real)
            ops.replay_bn_update(stash)                          # update #3 (the fake pass that is not recomputed)
            fake = gen_hr.detach()
            gp = self.gradient_penalty(imgs_hr, fake, alpha)     # update #4
            gp_term = (1.0 + self.lambda_gp) * gp                # :639 + :884-886 => 1 + lambda
            loss_D = real_term.detach() + fake_term.detach() + self.lambda_gp * gp.detach()
        self._mark('D passes + GP first order done (D stream)', dside)
        for t in (gen_hr, gen_in, d_gen, alpha, fake_term):
            t.record_stream(dside)
        # ---- main: the one walk of D(gen) ----
        torch.autograd.backward(fake_term, inputs=list(d_params) + [gen_in])
        ops.flush_pending_wgrads()                               # the fake term's weight gradients are all on their stream
        walked_main, walked_side = main.record_event(), side.record_event()
        g_adv = gen_in.grad.mul_(-self.weight_gan)               # d(weight_gan * loss_gan) / d gen_hr
        self._mark('D(gen) walked (main)')
        # ---- D stream: backward of the real and penalty terms, after the fake term's contributions to D's arena ----
        with torch.cuda.stream(dside):
            dside.wait_event(walked_main)
            dside.wait_event(walked_side)
            with ops.direct_param_grads(None), ops.backward_scope(stop_at=(self._interp,)):
                self._backward_terms([real_term, gp_term], d_params)
        self._mark('D bwd of the real + penalty terms done (D stream)', dside)
        # ---- main: the generator's backward ----
        torch.autograd.backward([loss_G_own, gen_hr], grad_tensors=[None, g_adv], inputs=g_params)
        self._exchange_start('G')
        self._mark('G bwd done (main)')
        self._mark('wgrads done (wgrad stream)', side)
        # D's arena is complete with the D stream's backward (which waited for the fake term's contributions: walked_main / walked_side);
        # its collective is ISSUED here, behind the generator's parts (one communicator: same order on every rank), but waits only for
        # the D stream -- not for the end of the generator's backward on the main stream
        self._exchange_start('D', producers=[dside])
        main.wait_stream(dside)
        _join_side(main, side)
        for t in (loss_D, gp):
            t.record_stream(main)
        out = dict(loss_G=loss_G, loss_D=loss_D, pixel=pixel.detach(), content=content.detach(), loss_gan=loss_gan,
                   gp=gp.detach(), gen_hr=fake)
        if os.environ.get('SRHIP_STEP_DEBUG') == '1':
            out['d_gen'] = d_gen.detach()
        return out

    def _compute_shared(self, imgs_lr, imgs_hr, alpha):
        """Same arithmetic, one discriminator forward fewer: D(gen_hr) of the G step (:847) and
        D(gen_hr.detach()) of the D step (:877) see the same input and the same weights, so one graph
        serves both -- the G-step backward walks it for d/d(gen_hr) only, the D-step backward for
        d/d(theta_D) only.  The running statistics of BatchNorm receive the update of the skipped forward
        at the position the reference applies it (after D(real))."""
        G, D, F = self.G, self.D, self.F
        g_params, d_params = self.arena_G.params, self.arena_D.params
        self._set_d_grad(True)
        self.arena_G.zero_grad()
        self.arena_D.zero_grad()
        side = self._wgrad_stream if self.overlap_wgrad else None
        if side is not None:
            # VGG features of the real batch depend on nothing the generator produces: compute them on the
            # side stream while the generator's forward runs (fills the partially occupied kernel tails)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                real_feat = F(imgs_hr)
            imgs_hr.record_stream(side)
        self._mark('start')
        gen_hr = G(imgs_lr)
        self._mark('G fwd done')
        pixel = ops.l1_mean(gen_hr, imgs_hr)
        early = os.environ.get('SRHIP_LATE_JOIN', '1') != '1' and side is not None     # A/B knob (old order)
        if early:
            _join_side(torch.cuda.current_stream(), side)
        fake_feat = F(gen_hr)
        stash = []
        for bn in self._bns:
            bn._stat_stash = stash
        d_gen = D(gen_hr)                                         # running-stat update #1
        for bn in self._bns:
            bn._stat_stash = None
        # the real batch's features are needed only now: joining the side stream here instead of right after the generator's
        # forward removes a 6 ms stall of the main stream (the side stream's VGG pass shares the GPU with G's forward and
        # finishes ~6 ms after it; profiles/r02_step_eager_kernel_stats.txt)
        if side is None:
            with torch.no_grad():
                real_feat = F(imgs_hr)
        else:
            _join_side(torch.cuda.current_stream(), side)
            real_feat.record_stream(torch.cuda.current_stream())   # allocated in the side stream's pool, read here
        content = ops.l1_mean(fake_feat, real_feat)
        loss_gan = -ops.mean(d_gen)
        loss_G = pixel + self.weight_content * content + self.weight_gan * loss_gan

        def d_forward():
            # ---------------- discriminator forward passes (sradsgan.py:865-884) ----------------
            terms = [-ops.mean(D(imgs_hr)), ops.mean(d_gen)]      # update #2 (real)
            loss_D = terms[0] + terms[1]
            ops.replay_bn_update(stash)                           # update #3 (the fake pass that is not recomputed)
            fake = gen_hr.detach()
            if self.use_gp:
                gp = self.gradient_penalty(imgs_hr, fake, alpha)  # update #4
                terms.append((1.0 + self.lambda_gp) * gp)         # :639 + :884-886 => 1 + lambda
                loss_D = loss_D + self.lambda_gp * gp
            else:
                gp = torch.zeros((), device=imgs_hr.device)
            return loss_D, gp, terms, fake

        if side is not None and self.overlap_d_step:
            # The discriminator's real / interpolate passes (incl. the first-order backward of the penalty) depend
            # on gen_hr and D's weights only, not on the generator's backward: enqueue them on the side stream
            # BEFORE the generator's backward so the two chains run concurrently on the GPU.  autograd replays every
            # node on the stream of its forward, so their double backward stays on the side stream too.
            main = torch.cuda.current_stream()
            dside = self._d_stream if self._d_stream is not None else side   # third stream: the D passes beside G's dgrads (main) and the wgrads (side)
            self._mark('fwd done (VGG, D(gen), losses)')
            late = os.environ.get('SRHIP_D_LATE', '0') == '1'               # A/B knob, off by default
            if late:
                # Alternative HOST order: the generator's backward first, the D passes second (the D stream waits only for a
                # forward-done event, so its kernels still run beside the generator's backward).  It removes a 6.6 ms dry
                # spell of the main stream when the host is slow (under rocprofv3) and times the same at full host speed --
                # but with the gradient exchange active it is 19 % SLOWER (74.0 vs 62.8 ms, single-rank communicator):
                # the RCCL enqueue in _exchange_start('G') holds the host until the generator's backward has run, and in
                # this order the D passes are not enqueued yet at that point.  Hence the D passes go first.
                fwd_done = torch.cuda.Event()
                fwd_done.record(main)
                with ops.backward_scope(skip_params=d_params):    # no discriminator wgrads in the G step (:857 -> :865)
                    torch.autograd.backward(loss_G, inputs=g_params, retain_graph=True)
                self._exchange_start('G')                         # G's gradients travel under the whole D step
                self._mark('G bwd done (main)')
                self._mark('G wgrads done (wgrad stream)', side)
                dside.wait_event(fwd_done)                        # gen_hr, d_gen and running-stat update #1 are in
            else:
                dside.wait_stream(main)
            self._mark('D passes begin (D stream)', dside)
            early = self.use_gp and os.environ.get('SRHIP_D_EARLY', '1') == '1' and not late
            with torch.cuda.stream(dside):
                loss_D, gp, terms, fake = d_forward()
                stop = (gen_hr,) + ((self._interp,) if self.use_gp else ())      # (gradient_penalty sets _interp)
                self._mark('D passes + GP first order done (D stream)', dside)
                if early:
                    # The real pass and the penalty live entirely on the D stream and do not depend on the generator's
                    # backward: walk them HERE, with the D stream as the calling stream.  (Called from the main stream after
                    # the generator's backward, the root gradient of each term is produced at that point of the MAIN
                    # stream's queue, so the D stream sat idle from ~27 ms to ~48 ms of a 60 ms step and its serial
                    # data-gradient / BatchNorm chain became a 10 - 15 ms tail, tools/step_timeline.py.)  Their weight-gradient
                    # kernels run in line on the D stream -- the weight-gradient stream is one queue in host order, and
                    # these would block the generator's weight gradients behind kernels that wait for the D stream.
                    with ops.direct_param_grads(None), ops.backward_scope(stop_at=stop):
                        self._backward_terms([terms[0], terms[2]], d_params)
                    self._mark('D bwd of the real + penalty terms done (D stream)', dside)
            for t in (gen_hr, d_gen, alpha):
                t.record_stream(dside)
            if not late:
                with ops.backward_scope(skip_params=d_params):
                    torch.autograd.backward(loss_G, inputs=g_params, retain_graph=True)
                self._exchange_start('G')
                self._mark('G bwd done (main)')
            if early:
                # D's arena: the fake term's contributions (weight gradients on the weight-gradient stream, BatchNorm's in line
                # on the main stream) after the D stream's
                side.wait_stream(dside)
                main.wait_stream(dside)
                with ops.backward_scope(stop_at=stop):
                    self._backward_terms([terms[1]], d_params)
            else:
                with ops.backward_scope(stop_at=stop):           # d/d(gen_hr) is not needed any more
                    for t in terms:                              # real (D stream), fake (main), penalty (D stream): one
                        self._backward_terms([t], d_params)      # after the other -- they add into the same arena slots
                        main.wait_stream(dside)
                        dside.wait_stream(main)
            self._mark('D bwd done (main)')
            self._mark('D bwd done (D stream)', dside)
            self._mark('wgrads done (wgrad stream)', side)
            main.wait_stream(dside)
            _join_side(main, side)
            for t in (loss_D, gp):
                t.record_stream(main)
        else:
            with ops.backward_scope(skip_params=d_params):
                torch.autograd.backward(loss_G, inputs=g_params, retain_graph=True)
            self._exchange_start('G')
            loss_D, gp, terms, fake = d_forward()
            with ops.backward_scope(stop_at=(gen_hr,) + ((self._interp,) if self.use_gp else ())):
                self._backward_terms(terms, d_params)
        self._exchange_start('D')
        out = dict(loss_G=loss_G.detach(), loss_D=loss_D.detach(), pixel=pixel.detach(),
                   content=content.detach(), loss_gan=loss_gan.detach(), gp=gp.detach(), gen_hr=fake)
        if os.environ.get('SRHIP_STEP_DEBUG') == '1':
            out['d_gen'] = d_gen.detach()
        return out

    def _adam(self, arena, lr, clip, scale):
        _hip.check(_hip.lib().srhip_adam_step(_ptr(arena.flat_p), _ptr(arena.flat_g), _ptr(arena.exp_avg),
                                              _ptr(arena.exp_avg_sq), _ptr(arena.step_state), arena.numel,
                                              float(lr), float(self.b1), float(self.b2), float(self.eps),
                                              float(scale), float(clip),
                                              ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'adam_step')

    def _update(self):
        """exchange + update: sradsgan.py:858 (optimizer_G.step), :887 (optimizer_D.step), :891-892 (clip).
        The all-reduces were launched from _compute (or are launched here when the compute part is a replayed
        hipGraph); each Adam kernel waits for its arena's exchange through a stream event -- no host synchronise."""
        gs = self.grad_sync
        scale = 1.0
        if gs is not None and gs.active:
            if self.use_graph and self._graph is not None:
                gs.start('G', self.arena_G.flat_g)
                gs.start('D', self.arena_D.flat_g)
            scale = gs.grad_scale
            gs.finish('G')
        self._adam(self.arena_G, self.lr_G, 0.0, scale)
        if gs is not None and gs.active:
            gs.finish('D')
        self._adam(self.arena_D, self.lr_D, self.clip_value, scale)
        ops.bump_weight_epoch()
        self._mark('update done')

    # ------------------------------------------------------------------------------------------ #
    def _run_compute(self, imgs_lr, imgs_hr, alpha):
        """The compute part on the current stream + the side streams, forked from and joined back into the current stream
        (so the same code runs eagerly and under stream capture)."""
        side = self._wgrad_stream if self.overlap_wgrad else None
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())     # arena zeroing / previous Adam before any wgrad
        with ops.direct_param_grads(side, group=self.wgrad_group):   # wgrad kernels accumulate straight into the gradient arenas
            out = self._compute(imgs_lr, imgs_hr, alpha)
        if side is not None:
            _join_side(torch.cuda.current_stream(), side)     # all weight gradients landed before the update
        return out

    def _capture(self, imgs_lr, imgs_hr, alpha):
        """Records the compute part -- all three streams -- into one hipGraph; nothing executes.  The side streams enter the
        capture at their first wait on an event of the capturing stream and leave it at the joins at the end of
        _compute_shared / _run_compute, so the capture ends with every forked stream joined (a HIP requirement).  Tensors
        that cross streams carry record_stream marks: under capture the caching allocator keeps such blocks out of reuse
        until the capture ends, which is exactly the lifetime the parallel branches need (costs pool memory, not time)."""
        self._static = dict(lr=imgs_lr.clone(), hr=imgs_hr.clone(), alpha=alpha.clone())
        self._graph = torch.cuda.CUDAGraph()
        dump = os.environ.get('SRHIP_GRAPH_DUMP')
        if dump:
            self._graph.enable_debug_mode()
        self._capturing = ops._state.capturing = True
        ops.plane_pool.begin_capture()
        try:
            # same arithmetic and the same per-stream program order as the eager path
            with torch.cuda.graph(self._graph):
                self._out = self._run_compute(self._static['lr'], self._static['hr'], self._static['alpha'])
        finally:
            self._capturing = ops._state.capturing = False
        if dump:
            self._graph.debug_dump(dump)

    def _bound_run_ahead(self, out):
        # Bound the host's run-ahead: the host enqueues a step in 35 - 50 ms, the GPU needs 61, so unchecked the host
        # drifts steps ahead, and every block another stream touched (record_stream) stays unavailable until that
        # stream's event has passed -- the allocator's reserved pool grew to 79 GB for 14 GB of live data
        # (tools/mem_growth.py).  Waiting for the end of the step before the previous one keeps the GPU fed (one full
        # step is always queued) and caps the pool.  (Graph replays allocate nothing; the bound then only keeps the
        # launch queue short.)
        if self.max_run_ahead > 0 and out['loss_G'].is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            self._step_events.append(ev)
            if len(self._step_events) > self.max_run_ahead:
                self._step_events.pop(0).synchronize()

    def __call__(self, imgs_lr, imgs_hr, alpha):
        self._calls += 1
        if not self.use_graph or self._calls == 1:
            # (graph mode: the first iteration runs eagerly -- library, allocator and packed-weight warm-up must happen
            # outside a capture; it is a real iteration, not a discarded one)
            out = self._run_compute(imgs_lr, imgs_hr, alpha)
            self._update()
            self._bound_run_ahead(out)
            return out
        if self._graph is None:
            self._capture(imgs_lr, imgs_hr, alpha)           # records only; nothing executes
        else:
            for k, t in (('lr', imgs_lr), ('hr', imgs_hr), ('alpha', alpha)):
                if t.data_ptr() != self._static[k].data_ptr():
                    self._static[k].copy_(t)
        self._graph.replay()
        self._update()
        self._bound_run_ahead(self._out)
        return self._out
