"""hipGraph replay of the training step's compute part (TrainStep(use_graph=True)).

Round 1 saw NaNs from the gradient-penalty path with exactly one pattern -- replays, ONE device-wide synchronisation, replays
-- while "never synchronise" and "synchronise every step" agreed with eager launches.  The ATen reductions that path used
(norm / pow / mean; bmm + softmax in the generator) are HIP kernels now and the symptom is gone; this test pins it:

  * 52 iterations replayed with device-wide synchronisations sprinkled in (after the warm-up step, every 7th step, twice
    back to back) are BIT-IDENTICAL -- all six scalars at every iteration, all weights and BatchNorm buffers at the end --
    to 52 iterations replayed with no synchronisation at all: the replay does not depend on the host's sync pattern;
  * both are bit-identical -- scalars of all 52 iterations, final weights and buffers -- to eager launches of the same
    program order (overlap_wgrad=False, overlap_d_step=False; the default eager path runs the discriminator passes on a
    second stream, which changes the order in which their weight-gradient contributions meet and with it the last bit of
    D's gradients, so it is a different -- equally deterministic -- trajectory)."""
import pytest
import torch

from oracle import sradsgan_ref as O
from tests.parity_util import build_pair

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
NAMES = ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')


def _run(use_graph, iters, sync_at, three_streams=False, groups=2, blocks=1):
    from sradsgan_amd.train_step import TrainStep
    (hg, hd, hf), _ = build_pair(groups, blocks, 4, DEV)
    step = TrainStep(hg, hd, hf, use_graph=use_graph, overlap_wgrad=three_streams, overlap_d_step=three_streams)
    batches = [(O.det_fill('graph.lr.%d' % (i % 3), (4, 3, 24, 24), 0.5, 0.5).to(DEV),
                O.det_fill('graph.hr.%d' % (i % 3), (4, 3, 96, 96), 0.5, 0.5).to(DEV),
                O.det_fill('graph.alpha.%d' % (i % 5), (4, 1, 1, 1), 0.5, 0.5).to(DEV)) for i in range(15)]
    scal = []
    for it in range(iters):
        out = step(*batches[it % 15])
        scal.append(torch.stack([out[k].double() for k in NAMES]).clone())
        if it in sync_at:
            torch.cuda.synchronize()
            if it % 14 == 0:
                torch.cuda.synchronize()
    torch.cuda.synchronize()
    weights = [p.detach().clone() for p in list(hg.parameters()) + list(hd.parameters())] + [b.detach().clone() for b in hd.buffers()]
    return torch.stack(scal).cpu(), weights


def test_graph_replay_is_independent_of_host_syncs_and_tracks_eager():
    # No warm-up run and no fresh process needed: TrainStep backpropagates loss_D term by term, which pins the order in
    # which the four contributions to a discriminator weight are added (train_step._backward_terms; before that the
    # order followed autograd's two per-thread node counters and with them the process's history -- an ulp in D's
    # gradients between graph and eager whenever other autograd work had run first, the round-2 probe of the same name, git history).
    iters = 52
    sync_at = {1, 2} | set(range(7, iters, 7))
    synced_s, synced_w = _run(True, iters, sync_at)
    free_s, free_w = _run(True, iters, set())
    assert torch.isfinite(synced_s).all() and torch.isfinite(free_s).all()
    bad = (synced_s != free_s).any(dim=1).nonzero().flatten().tolist()
    assert not bad, ('replay depends on the sync pattern; first differing iterations', bad[:3], (synced_s[bad[0]] - free_s[bad[0]]).tolist())
    for a, b in zip(synced_w, free_w):
        assert torch.equal(a, b)
    eager_s, eager_w = _run(False, iters, sync_at)
    rel = ((synced_s - eager_s).abs() / eager_s.abs().clamp_min(1e-3)).max(dim=1)[0]
    print('graph vs eager, max relative scalar difference per iteration:', ['%.1e' % v for v in rel.tolist()[:6]], '... max %.1e' % float(rel.max()))
    assert float(rel.max()) == 0.0                       # and bit-identical to eager launches of the same program order
    for a, b in zip(synced_w, eager_w):
        assert torch.equal(a, b)


def test_three_stream_capture_replays_bit_identically_to_the_three_stream_eager_step():
    """The default step runs on three HIP streams (main, weight gradients, discriminator passes).  use_graph=True captures
    all three into ONE hipGraph: the side streams fork from the capturing stream and join it again, their kernels become
    parallel branches.  Per stream the program order is the eager one, so every accumulation meets its operands in the same
    order: 56 iterations (1 eager warm-up + 55 replays) with device-wide synchronisations sprinkled in must be
    BIT-IDENTICAL -- six scalars per iteration, final weights, BatchNorm buffers -- to the eager three-stream step, and
    to the same replay without any synchronisation."""
    iters = 56
    sync_at = {1, 2} | set(range(7, iters, 7))
    graph_s, graph_w = _run(True, iters, sync_at, three_streams=True)
    free_s, free_w = _run(True, iters, set(), three_streams=True)
    eager_s, eager_w = _run(False, iters, sync_at, three_streams=True)
    assert torch.isfinite(graph_s).all()
    for name, other_s, other_w in (('unsynchronised replay', free_s, free_w), ('eager three-stream step', eager_s, eager_w)):
        bad = (graph_s != other_s).any(dim=1).nonzero().flatten().tolist()
        assert not bad, ('three-stream graph differs from the ' + name, bad[:3], (graph_s[bad[0]] - other_s[bad[0]]).tolist())
        for a, b in zip(graph_w, other_w):
            assert torch.equal(a, b), name


def test_three_stream_capture_with_recycled_plane_buffers_is_bit_identical_to_eager():
    """with 3 RABs per group the backward hands a padded-plane buffer back to the pool (released by the weight-gradient
    stream's launch) and takes it again for the next RAB on the main stream BEFORE the backward ends.  Eager launches order the reuse
    by querying the release events; under capture the pool must order it with event edges (ops._PlanePool.get waits for the
    releasing streams' events).  2 groups x 3 RABs, three streams: 1 eager warm-up + 11 replays, bit-identical to the eager step."""
    from sradsgan_amd import ops
    iters = 12
    created0 = ops.plane_pool.created
    graph_s, graph_w = _run(True, iters, {1, 5}, three_streams=True, groups=2, blocks=3)
    eager_s, eager_w = _run(False, iters, {1, 5}, three_streams=True, groups=2, blocks=3)
    assert ops.plane_pool.created > created0                # the planes path ran
    assert torch.isfinite(graph_s).all()
    bad = (graph_s != eager_s).any(dim=1).nonzero().flatten().tolist()
    assert not bad, ('graph replay with recycled plane buffers differs from eager', bad[:3], (graph_s[bad[0]] - eager_s[bad[0]]).tolist())
    for a, b in zip(graph_w, eager_w):
        assert torch.equal(a, b)
