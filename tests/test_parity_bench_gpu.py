"""End-to-end parity at the configurations the numbers are quoted on:

(i)  the full 12 x 3 generator, 54 -> 216, at a batch where `bench.py`'s kernels are the ones selected (B = 12:
     288 patch tiles per conv => conv_patch_kernel / wgrad_rowtap_kernel, not the small-grid LDS-DMA kernels the B = 2
     full-size test reaches): two training iterations against the CPU oracle AND against the same step forced onto the
     already pinned kernel family (srhip_debug_set(0, 23): no patch kernel => LDS-DMA fprop/dgrad, (1, 7): generic
     split-K wgrad; key 0 value -1 would also push the attention MLP's 24-row convs from the exact-fp32 small-grid
     kernel onto split-bf16 and so change arithmetic, not just kernels);
(ii) generator forward + backward at the real LR tile of every other scale of BASELINE configs[4]
     (x2: 108, x3: 72, x8: 27, x9: 24): SGAM at N = 11664, the r = 3 two-stage up-sampler, odd 27 x 27 maps;
(iii) post-step weights against the vectors recorded from the reference (train_full.npz G_after__* / D_after__*) and
     against the oracle's, by quantile (Adam maps a gradient whose sign is roundoff to a +-lr step on any two
     platforms, so a max-norm bound cannot be tight; a quantile can).
"""
import numpy as np
import pytest
import torch

from oracle import sradsgan_ref as O
from tests.parity_util import build_pair, grad_score, rel_err, weight_quantiles

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
NAMES = ['loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp']


def _batch(tag, it, batch, lr_side, scale):
    return (O.det_fill('%s.lr.%d' % (tag, it), (batch, 3, lr_side, lr_side), 0.5, 0.5),
            O.det_fill('%s.hr.%d' % (tag, it), (batch, 3, lr_side * scale, lr_side * scale), 0.5, 0.5),
            O.det_fill('%s.alpha.%d' % (tag, it), (batch, 1, 1, 1), 0.5, 0.5))


def _run_hip(batch, iters, tag, debug=()):
    """`iters` iterations from the deterministic initial weights; returns per-iteration scalars, the first iteration's
    gradients and the networks."""
    from sradsgan_amd import _hip
    from sradsgan_amd.train_step import TrainStep
    lib = _hip.lib()
    for key, value in debug:
        lib.srhip_debug_set(key, value)
    try:
        (hg, hd, hf), _ = build_pair(12, 3, 4, DEV)
        step = TrainStep(hg, hd, hf)
        scal, grads = [], None
        for it in range(iters):
            lr_img, hr_img, alpha = _batch(tag, it, batch, 54, 4)
            out = step(lr_img.to(DEV), hr_img.to(DEV), alpha.to(DEV))
            scal.append(np.array([float(out[k]) for k in NAMES]))
            if it == 0:
                grads = {('G.' + k): p.grad.detach().clone() for k, p in hg.named_parameters()}
                grads.update({('D.' + k): p.grad.detach().clone() for k, p in hd.named_parameters()})
        torch.cuda.synchronize()
        return scal, grads, (hg, hd)
    finally:
        for key, _ in debug:
            lib.srhip_debug_set(key, 1 if key in (8, 10, 11) else 0)      # (keys 8, 10 and 11 default to 1)


def test_bench_configuration_step_b12_against_oracle_and_pinned_kernels():
    B, tag = 12, 'bench_b12'
    scal, grads, (hg, hd) = _run_hip(B, 2, tag)
    # the same job (a) with the 64-wide one-tile patch kernel in its bit-identical 2 x 2 form instead of the K-split form
    # (srhip_debug_set(10, 0)) and (b) on the kernel family the small tests pin: (a) and (b) have bit-identical fprop / dgrad by
    # construction (the generic split-K wgrad sums in a different order), so they must agree tightly; the K-split kernel groups
    # its sums differently (1e-6 relative per activation), which near-ties of the 48 arg-max poolings turn into a few
    # re-routed gradients: a looser bar between the default and (a), the oracle bars below hold for the default
    scal_c, grads_c, _ = _run_hip(B, 2, tag, debug=((10, 0),))
    scal_p, grads_p, _ = _run_hip(B, 2, tag, debug=((0, 23), (1, 7)))

    def compare(sa, ga, sb, gb, what, sbar, gbar):
        for it in range(2):
            d = float(np.abs(sa[it] - sb[it]).max())
            print('b12 it %d: scalars %s max diff %.3e' % (it, what, d))
            assert d <= sbar * max(1.0, float(np.abs(sb[it]).max())), (what, it, sa[it], sb[it])
        worst = 0.0
        for net in ('G.', 'D.'):
            keys = [k for k in ga if k.startswith(net)]
            net_scale = max(float(gb[k].abs().max()) for k in keys)
            for k in keys:
                if k.endswith(('key_conv.bias',)) or any(k == 'D.model.%d.bias' % i for i in (2, 5, 8, 11, 14, 19, 22)):
                    continue                                            # identically zero gradients: roundoff only
                d = float((ga[k] - gb[k]).abs().max())
                worst = max(worst, d / max(float(gb[k].abs().max()), 1e-2 * net_scale))
        print('b12: first-iteration gradients %s, worst score %.3e' % (what, worst))
        assert worst <= gbar, (what, worst)

    compare(scal_c, grads_c, scal_p, grads_p, '2x2 patch form vs pinned kernels', 1e-5, 1e-4)
    compare(scal, grads, scal_c, grads_c, 'default (K-split) vs 2x2 patch form', 1e-4, 3e-2)
    # and against the CPU oracle (identical weights and inputs)
    _, (og, od, of) = build_pair(12, 3, 4, torch.device('cpu'))
    oG = torch.optim.Adam(og.parameters(), lr=2e-4, betas=(0.9, 0.999))
    oD = torch.optim.Adam(od.parameters(), lr=2e-4, betas=(0.9, 0.999))
    for it in range(2):
        lr_img, hr_img, alpha = _batch(tag, it, B, 54, 4)
        want = O.train_step(og, od, of, oG, oD, lr_img, hr_img, alpha)
        wv = np.array([want[k] for k in NAMES])
        d = float(np.abs(scal[it] - wv).max())
        print('b12 it %d: scalars vs oracle max diff %.3e  (HIP %s)' % (it, d, scal[it]))
        assert d < 1e-3, (it, scal[it], wv)
        if it == 0:
            import types

            class _G:                                               # grad_score wants named_parameters() with .grad
                def __init__(self, prefix):
                    self.items = [(k[len(prefix):], types.SimpleNamespace(grad=v)) for k, v in grads.items() if k.startswith(prefix)]

                def named_parameters(self):
                    return self.items
            sg, kg = grad_score((_G('G.'),), (og,), verbose=True)
            sd, kd = grad_score((_G('D.'),), (od,), verbose=True)
            print('b12: first-iteration gradients vs fp32 oracle: G %.3e (%s)  D %.3e (%s)' % (sg, kg, sd, kd))
            assert sg < 5e-3 and sd < 5e-2, (sg, kg, sd, kd)
    q = weight_quantiles((hg, hd), (og, od))
    print('b12: post-step weights vs oracle: %s' % q)
    assert q["frac_within"] >= 0.997 and q['max'] <= 2 * 2e-4 * 2 * 1.01 + 1e-7, q


@pytest.mark.parametrize('scale,lr_side', [(2, 108), (3, 72), (8, 27), (9, 24)])
def test_generator_forward_backward_at_real_tile_sizes(scale, lr_side):
    """BASELINE configs[4] tile sizes (HR 216 x 216): the full generator at B = 1 against the CPU oracle."""
    from sradsgan_amd import model as M
    og = O.GeneratorResNet(O.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=scale)
    O.det_init_(og, prefix='G.')
    hg = M.GeneratorResNet(M.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=scale)
    hg.load_state_dict(og.state_dict(), strict=True)
    hg.to(DEV)
    x = O.det_fill('tile_x%d.lr' % scale, (1, 3, lr_side, lr_side), 0.5, 0.5)
    hr = O.det_fill('tile_x%d.hr' % scale, (1, 3, lr_side * scale, lr_side * scale), 0.5, 0.5)
    yo = og(x)
    (yo - hr).abs().mean().backward()
    yh = hg(x.to(DEV))
    assert tuple(yh.shape) == (1, 3, lr_side * scale, lr_side * scale)
    (yh - hr.to(DEV)).abs().mean().backward()
    e = rel_err(yh, yo)
    s, k = grad_score((hg,), (og,), verbose=True)
    print('x%d @ LR %d: output rel err %.3e, worst gradient %s %.3e' % (scale, lr_side, e, k, s))
    assert e < 1e-3 and s < 5e-3, (e, s, k)


def test_post_step_weights_against_reference_vectors(golden):
    """train_full.npz holds 64 elements (of the digest sample) of nine weight / buffer tensors after the reference's own two
    iterations (B = 2, 54 -> 216).  The HIP path must land on them: every element within the 2 * lr * iters Adam bound,
    and all but a few within 2e-5 (elements whose gradient sign is roundoff move by +-lr on any two platforms)."""
    from sradsgan_amd.train_step import TrainStep
    g = golden('train_full')
    (hg, hd, hf), _ = build_pair(12, 3, 4, DEV)
    step = TrainStep(hg, hd, hf)
    for it in range(2):
        lr_img = O.det_fill('train_full.lr.%d' % it, (2, 3, 54, 54), 0.5, 0.5)
        hr_img = O.det_fill('train_full.hr.%d' % it, (2, 3, 216, 216), 0.5, 0.5)
        step(lr_img.to(DEV), hr_img.to(DEV), torch.from_numpy(g['alpha%d' % it]).to(DEV))
    gs, ds = hg.state_dict(), hd.state_dict()
    diffs, running = [], []
    for key in g.files:
        if key.startswith('G_after__') or key.startswith('D_after__'):
            sd = gs if key.startswith('G_') else ds
            name = key.split('__', 1)[1].replace('__', '.')
            got = O.digest(sd[name])[:64]                       # make_golden.py records digest(t)[:64] (large tensors: a strided sample)
            d = np.abs(got.astype(np.float64) - g[key].astype(np.float64))
            if 'running_' in name:
                running.append(float(d.max() / max(np.abs(g[key]).max(), 1e-6)))
            else:
                diffs.append(d)
                print('%-44s max |dw| %.3e  within 2e-5: %d / %d' % (key, d.max(), int((d <= 2e-5).sum()), d.size))
    alld = np.concatenate(diffs)
    assert alld.max() <= 2 * 2e-4 * 2 * 1.01 + 1e-7, alld.max()
    assert float((alld <= 2e-5).mean()) >= 0.97, float((alld <= 2e-5).mean())
    # running statistics after 8 updates: the last 4 see discriminator weights that already carry the +-lr sign-of-roundoff
    # steps above (one flipped element of model.2.weight moves its channel's batch mean by ~4e-5 of a ~3e-3 value)
    assert max(running) < 5e-2, running
