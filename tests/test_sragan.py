"""SRAGAN generator: the oracle against vectors recorded from the reference's model/sragan.py
GeneratorResNet (CPU), the HIP mirror against the same vectors, and one SRAGAN training iteration (the SRADSGAN loop
with this generator) against the oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import sradsgan_ref as O
from oracle import sragan_ref as A
from tests.parity_util import sibling_grad_check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden(scale):
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'sragan_x%d.npz' % scale))


def _case(net, scale, device='cpu', dtype=torch.float32):
    x = O.det_fill('sragan.x.%d' % scale, (2, 3, 12, 10), 0.5, 0.5).to(device, dtype)
    tgt = O.det_fill('sragan.t.%d' % scale, (2, 3, 12 * scale, 10 * scale), 0.5, 0.5).to(device, dtype)
    net.zero_grad()
    y = net(x)
    loss = torch.nn.functional.l1_loss(y, tgt)
    loss.backward()
    return y.detach(), float(loss.detach())


def _check(net, g, rtol):
    worst = 0.0
    for k, p in net.named_parameters():
        key = 'grad__' + k.replace('.', '__')
        if key in g:
            d = np.abs(O.digest(p.grad) - g[key]).max() / max(1.0, np.abs(g[key]).max())
            worst = max(worst, float(d))
            assert d <= rtol, (k, d)
    for k, b in net.named_buffers():
        ref = g['buf__' + k.replace('.', '__')]
        assert np.abs(O.digest(b.float()) - ref).max() <= rtol * max(1.0, np.abs(ref).max()), k
    return worst


@pytest.mark.parametrize('scale', [2, 3, 4])
def test_oracle_matches_reference_vectors(scale):
    g = _golden(scale)
    net = A.GeneratorResNet(n_residual_blocks=2, n_basic_blocks=3, upscale_factor=scale)
    O.det_init_(net, prefix='A.')
    assert sorted(net.state_dict().keys()) == list(g['keys'])
    y, loss = _case(net, scale)
    assert np.abs(y.numpy() - g['y']).max() < 2e-6 and abs(loss - float(g['loss'])) < 1e-6
    _check(net, g, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('scale', [2, 3, 4])
def test_hip_generator_matches_reference_vectors(scale):
    from sradsgan_amd.model import sragan as H
    dev = torch.device('cuda:0')
    g = _golden(scale)
    ref = A.GeneratorResNet(n_residual_blocks=2, n_basic_blocks=3, upscale_factor=scale)
    O.det_init_(ref, prefix='A.')
    net = H.GeneratorResNet(H.ResidualBlock_Block_WithAttention, n_residual_blocks=2, n_basic_blocks=3,
                            upscale_factor=scale)
    assert sorted(net.state_dict().keys()) == list(g['keys'])            # the reference's key set
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.to(dev)
    y, loss = _case(net, scale, dev)
    assert float((y.cpu() - torch.from_numpy(g['y'])).abs().max()) < 1e-4 and abs(loss - float(g['loss'])) < 1e-4
    # gradients: tests/parity_util.sibling_grad_check (reference vectors OR an fp64 oracle run at 1e-3; the wiring bound
    # only where the fp32 oracle itself shows a pre-activation within roundoff of a LeakyReLU kink).  x3 is the case that
    # motivated it: the up-sampler BatchNorm output closest to zero is 7e-7, and there it is the fp32 CPU run behind
    # the recorded vectors that lands on the other side of the fp64 run; the HIP path agrees with fp64 to 3e-4.
    ref64 = A.GeneratorResNet(n_residual_blocks=2, n_basic_blocks=3, upscale_factor=scale)
    O.det_init_(ref64, prefix='A.')
    rep = sibling_grad_check(net, g, ref, ref64.double(), lambda m, d, dt: _case(m, scale, d, dt))
    print('x%d gradient report: %s' % (scale, rep))
    for k, b in net.named_buffers():                                     # BatchNorm running statistics
        want = g['buf__' + k.replace('.', '__')]
        assert np.abs(O.digest(b.float()) - want).max() <= 1e-3 * max(1.0, np.abs(want).max()), k


@pytest.mark.gpu
def test_sragan_training_iteration_matches_oracle():
    """SRAGAN's iteration is SRADSGAN's (sragan.py:539-575 == sradsgan.py:829-892) around this generator: TrainStep
    with the SRAGAN generator against oracle.sradsgan_ref.train_step with the oracle's, two iterations."""
    from sradsgan_amd.model import sragan as H
    from sradsgan_amd.train_step import TrainStep
    dev = torch.device('cuda:0')
    og = A.GeneratorResNet(n_residual_blocks=2, n_basic_blocks=3, upscale_factor=4)
    od, of = O.Discriminator(), O.FeatureExtractor()
    O.det_init_(og, prefix='A.')
    O.det_init_(od, prefix='D.')
    O.det_init_(of, prefix='V.')
    for p in of.parameters():
        p.requires_grad_(False)
    G = H.GeneratorResNet(H.ResidualBlock_Block_WithAttention, n_residual_blocks=2, n_basic_blocks=3, upscale_factor=4)
    D, Fx = H.Discriminator(), H.FeatureExtractor()
    for m, r in ((G, og), (D, od), (Fx, of)):
        m.load_state_dict(r.state_dict(), strict=True)
        m.to(dev)
    step = TrainStep(G, D, Fx)
    oG = torch.optim.Adam(og.parameters(), lr=2e-4, betas=(0.9, 0.999))
    oD = torch.optim.Adam(od.parameters(), lr=2e-4, betas=(0.9, 0.999))
    for it in range(2):
        lr_img = O.det_fill('sragan.step.lr.%d' % it, (4, 3, 16, 16), 0.5, 0.5)
        hr_img = O.det_fill('sragan.step.hr.%d' % it, (4, 3, 64, 64), 0.5, 0.5)
        alpha = O.det_fill('sragan.step.alpha.%d' % it, (4, 1, 1, 1), 0.5, 0.5)
        want = O.train_step(og, od, of, oG, oD, lr_img, hr_img, alpha)
        got = step(lr_img.to(dev), hr_img.to(dev), alpha.to(dev))
        got = {k: float(got[k]) for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')}
        for k in got:
            assert abs(got[k] - want[k]) <= 1e-3 * max(1.0, abs(want[k])), (it, k, got[k], want[k])
