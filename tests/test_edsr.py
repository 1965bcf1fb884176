"""EDSR (BASELINE configs[0]): the oracle against vectors recorded from the reference's
model/edsr.py (CPU), and the HIP mirror against the oracle and the same vectors (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import edsr_ref as E
from oracle import sradsgan_ref as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden(scale):
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'edsr_x%d.npz' % scale))


def _run(net, scale, device='cpu'):
    x = O.det_fill('edsr.x.%d' % scale, (2, 3, 12, 10), 0.5, 0.5).to(device)
    tgt = O.det_fill('edsr.t.%d' % scale, (2, 3, 12 * scale, 10 * scale), 0.5, 0.5).to(device)
    y = net(x)
    loss = (y - tgt).abs().mean()
    loss.backward()
    return y, loss


@pytest.mark.parametrize('scale', [2, 3, 4])
def test_oracle_matches_reference_vectors(scale):
    g = _golden(scale)
    net = E.Net(3, 256, 2, scale)
    O.det_init_(net, prefix='E.')
    assert sorted(net.state_dict().keys()) == list(g['keys'])
    y, loss = _run(net, scale)
    assert np.abs(y.detach().numpy() - g['y']).max() < 2e-6 and abs(float(loss.detach()) - float(g['loss'])) < 1e-6
    for k, p in net.named_parameters():
        key = 'grad__' + k.replace('.', '__')
        if key in g:
            d = O.digest(p.grad)
            assert np.abs(d - g[key]).max() <= 1e-5 * max(1.0, np.abs(g[key]).max()), k


@pytest.mark.gpu
@pytest.mark.parametrize('scale', [2, 3, 4])
def test_hip_edsr_matches_oracle_and_reference_vectors(scale):
    from sradsgan_amd.model import edsr as H
    dev = torch.device('cuda:0')
    g = _golden(scale)
    ref = E.Net(3, 256, 2, scale)
    O.det_init_(ref, prefix='E.')
    net = H.Net(3, 256, 2, scale)
    assert sorted(net.state_dict().keys()) == list(g['keys'])            # the reference's key set
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.to(dev)
    y, loss = _run(net, scale, dev)
    yr, lr_ = _run(ref, scale)
    assert float((y.cpu() - torch.from_numpy(g['y'])).abs().max()) < 1e-4      # vs the reference itself
    assert abs(float(loss) - float(g['loss'])) < 1e-4 and abs(float(loss) - float(lr_)) < 1e-4
    # gradients: a fixed linear functional of the output (the L1 loss's sign(y - t) flips on roundoff), against an FP64
    # run of the oracle: on the CPUs tried, stock torch's fp32 conv backward is itself 4e-2 away from fp64 on the
    # 256 -> 1024 upsampler weight of this net (every layer upstream 5e-3) while the forward agrees to 1e-7, so the fp32
    # CPU numbers cannot referee a 1e-3 comparison.  The HIP path sits at 2e-6 (fp32 mode) / 7e-6 (bf16x3) from fp64.
    r = O.det_fill('edsr.r.%d' % scale, tuple(g['y'].shape), 1.0, 0.0)
    x = O.det_fill('edsr.x.%d' % scale, (2, 3, 12, 10), 0.5, 0.5)
    ref64 = E.Net(3, 256, 2, scale)
    O.det_init_(ref64, prefix='E.')
    ref64 = ref64.double()
    net.zero_grad()
    (net(x.to(dev)) * r.to(dev)).sum().backward()
    (ref64(x.double()) * r.double()).sum().backward()
    refg = dict(ref64.named_parameters())
    errs = {k: float((p.grad.cpu().double() - refg[k].grad).abs().max() / refg[k].grad.abs().max()) for k, p in net.named_parameters()}
    print({k: '%.1e' % v for k, v in errs.items()})
    # x2 / x3: no activation sits close enough to zero to flip, the agreement is at roundoff level.  x4 (two tied
    # stages, ~1 M LeakyReLU inputs): one pre-activation within 1e-6 of zero takes the other branch on the device and
    # moves the tied weight's gradient by ~1/sqrt(#pixels) = 1e-2 (every conv op of these shapes is within 6e-6 of
    # fp64 on its own, tools/ check); the bound there only guards the wiring.
    tol = 5e-5 if scale in (2, 3) else 3e-2
    for k, v in errs.items():
        assert v <= tol, (k, v)
