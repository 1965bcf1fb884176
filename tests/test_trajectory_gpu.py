"""north_star's PSNR clause over a TRAJECTORY, not over two Adam steps: 30 training iterations of the small generator on the
HIP path (default split-bf16 arithmetic) and on the CPU oracle (the reference's loop, sradsgan.py:818-892, restated in
oracle/sradsgan_ref.train_step) from identical weights over an identical batch sequence, then the validation metric of
sradsgan.py:1314-1325 (uint8 quantisation, PSNR) on a held-out tile.  Both runs are fp32 training runs of a chaotic system
(LeakyReLU masks, arg-max ties, Adam's sign-like first steps), so the loss curves are required to agree to 1e-3 only over the
first iterations and to stay inside a growing envelope afterwards; what must hold at the END is the clause itself: the PSNR
the two generators reach differs by less than 0.05 dB."""
import math

import numpy as np
import pytest
import torch

from oracle import sradsgan_ref as O
from tests.parity_util import build_pair

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
ITERS, BATCH, LR_SIDE, SCALE = 30, 2, 8, 4
NAMES = ['loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp']


def _tile(tag, n):
    """Smooth synthetic HR tiles (a few low-frequency waves per channel + 5 % deterministic noise), LR = 4x4 box average:
    there is something for the generator to learn, so the PSNR moves over the 30 iterations."""
    side = LR_SIDE * SCALE
    yy, xx = torch.meshgrid(torch.arange(side, dtype=torch.float32), torch.arange(side, dtype=torch.float32), indexing='ij')
    ph = O.det_fill(tag + '.phase', (n, 3, 4), 3.14159, 0.0)
    hr = torch.zeros(n, 3, side, side)
    for b in range(n):
        for c in range(3):
            p = ph[b, c]
            hr[b, c] = 0.5 + 0.2 * torch.sin(0.21 * xx + p[0]) * torch.cos(0.17 * yy + p[1]) + 0.15 * torch.sin(0.09 * (xx + yy) + p[2])
    hr = (hr + 0.05 * O.det_fill(tag + '.noise', (n, 3, side, side), 0.5, 0.0)).clamp(0.0, 1.0)
    lr = torch.nn.functional.avg_pool2d(hr, SCALE)
    return lr, hr


def _psnr(gen, lr, hr, device):
    gen.eval()
    with torch.no_grad():
        out = gen(lr.to(device)).cpu()
    gen.train()
    return [O.psnr_u8(O.to_uint8_hwc(hr[b]), O.to_uint8_hwc(out[b])) for b in range(out.shape[0])]


def test_thirty_iteration_trajectory_and_final_psnr_against_the_oracle():
    from sradsgan_amd.train_step import TrainStep
    (hg, hd, hf), (og, od, of) = build_pair(2, 1, SCALE, DEV)
    step = TrainStep(hg, hd, hf)
    oG = torch.optim.Adam(og.parameters(), lr=2e-4, betas=(0.9, 0.999))
    oD = torch.optim.Adam(od.parameters(), lr=2e-4, betas=(0.9, 0.999))
    batches = [_tile('traj.b%d' % i, BATCH) for i in range(4)]          # a fixed sequence, cycled
    lr_t, hr_t = _tile('traj.heldout', 2)
    p0 = _psnr(og, lr_t, hr_t, 'cpu')
    diffs = []
    for it in range(ITERS):
        lr_img, hr_img = batches[it % len(batches)]
        alpha = O.det_fill('traj.alpha.%d' % it, (BATCH, 1, 1, 1), 0.5, 0.5)
        want = O.train_step(og, od, of, oG, oD, lr_img, hr_img, alpha)
        got = step(lr_img.to(DEV), hr_img.to(DEV), alpha.to(DEV))
        gv = np.array([float(got[k]) for k in NAMES])
        wv = np.array([want[k] for k in NAMES])
        assert np.isfinite(gv).all() and np.isfinite(wv).all()
        diffs.append(np.abs(gv - wv) / np.maximum(1.0, np.abs(wv)))
    diffs = np.array(diffs)
    worst = diffs.max(axis=1)
    sep = next((i for i, d in enumerate(worst) if d > 1e-3), None)
    p_ora, p_hip = _psnr(og, lr_t, hr_t, 'cpu'), _psnr(hg, lr_t, hr_t, DEV)
    print('trajectory: worst scalar distance per iteration', ' '.join('%d:%.1e' % (i, d) for i, d in enumerate(worst)))
    print('trajectory: first iteration above 1e-3: %s; held-out PSNR before %s, after: oracle %s, HIP %s' % (
        sep, ['%.3f' % v for v in p0], ['%.3f' % v for v in p_ora], ['%.3f' % v for v in p_hip]))
    assert max(abs(a - b) for a, b in zip(p_ora, p_hip)) < 0.05          # north_star: PSNR within 0.05 dB of the reference
    assert max(abs(a - b) for a, b in zip(p_ora, p0)) > 0.05             # the run did train: the metric moved by more than the bar
    assert worst[:10].max() < 1e-3                                       # the curves coincide over the first ten iterations
    assert worst.max() < 2e-2                                            # and never separate (no divergence, no sign of a wiring error)
