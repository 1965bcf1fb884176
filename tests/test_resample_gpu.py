"""Input-pipeline resampling on the device vs Pillow vectors and the oracle: bit-exact (integer work)."""
import os

import numpy as np
import pytest
import torch

from oracle import pil_resample_ref as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = torch.device('cuda:0')


def test_device_resize_matches_pillow_vectors_bit_exact():
    from sradsgan_amd import data
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pil_resample.npz'))
    i = 0
    while 'out%d' % i in g:
        img, want, filt = g['img_' + str(g['tag%d' % i])], g['out%d' % i], str(g['filter%d' % i])
        got = data.resize_u8(torch.from_numpy(img)[None].to(DEV), want.shape[0], want.shape[1], filt)[0].cpu().numpy()
        assert np.array_equal(got, want), (i, img.shape, want.shape, filt)
        i += 1
    assert i >= 9


def test_device_training_batch_matches_reference_pipeline():
    """A batch of HR tiles -> (lr, hr, bc) exactly as RGB_TrainDatasetFromFolder.__getitem__ + default collate
    produce them (dataset.py:418-436), at the real tile size (216 -> 54 -> 216) and for x2 / x3."""
    from sradsgan_amd import data
    rng = np.random.RandomState(7)
    hr = rng.randint(0, 256, size=(3, 216, 216, 3)).astype(np.uint8)
    hr[1] = (127.5 + 127.5 * np.sin(np.arange(216)[None, :, None] / 3.0) * np.cos(np.arange(216)[:, None, None] / 4.0)).astype(np.uint8)
    for scale in (4, 2, 3):
        lr, h, bc = data.training_batch(torch.from_numpy(hr).to(DEV), scale)
        for b in range(hr.shape[0]):
            wl, wh, wb = R.training_triplet(hr[b], scale)
            assert torch.equal(lr[b].cpu(), torch.from_numpy(wl)), (scale, b)
            assert torch.equal(h[b].cpu(), torch.from_numpy(wh)), (scale, b)
            assert torch.equal(bc[b].cpu(), torch.from_numpy(wb)), (scale, b)
    assert lr.shape == (3, 3, 72, 72) and lr.is_contiguous(memory_format=torch.channels_last)


def test_training_batch_feeds_the_generator():
    from sradsgan_amd import data, model as M
    hr = torch.randint(0, 256, (2, 48, 40, 3), dtype=torch.uint8, device=DEV)
    lr, h, bc = data.training_batch(hr, 4)
    G = M.GeneratorResNet(M.ResGroup, n_residual_blocks=1, n_basic_blocks=1, upscale_factor=4).to(DEV).eval()
    with torch.no_grad():
        out = G(lr)
    assert out.shape == h.shape == bc.shape
