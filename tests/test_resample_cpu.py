"""Input-pipeline resampling, CPU side: the oracle (oracle/pil_resample_ref.py) against vectors produced by Pillow
itself (tests/golden/pil_resample.npz, oracle/make_golden_resample.py), and the library's host-side coefficient routine
(srhip_resample_coeffs) against the oracle -- no device work."""
import ctypes
import os

import numpy as np
import pytest

from oracle import pil_resample_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cases():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'pil_resample.npz'))
    i = 0
    while 'out%d' % i in g:
        yield g['img_' + str(g['tag%d' % i])], g['out%d' % i], str(g['filter%d' % i])
        i += 1


def test_oracle_matches_pillow_vectors_bit_exact():
    n = 0
    for img, want, filt in _cases():
        got = R.resize_u8(img, want.shape[0], want.shape[1], filt)
        assert got.dtype == np.uint8 and np.array_equal(got, want), (img.shape, want.shape, filt)
        n += 1
    assert n >= 9


def test_oracle_against_installed_pillow_when_present():
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, size=(45, 38, 3)).astype(np.uint8)
    for (oh, ow), filt, pf in (((15, 19), 'bicubic', Image.BICUBIC), ((90, 57), 'bicubic', Image.BICUBIC), ((9, 38), 'bilinear', Image.BILINEAR)):
        want = np.asarray(Image.fromarray(img).resize((ow, oh), pf))
        assert np.array_equal(R.resize_u8(img, oh, ow, filt), want)


@pytest.mark.parametrize('case', [(216, 54, 'bicubic'), (54, 216, 'bicubic'), (216, 72, 'bicubic'), (61, 23, 'bicubic'),
                                  (37, 80, 'bicubic'), (216, 54, 'bilinear'), (61, 20, 'bilinear'), (7, 7, 'bicubic')])
def test_host_coefficients_match_oracle(case):
    from sradsgan_amd import _hip, data
    in_size, out_size, filt = case
    lib = _hip.lib()
    ksize = lib.srhip_resample_ksize(in_size, out_size, data.FILTERS[filt])
    k_ref, b_ref, c_ref = R.precompute_coeffs(in_size, out_size, filt)
    assert ksize == k_ref
    bounds = np.empty(2 * out_size, np.int32)
    coeffs = np.empty(out_size * ksize, np.int32)
    assert lib.srhip_resample_coeffs(in_size, out_size, data.FILTERS[filt], ctypes.c_void_p(bounds.ctypes.data),
                                     ctypes.c_void_p(coeffs.ctypes.data)) == 0
    assert np.array_equal(bounds.reshape(-1, 2), b_ref)
    assert np.array_equal(coeffs.reshape(out_size, ksize), c_ref)


def test_training_triplet_shapes_and_scaling():
    rng = np.random.RandomState(1)
    hr = rng.randint(0, 256, size=(24, 24, 3)).astype(np.uint8)
    lr, h, bc = R.training_triplet(hr, 4)
    assert lr.shape == (3, 6, 6) and h.shape == (3, 24, 24) and bc.shape == (3, 24, 24)
    assert lr.dtype == np.float32 and float(h.max()) <= 1.0 and np.array_equal(h, hr.transpose(2, 0, 1).astype(np.float32) / np.float32(255))
