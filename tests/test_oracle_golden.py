"""Pins oracle/sradsgan_ref.py against vectors produced by the REFERENCE modules themselves
(oracle/make_golden.py, run in the build container against /root/reference).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import sradsgan_ref as O

TOL = dict(rtol=2e-4, atol=2e-6)


def _check(name, got, want, rtol=2e-4, atol=2e-6):
    got = O.digest(got)
    assert got.shape == want.shape, name
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol * scale, err_msg=name)


def _run_module(golden, tag, mod, x, grad_keys, **tol):
    g = golden(tag)
    O.det_init_(mod, prefix=tag + '.')
    x = x.clone().requires_grad_(True)
    y = mod(x)
    y.backward(O.det_fill(tag + '.dy', tuple(y.shape), 1.0))
    _check(tag + '.y', y, g['y'], **tol)
    _check(tag + '.dx', x.grad, g['dx'], **tol)
    params = dict(mod.named_parameters())
    for k in grad_keys:
        _check(tag + '.' + k, params[k].grad, g['grad__' + k.replace('.', '__')], **tol)


X64 = lambda: O.det_fill('x64', (2, 64, 10, 12), 1.0)
X3 = lambda: O.det_fill('x3', (2, 3, 10, 12), 0.5, 0.5)


def test_pixel_shuffle_index_exact(golden):
    g = golden('pixel_shuffle_index')
    for r in (2, 3):
        c, h, w = 2, 3, 4
        src = torch.arange(c * r * r * h * w, dtype=torch.float32).reshape(1, c * r * r, h, w)
        got = torch.nn.functional.pixel_shuffle(src, r).to(torch.int64).numpy()
        assert np.array_equal(got, g['r%d' % r])
        # closed form used by the HIP kernel: out[b,c,h*r+i,w*r+j] = in[b,c*r*r+i*r+j,h,w]
        b_, C, H, W = got.shape
        for cc in range(C):
            for hh in range(H):
                for ww in range(W):
                    i, j = hh % r, ww % r
                    assert got[0, cc, hh, ww] == src[0, cc * r * r + i * r + j, hh // r, ww // r]


def test_clam(golden):
    _run_module(golden, 'clam', O.CLAM(64), X64(), ['fc1.weight', 'fc2.weight'])


def test_slam(golden):
    _run_module(golden, 'slam', O.SLAM(7), X64(), ['conv1.weight'])


def test_cgam(golden):
    _run_module(golden, 'cgam', O.CGAM(64), X64() * 0.3, ['gamma'])


def test_sgam(golden):
    _run_module(golden, 'sgam', O.SGAM(64), X64(),
                ['gamma', 'query_conv.weight', 'key_conv.bias', 'value_conv.weight'])


def test_rab(golden):
    _run_module(golden, 'rab', O.RAB(64, 64), X64(),
                ['conv1.weight', 'conv2.bias', 'ca.fc1.weight', 'sa.conv1.weight', 'conv.weight'])


def test_resgroup(golden):
    _run_module(golden, 'resgroup', O.ResGroup(O.RAB, n_blocks=2), X64(),
                ['RG.1.conv2.weight', 'ca.fc2.weight', 'sa.conv1.weight', 'conv.bias'])


def test_msb(golden):
    _run_module(golden, 'msb', O.MSB(3, 64), X3(),
                ['conv1.weight', 'conv2.0.weight', 'conv2.1.bias', 'conv.weight'])


@pytest.mark.parametrize('s', [2, 3, 4, 8, 9])
def test_gab_up(golden, s):
    _run_module(golden, 'gabup_x%d' % s, O.GAB_UP(upscale_factor=s), X64()[:1, :, :6, :7] * 0.3,
                ['upsampling.0.weight', 'upsampling.0.bias', 'conv.weight', 'ca.gamma', 'sa.gamma'])


@pytest.mark.parametrize('s', [2, 3, 4, 8, 9])
def test_generator_small(golden, s):
    g = O.GeneratorResNet(O.ResGroup, n_residual_blocks=2, n_basic_blocks=1, upscale_factor=s)
    _run_module(golden, 'gen_small_x%d' % s, g, X3()[:1],
                ['conv1.0.weight', 'res_groups.0.RG.0.conv1.weight', 'res_groups.1.conv.weight',
                 'GAB_UP.upsampling.0.weight', 'MSB.conv.weight', 'conv3.0.bias'])


def test_generator_state_dict_contract():
    g = O.GeneratorResNet(O.ResGroup, upscale_factor=4)
    sd = g.state_dict()
    assert len(sd) == 412                                   # incl. the tied upsampler twice
    assert sum(p.numel() for p in g.parameters()) == 11069493
    assert g.GAB_UP.upsampling[0] is g.GAB_UP.upsampling[3]
    assert 'GAB_UP.upsampling.3.weight' in sd and 'res_groups.11.RG.2.sa.conv1.weight' in sd
    g3 = O.GeneratorResNet(O.ResGroup, upscale_factor=9)
    assert sum(p.numel() for p in g3.parameters()) == 11254133
    d = O.Discriminator()
    assert len(d.state_dict()) == 56 and sum(p.numel() for p in d.parameters()) == 4701987


def test_discriminator(golden):
    g = golden('disc')
    d = O.Discriminator()
    O.det_init_(d, prefix='D.')
    img = O.det_fill('dimg', (2, 3, 32, 32), 0.5, 0.5).requires_grad_(True)
    out = d(img)
    out.backward(O.det_fill('D.dy', tuple(out.shape), 1.0))
    sd = d.state_dict()
    _check('y', out, g['y'])
    _check('dx', img.grad, g['dx'], rtol=1e-3, atol=1e-5)
    for k, gk in [('model.3.running_mean', 'rm3'), ('model.3.running_var', 'rv3'),
                  ('model.23.running_mean', 'rm23'), ('model.23.running_var', 'rv23')]:
        _check(k, sd[k], g[gk])
    assert int(sd['model.3.num_batches_tracked']) == int(g['nbt'])
    params = dict(d.named_parameters())
    for k in ['model.0.weight', 'model.3.weight', 'model.17.fc1.weight', 'model.18.conv1.weight',
              'model.25.weight', 'model.22.bias']:
        _check(k, params[k].grad, g['grad__' + k.replace('.', '__')], rtol=1e-3, atol=1e-5)


def test_gradient_penalty(golden):
    g = golden('gradient_penalty')
    d = O.Discriminator()
    O.det_init_(d, prefix='D.')
    real = O.det_fill('gp.real', (2, 3, 32, 32), 0.5, 0.5)
    fake = O.det_fill('gp.fake', (2, 3, 32, 32), 0.5, 0.5)
    gp = O.gradient_penalty(d, real, fake, torch.from_numpy(g['alpha']))
    assert abs(gp.item() - float(g['gp'])) < 1e-5
    params = dict(d.named_parameters())
    for k in ['model.0.weight', 'model.3.weight', 'model.3.bias', 'model.11.weight', 'model.17.fc2.weight',
              'model.18.conv1.weight', 'model.25.weight']:
        _check(k, params[k].grad, g['grad__' + k.replace('.', '__')], rtol=2e-3, atol=2e-5)


def _train(golden, tag, n_groups, n_blocks, batch, lr_side, scale, iters):
    g = golden(tag)
    G = O.GeneratorResNet(O.ResGroup, n_residual_blocks=n_groups, n_basic_blocks=n_blocks, upscale_factor=scale)
    D, Fx = O.Discriminator(), O.FeatureExtractor()
    O.det_init_(G, prefix='G.'), O.det_init_(D, prefix='D.'), O.det_init_(Fx, prefix='F.')
    oG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.9, 0.999))
    oD = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.9, 0.999))
    for it in range(iters):
        lr_img = O.det_fill('%s.lr.%d' % (tag, it), (batch, 3, lr_side, lr_side), 0.5, 0.5)
        hr_img = O.det_fill('%s.hr.%d' % (tag, it), (batch, 3, lr_side * scale, lr_side * scale), 0.5, 0.5)
        s = O.train_step(G, D, Fx, oG, oD, lr_img, hr_img, torch.from_numpy(g['alpha%d' % it]))
        got = np.array([s['loss_G'], s['loss_D'], s['pixel'], s['content'], s['loss_gan'], s['gp']])
        np.testing.assert_allclose(got, g['scalars%d' % it], rtol=2e-4, atol=2e-5, err_msg='%s it%d' % (tag, it))
    gs, ds = G.state_dict(), D.state_dict()
    for k in ['conv1.0.weight', 'res_groups.0.RG.0.conv2.bias', 'GAB_UP.sa.gamma', 'GAB_UP.upsampling.0.weight',
              'conv3.0.weight']:
        _check(k, O.digest(gs[k])[:64], g['G_after__' + k.replace('.', '__')], rtol=1e-3, atol=2e-5)
    for k in ['model.0.weight', 'model.3.weight', 'model.3.running_mean', 'model.25.weight']:
        _check(k, O.digest(ds[k])[:64], g['D_after__' + k.replace('.', '__')], rtol=1e-3, atol=2e-5)


def test_train_two_iterations_small(golden):
    _train(golden, 'train_small', 2, 1, 2, 8, 4, 2)


def test_metric_quantisation_wraps_like_topilimage():
    t = torch.tensor([[[-0.02, 1.004, 0.5, 0.999]]]).expand(3, 1, 4)
    u8 = O.to_uint8_hwc(t)
    assert u8[0, :, 0].tolist() == [251, 0, 127, 254]      # -0.02 -> 251, 1.004 -> 0
    a = np.full((8, 8, 3), 10, np.uint8)
    b = np.full((8, 8, 3), 12, np.uint8)
    assert abs(O.psnr_u8(a, b) - 10 * np.log10(255 ** 2 / 4.0)) < 1e-12
    assert abs(O.ssim_u8(a, a) - 1.0) < 1e-12


def test_train_two_iterations_full_size(golden):
    """BASELINE config shape (x4, 54->216, 12 groups x 3 RAB) at B=2, reference scalars."""
    _train(golden, 'train_full', 12, 3, 2, 54, 4, 2)
