"""HIP model graph vs the CPU oracle (pinned to the reference by tests/golden) on identical
deterministic weights and inputs.  Tolerances: 1e-3 relative in fp32 (BASELINE.json north_star);
observed differences are ~1e-5 because the MFMA path is exact fp32."""
import numpy as np
import pytest
import torch

from oracle import sradsgan_ref as O
from tests.parity_util import ZERO_GRAD_KEYS, build_pair, grad_score, rel_err, train_parity

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
TOL = 1e-3
FLOOR = 1e-4          # gradients that are mathematically zero (e.g. SGAM key bias) are pure roundoff


def _close(got, want, tol=TOL, msg=''):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    scale = max(float(np.abs(want).max()), FLOOR)
    err = float(np.abs(got - want).max())
    assert err <= tol * scale, '%s: max abs err %.3e vs scale %.3e' % (msg, err, scale)


def _close_but_for_flips(got, want, msg=''):
    """Parameter gradients of the default split-bf16 arithmetic on the x8 / x9 generators (three / two tied up-sampling stages,
    ~1e6 LeakyReLU inputs behind one weight tensor): a pre-activation within the arithmetic's ~5e-6 of zero takes the other slope
    than in the oracle, which moves the gradients of the few weights behind that one element by a finite amount (tools/_x9diag.py:
    0.7 % of the up-sampler's weight gradients, ONE bias element, 2.8e-3 of the tensor's scale; the same graph in exact-fp32 conv
    arithmetic meets 1e-3 everywhere and is asserted right next to this).  Bar: 99 % of the elements within TOL, none beyond 1e-2."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    scale = max(float(np.abs(want).max()), FLOOR)
    err = np.abs(got - want)
    frac = float((err > TOL * scale).mean())
    assert frac <= 0.01 and float(err.max()) <= 1e-2 * scale, \
        '%s: %.2f %% of the elements beyond %.0e, max abs err %.3e vs scale %.3e' % (msg, 100 * frac, TOL, float(err.max()), scale)


def _module_case(golden, tag, hip_mod, ora_mod, x, grad_keys, flips=False):
    close_grad = _close_but_for_flips if flips else _close
    g = golden(tag)
    O.det_init_(ora_mod, prefix=tag + '.')
    hip_mod.load_state_dict(ora_mod.state_dict(), strict=True)
    hip_mod.to(DEV)
    xo = x.clone().requires_grad_(True)
    xh = x.clone().to(DEV).requires_grad_(True)
    dy = None
    yo = ora_mod(xo)
    yh = hip_mod(xh)
    dy = O.det_fill(tag + '.dy', tuple(yo.shape), 1.0)
    yo.backward(dy)
    yh.backward(dy.to(DEV))
    _close(yh.detach().cpu(), yo.detach(), msg='y vs oracle')
    _close(xh.grad.cpu(), xo.grad, msg='dx vs oracle')
    # and against the vectors recorded from the reference itself
    _close(O.digest(yh), g['y'], msg='y vs reference')
    _close(O.digest(xh.grad), g['dx'], msg='dx vs reference')
    hp, op = dict(hip_mod.named_parameters()), dict(ora_mod.named_parameters())
    for k in grad_keys:
        if k.endswith(ZERO_GRAD_KEYS):
            # identically zero in exact arithmetic (SGAM's key bias: soft-max shift invariance): every platform sees only the
            # roundoff of a cancelling sum over all pixels, so the yardstick is the sibling weight's gradient, not the value
            scale = float(op[k.replace('bias', 'weight')].grad.abs().max())
            err = float(hp[k].grad.cpu().abs().max())
            assert err <= TOL * scale, '%s: |roundoff| %.3e vs weight-gradient scale %.3e' % (k, err, scale)
            continue
        close_grad(hp[k].grad.cpu(), op[k].grad, msg=k + ' vs oracle')
        close_grad(O.digest(hp[k].grad), g['grad__' + k.replace('.', '__')], msg=k + ' vs reference')


X64 = lambda: O.det_fill('x64', (2, 64, 10, 12), 1.0)
X3 = lambda: O.det_fill('x3', (2, 3, 10, 12), 0.5, 0.5)


def test_clam(golden):
    from sradsgan_amd import model as M
    _module_case(golden, 'clam', M.CLAM(64), O.CLAM(64), X64(), ['fc1.weight', 'fc2.weight'])


def test_slam(golden):
    from sradsgan_amd import model as M
    _module_case(golden, 'slam', M.SLAM(7), O.SLAM(7), X64(), ['conv1.weight'])


def test_cgam(golden):
    from sradsgan_amd import model as M
    _module_case(golden, 'cgam', M.CGAM(64), O.CGAM(64), X64() * 0.3, ['gamma'])


def test_sgam(golden):
    from sradsgan_amd import model as M
    _module_case(golden, 'sgam', M.SGAM(64), O.SGAM(64), X64(),
                 ['gamma', 'query_conv.weight', 'key_conv.bias', 'value_conv.weight'])


def test_rab(golden):
    from sradsgan_amd import model as M
    _module_case(golden, 'rab', M.RAB(64, 64), O.RAB(64, 64), X64(),
                 ['conv1.weight', 'conv2.bias', 'ca.fc1.weight', 'sa.conv1.weight', 'conv.weight'])


def test_nan_activation_surfaces_through_the_generator_attention_pools():
    """The CLAM / SLAM max pools of attn_tail.hip compare with NaN propagation like ATen's max (round 3 fixed only the
    discriminator's cbam.hip pools; with `v > mx` from -inf the generator's 48 pools dropped a NaN): one NaN activation must
    show up in the spatial max of its channel (CLAM) and in the channel max of its pixel (SLAM), whichever lane, segment or
    shuffle step meets it, and nowhere else."""
    from sradsgan_amd import ops
    g = torch.Generator().manual_seed(11)
    n, c, h, w = 2, 64, 13, 17
    fc1, fc2 = torch.randn(4, c, 1, 1, generator=g).to(DEV), torch.randn(c, 4, 1, 1, generator=g).to(DEV)
    w7, wc, bc = torch.randn(1, 2, 7, 7, generator=g).to(DEV), torch.randn(c, c, 1, 1, generator=g).to(DEV), torch.randn(c, generator=g).to(DEV)
    for (b, ch, y, x) in [(0, 0, 0, 0), (1, 63, 12, 16), (0, 21, 6, 9), (1, 34, 3, 1)]:
        u = torch.randn(n, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        u[b, ch, y, x] = float('nan')
        skip = torch.zeros_like(u)
        out, (avg, mx, arg, s_, pooled, argc, m) = ops._tail_forward(u, skip, fc1, fc2, w7, wc, bc)
        mx = mx.cpu()
        assert torch.isnan(mx[b, ch]) and int(torch.isnan(mx).sum()) == 1, 'CLAM max dropped or smeared the NaN'
        # SLAM pools y = s * u over channels; s of image b is NaN for every channel once the MLP has seen the NaN, so the
        # whole image b is NaN there and the other image is clean
        pm = pooled.view(n, h, w, 2)[..., 1].cpu()
        assert torch.isnan(pm[b]).all() and torch.isfinite(pm[1 - b]).all()
        assert torch.isnan(out[b]).all() and torch.isfinite(out[1 - b]).all()


@pytest.mark.parametrize('shape', [(2, 54, 54), (1, 27, 27), (3, 10, 12), (2, 13, 70), (1, 3, 5), (2, 108, 24)])
def test_inference_tail_is_bit_identical_to_the_training_forward(shape):
    """srhip_attn_tail_eval (round 4: pooling partials + ONE fused kernel -- MLP, pooled map with halo, 7x7 conv, 1x1 conv on the
    MFMA, gate + bias + skip) against the training-mode launches (srhip_attn_tail_fwd + the 1x1 conv with both scales folded):
    every step keeps their arithmetic and order, so the outputs must be equal bit for bit, on ragged images (edge tiles, halo
    outside the image), with and without the 1x1 conv's bias, for the whole RAB as well."""
    from sradsgan_amd import ops, model as M
    n, h, w = shape
    g = torch.Generator().manual_seed(h * 100 + w)
    c = 64
    u = torch.randn(n, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    skip = torch.randn(n, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    fc1 = torch.nn.Parameter((torch.randn(4, c, 1, 1, generator=g) * 0.3).to(DEV))
    fc2 = torch.nn.Parameter((torch.randn(c, 4, 1, 1, generator=g) * 0.3).to(DEV))
    w7 = torch.nn.Parameter((torch.randn(1, 2, 7, 7, generator=g) * 0.2).to(DEV))
    wc = torch.nn.Parameter((torch.randn(c, c, 1, 1, generator=g) * 0.1).to(DEV))
    bc = torch.nn.Parameter(torch.randn(c, generator=g).to(DEV))
    from sradsgan_amd import _hip
    lib = _hip.lib()
    with ops.conv_math('bf16x3'):
        try:
            # small problems send the training path's 1x1 conv to the exact-fp32 register kernel: pin it to the split-bf16
            # LDS-DMA kernel the bench sizes take (srhip_debug_set(0, -1)), whose arithmetic the fused kernel reproduces
            lib.srhip_debug_set(0, -1)
            for bias in (bc, None):
                ref, _ = ops._tail_forward(u, skip, fc1, fc2, w7, wc, bias)
                with torch.no_grad():
                    got = ops.attention_tail(u, skip, fc1, fc2, w7, wc, bias)
                assert torch.equal(ref, got), float((ref - got).abs().max())
            torch.manual_seed(7)
            rab = M.RAB(64, 64).to(DEV)
            x = torch.randn(n, c, h, w, generator=g).to(DEV)
            y_train = rab(x).detach()
            with torch.no_grad():
                y_eval = rab(x)
            assert torch.equal(y_train, y_eval)
        finally:
            lib.srhip_debug_set(0, 0)


def test_resgroup(golden):
    from sradsgan_amd import model as M
    _module_case(golden, 'resgroup', M.ResGroup(M.RAB, n_blocks=2), O.ResGroup(O.RAB, n_blocks=2), X64(),
                 ['RG.1.conv2.weight', 'ca.fc2.weight', 'sa.conv1.weight', 'conv.bias'])


def test_msb(golden):
    from sradsgan_amd import model as M
    _module_case(golden, 'msb', M.MSB(3, 64), O.MSB(3, 64), X3(),
                 ['conv1.weight', 'conv2.0.weight', 'conv2.1.bias', 'conv.weight'])


@pytest.mark.parametrize('s', [2, 3, 4, 8, 9])
def test_gab_up(golden, s):
    from sradsgan_amd import model as M
    _module_case(golden, 'gabup_x%d' % s, M.GAB_UP(upscale_factor=s), O.GAB_UP(upscale_factor=s),
                 X64()[:1, :, :6, :7] * 0.3,
                 ['upsampling.0.weight', 'upsampling.0.bias', 'conv.weight', 'ca.gamma', 'sa.gamma'])


@pytest.mark.parametrize('s', [2, 3, 4, 8, 9])
def test_generator_small(golden, s):
    from sradsgan_amd import model as M, ops
    keys = ['conv1.0.weight', 'res_groups.0.RG.0.conv1.weight', 'res_groups.1.conv.weight',
            'GAB_UP.upsampling.0.weight', 'MSB.conv.weight', 'conv3.0.bias']
    mk = lambda: (M.GeneratorResNet(M.ResGroup, n_residual_blocks=2, n_basic_blocks=1, upscale_factor=s),
                  O.GeneratorResNet(O.ResGroup, n_residual_blocks=2, n_basic_blocks=1, upscale_factor=s))
    if s >= 8:
        # x8 / x9 (reference-generated vectors since round 5): the graph in exact-fp32 conv arithmetic meets the 1e-3 bar on every
        # tensor (kernels and wiring, tied stages included); the default arithmetic then only has to be right up to LeakyReLU flips
        with ops.conv_math('fp32'):
            _module_case(golden, 'gen_small_x%d' % s, *mk(), X3()[:1], keys)
        if ops.get_conv_math() != 'fp32':
            _module_case(golden, 'gen_small_x%d' % s, *mk(), X3()[:1], keys, flips=True)
        return
    _module_case(golden, 'gen_small_x%d' % s, *mk(), X3()[:1], keys)


def test_state_dict_keys_match_reference_contract():
    from sradsgan_amd import model as M
    g, og = M.GeneratorResNet(M.ResGroup, upscale_factor=4), O.GeneratorResNet(O.ResGroup, upscale_factor=4)
    assert list(g.state_dict().keys()) == list(og.state_dict().keys()) and len(g.state_dict()) == 412
    assert g.GAB_UP.upsampling[0] is g.GAB_UP.upsampling[3]
    d, od = M.Discriminator(), O.Discriminator()
    assert list(d.state_dict().keys()) == list(od.state_dict().keys()) and len(d.state_dict()) == 56
    assert list(M.FeatureExtractor().state_dict().keys()) == list(O.FeatureExtractor().state_dict().keys())


def test_discriminator_and_running_stats(golden):
    from sradsgan_amd import model as M
    g = golden('disc')
    od = O.Discriminator()
    O.det_init_(od, prefix='D.')
    hd = M.Discriminator()
    hd.load_state_dict(od.state_dict())
    hd.to(DEV)
    img = O.det_fill('dimg', (2, 3, 32, 32), 0.5, 0.5)
    xh = img.clone().to(DEV).requires_grad_(True)
    out = hd(xh)
    out.backward(O.det_fill('D.dy', tuple(out.shape), 1.0).to(DEV))
    _close(O.digest(out), g['y'], msg='y')
    _close(O.digest(xh.grad), g['dx'], 2e-3, msg='dx')
    sd = hd.state_dict()
    for k, gk in [('model.3.running_mean', 'rm3'), ('model.3.running_var', 'rv3'),
                  ('model.23.running_mean', 'rm23'), ('model.23.running_var', 'rv23')]:
        _close(O.digest(sd[k]), g[gk], msg=k)
    assert int(sd['model.3.num_batches_tracked']) == int(g['nbt'])
    hp = dict(hd.named_parameters())
    for k in ['model.0.weight', 'model.3.weight', 'model.17.fc1.weight', 'model.18.conv1.weight',
              'model.25.weight', 'model.22.bias']:
        _close(O.digest(hp[k].grad), g['grad__' + k.replace('.', '__')], 2e-3, msg=k)


def test_gradient_penalty_double_backward(golden):
    from sradsgan_amd import model as M
    from sradsgan_amd.train_step import TrainStep
    g = golden('gradient_penalty')
    od = O.Discriminator()
    O.det_init_(od, prefix='D.')
    hd = M.Discriminator()
    hd.load_state_dict(od.state_dict())
    hd.to(DEV)
    step = TrainStep(torch.nn.Linear(1, 1).to(DEV), hd, torch.nn.Linear(1, 1).to(DEV))
    real = O.det_fill('gp.real', (2, 3, 32, 32), 0.5, 0.5).to(DEV)
    fake = O.det_fill('gp.fake', (2, 3, 32, 32), 0.5, 0.5).to(DEV)
    gp = step.gradient_penalty(real, fake, torch.from_numpy(g['alpha']).to(DEV))
    gp.backward()
    assert abs(gp.item() - float(g['gp'])) < 1e-4
    hp = dict(hd.named_parameters())
    for k in ['model.0.weight', 'model.3.weight', 'model.3.bias', 'model.11.weight', 'model.17.fc2.weight',
              'model.18.conv1.weight', 'model.25.weight']:
        _close(O.digest(hp[k].grad), g['grad__' + k.replace('.', '__')], 5e-3, msg=k)


def test_gradient_penalty_backward_with_folded_batch_norm_sums_is_bit_identical():
    """ops.bn_fold_second_order (TrainStep._backward_terms): in the penalty's double backward (sradsgan.py:621-639, 886) a BatchNorm
    input's gradient from the first-order backward's node is held back and added by the forward node's own backward pass
    (srhip_bn_train_bwd_acc_xa) instead of by autograd: the same sums (a + b == b + a), so every discriminator gradient is bit-identical;
    a held-back gradient nobody takes is an error."""
    from sradsgan_amd import model as M, ops
    from sradsgan_amd.train_step import TrainStep
    torch.manual_seed(11)
    hd = M.Discriminator().to(DEV)
    step = TrainStep(torch.nn.Linear(1, 1).to(DEV), hd, torch.nn.Linear(1, 1).to(DEV))
    real, fake = torch.rand(3, 3, 48, 40, device=DEV), torch.rand(3, 3, 48, 40, device=DEV)
    alpha = torch.rand(3, 1, 1, 1, device=DEV)
    grads = []
    for fold in (True, False):
        for p in hd.parameters():
            p.grad = None
        gp = step.gradient_penalty(real, fake, alpha)
        if fold:
            assert ops._BN_FOLD
            with ops.bn_fold_second_order():
                gp.backward()
        else:
            gp.backward()
        grads.append([None if p.grad is None else p.grad.clone() for p in hd.parameters()])   # (the last conv's bias has no path to the penalty)
    assert sum(a is not None for a in grads[0]) >= 20
    assert all((a is None and b is None) or torch.equal(a, b) for a, b in zip(*grads))
    with pytest.raises(RuntimeError):
        with ops.bn_fold_second_order():
            ops._state.bn_fold[(1, (1,))] = torch.zeros(1)


def test_train_two_iterations_small(golden):
    worst, wdiff = train_parity(DEV, 'train_small', 2, 1, 2, 8, 4, 2, golden('train_small'))
    assert worst < TOL and wdiff < 5e-3, (worst, wdiff)


@pytest.mark.parametrize('scale,lr_side', [(2, 16), (3, 12), (8, 4), (9, 4)])
def test_train_two_iterations_other_scales(scale, lr_side):
    """BASELINE configs[4]'s other factors (x2 / x3 / x8 / x9: one, one, three and two up-sampler stages, r = 2 or 3,
    tied stage weights): two full iterations against the oracle's on identical weights and inputs."""
    from tests.parity_util import well_conditioned_tag
    # the input is picked by an oracle-only criterion (parity_util.well_conditioned_tag): at these tile sizes the
    # deepest discriminator BatchNorms normalise over 8 samples and a LeakyReLU input within ~1e-5 of zero there makes
    # the gradient comparison a coin flip between any two fp32 implementations
    tag = well_conditioned_tag('train_x%d' % scale, 1, 2, 2, lr_side, scale)
    worst, wdiff = train_parity(DEV, tag, 1, 2, 2, lr_side, scale, 2, None, fp64_ref=True)
    assert worst < TOL and wdiff < 5e-3, (worst, wdiff)


def test_near_tie_inputs_keep_the_losses_and_all_but_a_few_gradient_entries():
    """the tests above PICK inputs whose deep discriminator pre-activations stay away from the LeakyReLU kink.
    This one picks the opposite -- among the x2 candidates the input whose closest deep pre-activation is SMALLEST in the oracle
    (3.7e-6 for the first candidate: inside the ~5e-6 of either conv arithmetic) -- and asserts what must survive a flipped branch
    (sradsgan.py:470-508 under :829-892): the six loss scalars of both iterations stay within 1e-3 (the forward is continuous at the
    kink), the generator's gradients stay within the usual bar (the flip sits behind weight_gan = 1e-3), and the discriminator's
    gradients differ only LOCALLY: every tensor keeps >= 99 % of its elements within 2e-2, the network >= 99.9 % (measured: 99.94 % /
    99.991 %, while the max-norm score of the flipped layer's weight gradient reads 5.9e-2)."""
    from sradsgan_amd.train_step import TrainStep
    from tests.parity_util import closest_deep_preactivation, grad_fraction
    scale, lr_side, batch = 2, 16, 2
    cands = ['train_x2' + sfx for sfx in [''] + list('abcdef')]
    dist = {t: closest_deep_preactivation(t, 1, 2, batch, lr_side, scale) for t in cands}
    tag = min(dist, key=dist.get)
    print('near-tie input: %s, closest deep pre-activation %.2e (candidates: %s)' % (tag, dist[tag], {k: '%.1e' % v for k, v in dist.items()}))
    assert dist[tag] < 1e-5, 'no near-tie candidate left: extend the candidate list'
    (hg, hd, hf), (og, od, of) = build_pair(1, 2, scale, DEV)
    step = TrainStep(hg, hd, hf)
    oG = torch.optim.Adam(og.parameters(), lr=2e-4, betas=(0.9, 0.999))
    oD = torch.optim.Adam(od.parameters(), lr=2e-4, betas=(0.9, 0.999))
    names = ['loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp']
    for it in range(2):
        lr_img = O.det_fill('%s.lr.%d' % (tag, it), (batch, 3, lr_side, lr_side), 0.5, 0.5)
        hr_img = O.det_fill('%s.hr.%d' % (tag, it), (batch, 3, lr_side * scale, lr_side * scale), 0.5, 0.5)
        alpha = O.det_fill('%s.alpha.%d' % (tag, it), (batch, 1, 1, 1), 0.5, 0.5)
        want = O.train_step(og, od, of, oG, oD, lr_img, hr_img, alpha)
        got = step(lr_img.to(DEV), hr_img.to(DEV), alpha.to(DEV))
        worst = max(abs(float(got[k]) - want[k]) for k in names)
        sg, kg = grad_score((hg,), (og,))
        sd, kd = grad_score((hd,), (od,))
        fd, fk, fall = grad_fraction((hd,), (od,), 2e-2)
        fg, fgk, fgall = grad_fraction((hg,), (og,), 5e-3)
        print('near-tie it %d: scalars %.2e; G score %.3e (%s), D score %.3e (%s); D elements within 2e-2: worst tensor %.4f (%s), all %.5f; '
              'G elements within 5e-3: worst tensor %.4f (%s), all %.5f' % (it, worst, sg, kg, sd, kd, fd, fk, fall, fg, fgk, fgall))
        assert worst < TOL, (it, worst)
        if it == 0:                                      # identical weights on both sides: the clean comparison
            assert sg < 5e-3, (sg, kg)
            assert fd >= 0.99 and fall >= 0.999, (fd, fk, fall)


def test_train_two_iterations_full_size(golden):
    """x4, 54->216, 12 groups x 3 RAB, B=2: losses/PSNR-relevant scalars within 1e-3 of the reference; gradients of
    the first iteration scored against an fp64 evaluation of the same graph (parity_util.train_parity, fp64_ref)."""
    worst, wdiff = train_parity(DEV, 'train_full', 12, 3, 2, 54, 4, 2, golden('train_full'), fp64_ref=True)
    assert worst < TOL and wdiff < 5e-3, (worst, wdiff)


def test_adam_kernel_matches_torch():
    """srhip_adam_step (flat arena, fused clip) vs torch.optim.Adam + clamp_ on identical gradients
    (sradsgan.py:724-725, 858, 887, 891-892), three steps."""
    from sradsgan_amd.dp import ParamArena
    from sradsgan_amd.train_step import TrainStep
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 7, 3), torch.nn.Conv2d(7, 5, 1))
    ref = torch.nn.Sequential(torch.nn.Conv2d(3, 7, 3), torch.nn.Conv2d(7, 5, 1))
    ref.load_state_dict(net.state_dict())
    net.to(DEV)
    step = TrainStep(net, torch.nn.Linear(1, 1).to(DEV), torch.nn.Linear(1, 1).to(DEV))
    opt = torch.optim.Adam(ref.parameters(), lr=2e-4, betas=(0.9, 0.999))
    for it in range(3):
        for (p, q) in zip(net.parameters(), ref.parameters()):
            g = torch.randn(q.shape) * (10.0 ** (-it * 3))
            q.grad = g.clone()
            p.grad.copy_(g.to(DEV))
        opt.step()
        with torch.no_grad():
            for q in ref.parameters():
                q.clamp_(-0.05, 0.05)
        step._adam(step.arena_G, 2e-4, 0.05, 1.0)
        for (p, q) in zip(net.parameters(), ref.parameters()):
            assert float((p.detach().cpu() - q.detach()).abs().max()) < 2e-7, it


def test_vgg_features_fused_forward_and_input_gradient():
    """FeatureExtractor (vgg19.features[:12] structure, sradsgan.py:88-99) with frozen weights runs as one
    fused node (ReLU masks in the dgrad / max-pool epilogues): forward and d/d(img) vs the CPU oracle."""
    from sradsgan_amd import model as M
    of = O.FeatureExtractor()
    O.det_init_(of, prefix='F.')
    hf = M.FeatureExtractor()
    hf.load_state_dict(of.state_dict())
    hf.to(DEV)
    for p in hf.parameters():
        p.requires_grad_(False)
    img = O.det_fill('vgg.img', (2, 3, 24, 32), 0.5, 0.5)
    xo = img.clone().requires_grad_(True)
    xh = img.clone().to(DEV).requires_grad_(True)
    yo, yh = of(xo), hf(xh)
    dy = O.det_fill('vgg.dy', tuple(yo.shape), 1.0)
    yo.backward(dy)
    yh.backward(dy.to(DEV))
    _close(yh.detach().cpu(), yo.detach(), msg='features')
    _close(xh.grad.cpu(), xo.grad, msg='d/d(img)')


def test_generator_inference_psnr_matches_oracle():
    """BASELINE metric 'PSNR vs reference' (mfeNew_validate, sradsgan.py:1305-1325): full-size generator
    (12 groups x 3 RAB, x4, 54 -> 216) on identical weights/inputs; output within 1e-3, and PSNR / MSE /
    ERGAS of the uint8-quantised images (ToPILImage = mul(255).byte(), truncate + wrap) against a synthetic
    HR target within 0.05 dB of the oracle's."""
    from sradsgan_amd import model as M
    og = O.GeneratorResNet(O.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=4)
    O.det_init_(og, prefix='G.')
    hg = M.GeneratorResNet(M.ResGroup, n_residual_blocks=12, n_basic_blocks=3, upscale_factor=4)
    hg.load_state_dict(og.state_dict(), strict=True)
    hg.to(DEV).eval()
    og.eval()
    lr = O.det_fill('psnr.lr', (2, 3, 54, 54), 0.5, 0.5)
    hr = O.det_fill('psnr.hr', (2, 3, 216, 216), 0.5, 0.5)
    with torch.no_grad():
        yo = og(lr)
        yh = hg(lr.to(DEV)).cpu()
    assert rel_err(yh, yo) < TOL
    for b in range(2):
        ref_img, hip_img, tgt = O.to_uint8_hwc(yo[b]), O.to_uint8_hwc(yh[b]), O.to_uint8_hwc(hr[b])
        assert float(np.mean(ref_img != hip_img)) < 2e-3            # quantisation flips only at rounding boundaries
        assert abs(O.psnr_u8(tgt, hip_img) - O.psnr_u8(tgt, ref_img)) < 0.05
        assert abs(O.ergas2(tgt, hip_img) - O.ergas2(tgt, ref_img)) < 1e-2
        assert abs(O.ssim_u8(tgt, hip_img) - O.ssim_u8(tgt, ref_img)) < 1e-3


def test_device_validation_metrics_match_reference_arithmetic():
    """sradsgan_amd.validate.quantized_metrics (uint8 wrap quantisation, MSE, PSNR, ERGAS, SSIM on the device)
    vs the oracle's numpy restatement of sradsgan.py:1314-1325 / utils.py:923-962, incl. out-of-range pixels."""
    from sradsgan_amd.validate import quantized_metrics
    g = torch.Generator().manual_seed(9)
    sr = torch.rand(3, 3, 40, 56, generator=g) * 1.2 - 0.1        # some values < 0 and > 1: exercises the wrap
    hr = torch.rand(3, 3, 40, 56, generator=g)
    got = quantized_metrics(sr.to(DEV), hr.to(DEV), 4)
    for b in range(3):
        a_img, t_img = O.to_uint8_hwc(sr[b]), O.to_uint8_hwc(hr[b])
        assert abs(float(got['mse'][b]) - O.mse_u8(t_img, a_img)) < 1e-9
        assert abs(float(got['psnr'][b]) - O.psnr_u8(t_img, a_img)) < 1e-9
        assert abs(float(got['ergas'][b]) - O.ergas2(t_img, a_img, 4)) < 1e-9
        assert abs(float(got['ssim'][b]) - O.ssim_u8(a_img, t_img)) < 1e-9


def test_graphed_evaluator_matches_eager_across_replays():
    """The captured inference+metrics graph must reproduce the eager pass bit for bit on every replay, with device
    syncs and fresh inputs in between (the training-step graph is not the default because its replay was racy)."""
    from sradsgan_amd import model as M, validate
    torch.manual_seed(3)
    G = M.GeneratorResNet(M.ResGroup, n_residual_blocks=2, n_basic_blocks=1, upscale_factor=4).to(DEV).eval()
    ev = validate.GraphedEvaluator(G, 4)
    for it in range(4):
        lr = torch.rand(2, 3, 12, 10, device=DEV)
        hr = torch.rand(2, 3, 48, 40, device=DEV)
        want = validate.evaluate(G, lr, hr, 4)
        torch.cuda.synchronize()
        got = ev(lr, hr)
        torch.cuda.synchronize()
        assert torch.equal(got['recon'], want['recon']), it
        for k in ('mse', 'psnr', 'ssim', 'ergas'):
            assert torch.equal(got['sr'][k], want['sr'][k]), (it, k)


def test_checkpoint_resume_reproduces_the_next_step(tmp_path):
    """Weights (reference file format) + optimiser state saved after step 1 and loaded into a fresh trainer give the
    same step 2 as the uninterrupted run: arena views survive load_state_dict, packed weights are refreshed."""
    from sradsgan_amd import checkpoint as C
    from sradsgan_amd.train_step import TrainStep

    def make():
        (g, d, f), _ = build_pair(2, 1, 4, DEV)
        return g, d, f, TrainStep(g, d, f)
    gen = torch.Generator().manual_seed(9)
    batches = [(torch.rand(2, 3, 8, 8, generator=gen).to(DEV), torch.rand(2, 3, 32, 32, generator=gen).to(DEV),
                torch.rand(2, 1, 1, 1, generator=gen).to(DEV)) for _ in range(2)]
    g, d, f, step = make()
    step(*batches[0])
    C.save_epoch_network(str(tmp_path), g, 'generator', 1)
    C.save_epoch_network(str(tmp_path), d, 'discriminator', 1)
    C.save_optimizer_state(str(tmp_path / 'opt.pt'), step)
    want = {k: float(v) for k, v in step(*batches[1]).items() if k != 'gen_hr'}
    g2, d2, f2, step2 = make()
    C.load_epoch_network(C.epoch_path(str(tmp_path), 'generator', 1), g2)
    C.load_epoch_network(C.epoch_path(str(tmp_path), 'discriminator', 1), d2)
    C.load_optimizer_state(str(tmp_path / 'opt.pt'), step2)
    assert step2.arena_G.check_views() and step2.arena_D.check_views()
    got = {k: float(v) for k, v in step2(*batches[1]).items() if k != 'gen_hr'}
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-6 * max(1.0, abs(want[k])), (k, got[k], want[k])


def test_training_step_is_deterministic():
    """Two runs of the same two iterations from the same weights give bit-identical scalars and weights: split-K
    reductions are ordered, weight gradients of one parameter are accumulated on one stream in program order, and no
    kernel uses floating-point atomics -- the three-stream overlap does not change a single bit."""
    from sradsgan_amd.train_step import TrainStep
    results = []
    for run in range(2):
        (hg, hd, hf), _ = build_pair(2, 2, 4, DEV)
        step = TrainStep(hg, hd, hf)
        scal = []
        for it in range(2):
            lr_img = O.det_fill('det.lr.%d' % it, (4, 3, 24, 24), 0.5, 0.5).to(DEV)
            hr_img = O.det_fill('det.hr.%d' % it, (4, 3, 96, 96), 0.5, 0.5).to(DEV)
            alpha = O.det_fill('det.alpha.%d' % it, (4, 1, 1, 1), 0.5, 0.5).to(DEV)
            out = step(lr_img, hr_img, alpha)
            scal.append(torch.stack([out[k].double() for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')]).cpu())
        torch.cuda.synchronize()
        results.append((scal, [p.detach().cpu().clone() for p in list(hg.parameters()) + list(hd.parameters())],
                        [b.detach().cpu().clone() for b in hd.buffers()]))
    for a, b in zip(results[0][0], results[1][0]):
        assert torch.equal(a, b), (a, b)
    for group in (1, 2):
        for a, b in zip(results[0][group], results[1][group]):
            assert torch.equal(a, b)


def test_grouped_weight_gradient_launches_do_not_change_the_step():
    """TrainStep pairs the weight gradients of consecutive residual blocks into one launch (srhip_conv2d_wgrad_multi, wgrad_group =
    2).  Group sizes 1 (every conv on its own), 2 and 4 differ only in the split-K summation order of those gradients: two
    iterations of a 4 x 3 generator at the real x4 tile (B = 8: large enough for the row-tap kernel to be the one selected) must
    agree to summation-order level -- scalars 1e-6, first-iteration gradients 2e-5 of each tensor's scale -- and every pending
    gradient must have been flushed before the optimiser ran (identical parameter movement pattern: no tensor left untouched)."""
    from sradsgan_amd.train_step import TrainStep
    res = {}
    for group in (1, 2, 4):
        (hg, hd, hf), _ = build_pair(4, 3, 4, DEV)
        step = TrainStep(hg, hd, hf)
        step.wgrad_group = group
        scal = []
        for it in range(2):
            lr_img = O.det_fill('grp.lr.%d' % it, (8, 3, 54, 54), 0.5, 0.5).to(DEV)
            hr_img = O.det_fill('grp.hr.%d' % it, (8, 3, 216, 216), 0.5, 0.5).to(DEV)
            alpha = O.det_fill('grp.alpha.%d' % it, (8, 1, 1, 1), 0.5, 0.5).to(DEV)
            out = step(lr_img, hr_img, alpha)
            scal.append(torch.stack([out[k].double() for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')]).cpu())
            if it == 0:
                grads = {k: p.grad.detach().clone() for k, p in hg.named_parameters()}
        torch.cuda.synchronize()
        res[group] = (scal, grads)
    for group in (2, 4):
        for a, b in zip(res[group][0], res[1][0]):
            assert float((a - b).abs().max()) < 1e-6, (group, a, b)
        worst = max(float((res[group][1][k] - g1).abs().max() / g1.abs().max().clamp_min(1e-12)) for k, g1 in res[1][1].items()
                    if float(g1.abs().max()) > 0)
        print('wgrad_group %d vs 1: worst relative gradient difference %.2e' % (group, worst))
        assert worst < 2e-5, (group, worst)


@pytest.mark.parametrize('scale,batch,blocks,groups', [(4, 32, 12, 3), (2, 2, 2, 2), (3, 4, 2, 1), (8, 8, 2, 2), (9, 6, 2, 1)])
def test_first_step_of_a_model_does_not_depend_on_allocator_history(scale, batch, blocks, groups):
    """The first step of a model packs weights lazily on whichever stream needs them first (VGG's for the real batch on the
    weight-gradient stream, the discriminator's data-gradient images on the D stream); every other stream must wait for
    those pack kernels (ops._pack_fence).  Without the fence the main stream ran D(gen)'s data gradients on unpacked
    images: zeros on fresh memory -- a silently wrong first step --, NaNs on recycled memory.  Two identical models in one
    process, the second on recycled (NaN-poisoned) memory, at the bench configuration and at the other scales' real tile
    sizes: identical, finite first-step gradients and scalars."""
    from sradsgan_amd.train_step import TrainStep
    B, side = batch, 216 // scale
    lr = O.det_fill('first.lr', (B, 3, side, side), 0.5, 0.5).to(DEV)
    hr = O.det_fill('first.hr', (B, 3, side * scale, side * scale), 0.5, 0.5).to(DEV)
    al = O.det_fill('first.alpha', (B, 1, 1, 1), 0.5, 0.5).to(DEV)

    def run():
        (hg, hd, hf), _ = build_pair(blocks, groups, scale, DEV)
        step = TrainStep(hg, hd, hf)
        out = step(lr, hr, al)
        torch.cuda.synchronize()
        grads = [p.grad.detach().clone() for p in list(hg.parameters()) + list(hd.parameters())]
        return torch.stack([out[k].double() for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')]).cpu(), grads

    s1, g1 = run()
    keep = [torch.full(((256 << 20) // 4,), float('nan'), device=DEV) for _ in range(24)] + \
           [torch.full(((2 << 20) // 4,), float('nan'), device=DEV) for _ in range(64)]
    torch.cuda.synchronize()
    del keep
    s2, g2 = run()
    assert torch.isfinite(s1).all() and torch.isfinite(s2).all()
    assert all(bool(torch.isfinite(g).all()) for g in g1 + g2)
    assert torch.equal(s1, s2)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))


@pytest.mark.parametrize('op', ['attention_tail', 'conv_residual'])
def test_weight_gradient_on_a_lagging_side_stream_reads_the_gradient_it_was_launched_with(op):
    """The attention tail hands its incoming gradient back unchanged as the gradient of `skip`, a conv with a fused residual
    hands it back as the residual's; both launch a weight-gradient kernel that reads the same tensor on the side stream.
    Autograd's input buffer adds a later contribution to `skip` IN PLACE into a gradient it holds the only reference to -- on
    the main stream.  With the side stream held up (here by a few large matrix products queued on it before the backward; in
    the step by whatever else the chip is doing) the kernel read g + the other branch's gradient: weight gradients of the
    ResGroup tail convs off by 10-30 %, at random.  ops._hold_for_side keeps a second reference until the side stream has
    passed the kernel, so the engine allocates the sum instead.  Same inputs with the weight gradients in line (no side
    stream) and on a lagging side stream: identical parameter gradients."""
    from sradsgan_amd import ops
    n, h, w = 4, 54, 54
    g = torch.Generator().manual_seed(41)

    def cl(*shape, s=1.0):
        return (torch.randn(*shape, generator=g) * s).to(DEV).contiguous(memory_format=torch.channels_last)

    x0, u0, c1, c2 = cl(n, 64, h, w), cl(n, 64, h, w), cl(n, 64, h, w), cl(n, 64, h, w)
    shapes = {'attention_tail': [(4, 64, 1, 1), (64, 4, 1, 1), (1, 2, 7, 7), (64, 64, 1, 1), (64,)],
              'conv_residual': [(64, 64, 3, 3), (64,)]}[op]
    init = [(torch.randn(*s, generator=g) * 0.1).to(DEV) for s in shapes]
    big = torch.randn(8192, 8192, device=DEV)

    def run(side):
        ps = [torch.nn.Parameter(t.clone()) for t in init]
        for p in ps:
            p.grad = torch.zeros_like(p)
        x, u = x0.clone().requires_grad_(True), u0.clone().requires_grad_(True)
        skip = x * 1.0
        y2 = skip * 2.0                    # created before the op below: its gradient reaches skip's buffer AFTER the op's
        y1 = ops.attention_tail(u, skip, *ps) if op == 'attention_tail' else ops.conv2d(u, ps[0], ps[1], 1, 1, None, skip)
        loss = (y1 * c1).sum() + (y2 * c2).sum()
        torch.cuda.synchronize()
        if side is not None:
            with torch.cuda.stream(side):
                for _ in range(6):
                    big @ big              # ~60 ms of work in front of the weight-gradient kernels
        with ops.direct_param_grads(side):
            loss.backward()
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        return [p.grad.clone() for p in ps] + [x.grad.clone(), u.grad.clone()]

    ref = run(None)
    got = run(torch.cuda.Stream())
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (op, i, float((a - b).abs().max()), float(a.abs().max()))


@pytest.mark.parametrize('n_terms', [2, 3, 14, 16, 17])
def test_bus_sum_is_bit_identical_to_chained_adds(n_terms):
    """srhip_sum_n (the generator's stratified bus, sradsgan.py:455-460, as one pass) adds its terms in the order of the chained
    `bus = bus + out` it replaces: same bits, and every term receives the incoming gradient.  17 terms take the fallback."""
    from sradsgan_amd import ops
    g = torch.Generator().manual_seed(n_terms)
    ts = [torch.randn(2, 64, 13, 17, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
          for _ in range(n_terms)]
    ref = ts[0]
    for t in ts[1:]:
        ref = ref + t
    got = ops.sum_tensors(ts)
    assert torch.equal(ref, got)
    dy = torch.randn(2, 64, 13, 17, generator=g).to(DEV)
    got.backward(dy)
    assert all(torch.equal(t.grad, dy) for t in ts)


@pytest.mark.parametrize('shape', [(2, 54, 54), (3, 27, 27), (1, 13, 70), (2, 24, 24), (16, 54, 54)])
def test_conv_epilogue_pooling_partials_match_the_pooling_pass(shape):
    """srhip_conv2d_fwd_pool (ABI 8): RAB conv2 leaves the CLAM pooling partials of its output behind -- from the persistent patch
    kernel's epilogue (one partial per tile and wave row) when that kernel takes the launch, from the stand-alone pooling pass
    otherwise.  Reduced by the tail's MLP kernel both must give the pass's result: max and first arg-max pixel exactly (the merge
    is order-independent), the mean to rounding; the conv output itself is the plain call's, bit for bit.  Ragged tiles, tiles
    that hang over the image, a NaN, and ties (a constant channel: arg-max = pixel 0)."""
    from sradsgan_amd import ops, _hip
    lib = _hip.lib()
    n, h, w = shape
    g = torch.Generator().manual_seed(n * 1000 + h)
    t = torch.randn(n, 256, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    w2 = torch.nn.Parameter((torch.randn(64, 256, 3, 3, generator=g) * 0.05).to(DEV))
    w2.data[5] = 0.0                                                # channel 5: constant (= bias) everywhere: all pixels tie
    b2 = torch.randn(64, generator=g).to(DEV)
    fc1 = (torch.randn(4, 64, 1, 1, generator=g) * 0.3).to(DEV)
    fc2 = (torch.randn(64, 4, 1, 1, generator=g) * 0.3).to(DEV)
    w7 = (torch.randn(1, 2, 7, 7, generator=g) * 0.2).to(DEV)
    wc = (torch.randn(64, 64, 1, 1, generator=g) * 0.1).to(DEV)
    skip = torch.randn(n, 64, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    with ops.conv_math('bf16x3'):
        lib.srhip_debug_set(0, -2)                                  # small problems: send them to the patch family (the walk takes 64-wide tiles at any count)
        try:
            u_ref = ops.conv2d_fwd_raw(t, w2, b2, 1, 1)
            for grid in (3, 1, 8, 0):                               # blocks that walk MANY tiles: the three extra stores per tile are counted waits like the others
                lib.srhip_debug_set(5, grid)
                for rep in range(2):
                    u, pool = ops.conv2d_fwd_pool_raw(t, w2, b2)
                    assert torch.equal(u, u_ref), grid
                    _, sv = ops._tail_forward(u, skip, fc1, fc2, w7, wc, None, pool)
                    if grid == 3 and rep == 0:
                        first = sv
                    assert all(torch.equal(a, b) for a, b in zip(first[:3], sv[:3])), grid
            assert pool[2] != lib.srhip_clam_pool_segments() or h * w > 64 * 128, 'the epilogue did not serve the request'
            out_ref, saved_ref = ops._tail_forward(u_ref, skip, fc1, fc2, w7, wc, None)
            out, saved = ops._tail_forward(u, skip, fc1, fc2, w7, wc, None, pool)
            avg_r, mx_r, arg_r = saved_ref[0], saved_ref[1], saved_ref[2]
            avg, mx, arg = saved[0], saved[1], saved[2]
            assert torch.equal(mx, mx_r) and torch.equal(arg, arg_r)
            assert int(arg[0, 5]) == 0
            assert float((avg - avg_r).abs().max()) <= 2e-6 * float(avg_r.abs().max())
            assert float((out - out_ref).abs().max()) <= 1e-5 * float(out_ref.abs().max())
            with torch.no_grad():
                ev = ops._tail_forward_eval(u, skip, fc1, fc2, w7, wc, None, pool)
            assert float((ev - out_ref).abs().max()) <= 1e-5 * float(out_ref.abs().max())
            # a NaN input pixel poisons the 3x3 neighbourhood of u in every channel: it must surface in the pooled maximum
            t2 = t.clone()
            t2[0, 7, h // 2, w // 2] = float('nan')
            u2, pool2 = ops.conv2d_fwd_pool_raw(t2, w2, b2)
            _, saved2 = ops._tail_forward(u2, skip, fc1, fc2, w7, wc, None, pool2)
            assert torch.isnan(saved2[1][0]).all() and torch.isfinite(saved2[1][1:]).all()
        finally:
            lib.srhip_debug_set(0, 0)
            lib.srhip_debug_set(5, 0)


def test_group_input_gradients_folded_into_the_first_rab_match_autograd_sums():
    """ops.carry_open (round 5): a ResGroup input's gradients from the group skip and the trunk's bus are stashed and added by the
    first RAB's conv1 data gradient (srhip_conv2d_dgrad_res3) instead of two autograd add passes.  Same graph with the mechanism
    off (autograd sums): every gradient agrees to fp32 summation-order roundoff, nothing stays stashed, and the mechanism was used."""
    from sradsgan_amd import model as M, ops
    torch.manual_seed(11)
    net = M.GeneratorResNet(M.ResGroup, n_residual_blocks=3, n_basic_blocks=2, upscale_factor=2).to(DEV)
    x = torch.rand(2, 3, 24, 20, device=DEV)
    dy = torch.randn(2, 3, 48, 40, device=DEV)

    def run(carry):
        old, ops._CARRY = ops._CARRY, carry
        try:
            for p in net.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            t0 = ops._state.carry_token
            net(xi).backward(dy)
            return xi.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}, ops._state.carry_token - t0
        finally:
            ops._CARRY = old

    dx1, g1, used1 = run(True)
    assert used1 == 3 and not ops._state.carry and not ops._state.carry_expect        # three groups, everything consumed
    dx0, g0, used0 = run(False)
    assert used0 == 0
    _close(dx1.cpu(), dx0.cpu(), tol=1e-5, msg='dx')
    for k in g0:
        _close(g1[k].cpu(), g0[k].cpu(), tol=2e-5, msg=k)


def test_a_missing_stash_is_an_error_not_a_wrong_gradient():
    """The first RAB counts the stashed gradients against the consumers that committed at forward time: one missing raises."""
    from sradsgan_amd import model as M, ops
    grp = M.ResGroup(M.RAB, n_blocks=1).to(DEV)
    x = torch.rand(1, 64, 12, 12, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = grp(x * 1.0)
    assert len(ops._state.carry_expect) >= 1
    tok = max(ops._state.carry_expect)
    ops._state.carry_expect[tok] += 1                 # a consumer that committed and never ran
    with pytest.raises(RuntimeError, match='gradients of the block input'):
        y.sum().backward()
    ops._state.carry.clear(); ops._state.carry_expect.clear()


def test_group_tail_hands_its_output_over_as_planes_bit_identical():
    """A ResGroup that feeds a ResGroup leaves its output also as padded planes (ops.attention_tail emit_pp -> srhip_conv2d_fwd_dual):
    the next group's first RAB and its weight gradient read them instead of a pp_from_f32 pass.  The planes hold exactly the split
    the pass would produce, so output and every gradient are bit-identical to the run without the hand-over -- and it happened."""
    from sradsgan_amd import model as M, ops
    if ops.get_conv_math() != 'bf16x3':
        pytest.skip('padded planes are a split-bf16 format')
    torch.manual_seed(5)
    net = M.GeneratorResNet(M.ResGroup, n_residual_blocks=3, n_basic_blocks=2, upscale_factor=2).to(DEV)
    x = torch.rand(2, 3, 24, 20, device=DEV)
    dy = torch.randn(2, 3, 48, 40, device=DEV)
    seen = []
    orig = ops.rab_block

    def spy(xx, *a, **k):
        tag = getattr(xx, '_srhip_pp', None)
        seen.append(tag is not None and tag[1] == xx._version)
        return orig(xx, *a, **k)

    def run(handover):
        old, ops._X_PP = ops._X_PP, handover
        try:
            for p in net.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            y = net(xi)
            y.backward(dy)
            return y.detach().clone(), xi.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}
        finally:
            ops._X_PP = old

    import sradsgan_amd.model.sradsgan as MS
    MS.ops.rab_block = spy
    try:
        y1, dx1, g1 = run(True)
    finally:
        MS.ops.rab_block = orig
    assert seen == [False, True, True, True, True, True], seen      # every RAB but the trunk's first finds its input as planes
    y0, dx0, g0 = run(False)
    assert torch.equal(y1, y0) and torch.equal(dx1, dx0)
    for k in g0:
        assert torch.equal(g1[k], g0[k]), k


@pytest.mark.parametrize('shape', [(2, 23, 37), (3, 27, 27), (1, 9, 20), (8, 54, 54), (1, 108, 108), (2, 5, 70)])
def test_tail_backward_with_the_7x7_data_gradient_inside_the_main_pass(shape):
    """Round 6: tail_bwd_main2_kernel computes dpooled = conv_transpose7x7(da, w7) itself from an LDS tile of da (16 lanes of a pixel
    split the 49 taps) instead of reading what slam_conv7_bwd_kernel wrote -- one launch less in every tail's serial chain (autograd of
    sradsgan.py:141-151 inside :254-274).  Against the round-5 launch sequence (srhip_debug_set(7, 32)), which the oracle tests pinned:
    output, skip gradient and the 7x7 weight gradient (same strips, same reduce order) bit-identical, every other gradient within 2e-5 of
    its largest magnitude (another summation order of 49 products).  Shapes: images narrower / wider than a block's pixel range, partial
    last groups, one group per block and sixteen."""
    from sradsgan_amd import _hip, ops
    n, h, w = shape
    g = torch.Generator().manual_seed(sum(shape) + 6)
    cl = lambda t: t.to(DEV).contiguous(memory_format=torch.channels_last)
    u0, skip0, dy = (cl(torch.randn(n, 64, h, w, generator=g)) for _ in range(3))
    par0 = [(torch.randn(4, 64, 1, 1, generator=g) * 0.2), (torch.randn(64, 4, 1, 1, generator=g) * 0.2), (torch.randn(1, 2, 7, 7, generator=g) * 0.1),
            (torch.randn(64, 64, 1, 1, generator=g) * 0.1), (torch.randn(64, generator=g) * 0.1)]
    res = []
    for old_sequence in (False, True):
        _hip.lib().srhip_debug_set(7, 32 if old_sequence else 0)
        try:
            u, skip = u0.clone().requires_grad_(True), skip0.clone().requires_grad_(True)
            par = [p.clone().to(DEV).requires_grad_(True) for p in par0]
            out = ops.attention_tail(u, skip, *par)
            out.backward(dy)
            torch.cuda.synchronize()
            res.append([out.detach().clone(), u.grad.clone(), skip.grad.clone()] + [p.grad.clone() for p in par])
        finally:
            _hip.lib().srhip_debug_set(7, 0)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2])
    assert torch.equal(res[0][5], res[1][5]), 'dw7'
    for a, b, name in zip(res[0][1:], res[1][1:], ['du', 'dskip', 'dfc1', 'dfc2', 'dw7', 'dwc', 'dbc']):
        err = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20)
        assert err < 2e-5, (name, err)


@pytest.mark.parametrize('case', [((2, 13, 17), (64, 64, 64)), ((32, 54, 54), (64, 64, 64)), ((1, 5, 3), (4, 128, 8, 36)), ((3, 9, 9), (64, 192))])
def test_cat_channels_is_torch_cat(case):
    """srhip_cat_channels / srhip_split_channels (ABI 11): torch.cat(dim=1) of NHWC tensors and its backward in one pass each -- the
    multi-scale block's concatenation (sradsgan.py:340-344).  Data movement only: output and gradients equal torch's bit for bit."""
    from sradsgan_amd import ops
    (n, h, w), chans = case
    g = torch.Generator().manual_seed(sum(chans) + n)
    xs = [torch.randn(n, c, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last) for c in chans]
    dy = torch.randn(n, sum(chans), h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    a = [x.clone().requires_grad_(True) for x in xs]
    b = [x.clone().requires_grad_(True) for x in xs]
    ya = ops.cat_channels(a)
    yb = torch.cat(b, dim=1)
    assert type(ya.grad_fn).__name__.startswith('_CatChannels')          # the HIP path ran
    assert torch.equal(ya, yb)
    ya.backward(dy)
    yb.backward(dy)
    for p, q in zip(a, b):
        assert torch.equal(p.grad, q.grad)


def test_weight_gradient_slots_leave_the_training_step_bit_identical():
    """Round 6 experiment (ops.release_ready_pair, SRHIP_WGRAD_SLOTS): complete pairs of RAB weight gradients start in slots behind
    conv1's data gradient instead of the moment they complete.  Same kernels, same accumulation per parameter: one full iteration
    (sradsgan.py:818-892) is bit-identical with the option on and off (3 groups x 2 RABs: pairs complete, wait and are flushed)."""
    from sradsgan_amd import ops
    from sradsgan_amd.train_step import TrainStep
    if ops.get_conv_math() != 'bf16x3':
        pytest.skip('the flat weight gradient is a split-bf16 path')
    B, side, scale = 4, 24, 4
    lr = O.det_fill('slots.lr', (B, 3, side, side), 0.5, 0.5).to(DEV)
    hr = O.det_fill('slots.hr', (B, 3, side * scale, side * scale), 0.5, 0.5).to(DEV)
    al = O.det_fill('slots.alpha', (B, 1, 1, 1), 0.5, 0.5).to(DEV)

    def run(on):
        old, ops._WGRAD_SLOTS = ops._WGRAD_SLOTS, on
        try:
            (hg, hd, hf), _ = build_pair(3, 2, scale, DEV)
            step = TrainStep(hg, hd, hf)
            out = step(lr, hr, al)
            torch.cuda.synchronize()
            grads = [p.grad.detach().clone() for p in list(hg.parameters()) + list(hd.parameters())]
            return torch.stack([out[k].double() for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')]).cpu(), grads
        finally:
            ops._WGRAD_SLOTS = old

    s1, g1 = run(True)
    s0, g0 = run(False)
    assert torch.isfinite(s1).all() and torch.equal(s1, s0)
    assert all(torch.equal(a, b) for a, b in zip(g1, g0))


def test_compact_record_shortcuts_leave_the_training_step_bit_identical():
    """Round 5 replaced several re-reads of large tensors by compact records or by the producing pass itself: the RAB's LeakyReLU mask
    as sign words (ops._PP_SIGNS), VGG's max-pool arg-max records (_POOL_IDX), the head activation's sign bits in the penalty's double
    backward (_LRELU_BITS), the BatchNorm backward's mask from the recomputed pre-activation (_BN_BWD_X) and a BatchNorm input's two
    gradients summed by the BatchNorm backward (_BN_FOLD).  None of them changes a single bit of one full training iteration
    (sradsgan.py:818-892): scalars and every generator / discriminator gradient equal with all of them off (the round-4 data flow)."""
    from sradsgan_amd import ops
    from sradsgan_amd.train_step import TrainStep
    if ops.get_conv_math() != 'bf16x3':
        pytest.skip('the plane-format shortcuts are split-bf16 paths')
    B, side, scale = 6, 27, 4                       # 6 x 64 x 108 x 108 = 4.5 M elements: the head activation is large enough for the sign-bit path
    lr = O.det_fill('rec.lr', (B, 3, side, side), 0.5, 0.5).to(DEV)
    hr = O.det_fill('rec.hr', (B, 3, side * scale, side * scale), 0.5, 0.5).to(DEV)
    al = O.det_fill('rec.alpha', (B, 1, 1, 1), 0.5, 0.5).to(DEV)
    knobs = ('_PP_SIGNS', '_POOL_IDX', '_LRELU_BITS', '_BN_BWD_X', '_BN_FOLD')
    assert all(getattr(ops, k) for k in knobs)

    def run(on):
        old = {k: getattr(ops, k) for k in knobs}
        for k in knobs:
            setattr(ops, k, on)
        try:
            (hg, hd, hf), _ = build_pair(2, 2, scale, DEV)
            step = TrainStep(hg, hd, hf)
            out = step(lr, hr, al)
            torch.cuda.synchronize()
            grads = [p.grad.detach().clone() for p in list(hg.parameters()) + list(hd.parameters())]
            return torch.stack([out[k].double() for k in ('loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp')]).cpu(), grads
        finally:
            for k, v in old.items():
                setattr(ops, k, v)

    s1, g1 = run(True)
    s0, g0 = run(False)
    assert torch.isfinite(s1).all() and torch.equal(s1, s0)
    assert all(torch.equal(a, b) for a, b in zip(g1, g0))
