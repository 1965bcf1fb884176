"""Global attention (CGAM, flash-style SGAM) and loss-reduction kernels through the C ABI against fp64 torch
evaluations of the reference formulas (SRADSGAN/model/sradsgan.py:153-213, :630-637, :686), at the sizes the
kernels tile raggedly: one partial tile (6x7), several tiles with a tail (10x12, 54x54 = BASELINE x4) and the x2
tile (108x108, N = 11664, where the reference materialises two 544 MB matrices per image)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _err(got, want):
    want = want.double().cpu()
    return float((got.double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-20))


def _cgam_ref(x, gamma):
    b, c, h, w = x.shape
    xf = x.reshape(b, c, h * w)
    energy = xf @ xf.transpose(1, 2)
    energy = energy.max(dim=-1, keepdim=True)[0] - energy
    return gamma * (torch.softmax(energy, dim=-1) @ xf).reshape(b, c, h, w) + x


def _sgam_ref(x, q, k, v, gamma):
    b, c, h, w = x.shape
    qf, kf, vf = q.reshape(b, -1, h * w), k.reshape(b, -1, h * w), v.reshape(b, -1, h * w)
    att = torch.softmax(qf.transpose(1, 2) @ kf, dim=-1)
    return gamma * (vf @ att.transpose(1, 2)).reshape(b, c, h, w) + x


@pytest.mark.parametrize('b,h,w', [(1, 6, 7), (2, 10, 12), (2, 54, 54), (1, 27, 27)])
def test_cgam_kernels_against_fp64(b, h, w):
    from sradsgan_amd import ops
    x = _rand((b, 64, h, w), 1, 0.3)
    dy = _rand((b, 64, h, w), 2)
    gamma = torch.tensor([0.7])
    xr, gr = x.double().requires_grad_(True), gamma.double().requires_grad_(True)
    yr = _cgam_ref(xr, gr)
    yr.backward(dy.double())
    xh, gh = x.to(DEV).requires_grad_(True), gamma.to(DEV).requires_grad_(True)
    yh = ops.cgam(xh, gh)
    yh.backward(dy.to(DEV))
    assert _err(yh.detach(), yr.detach()) < 2e-5
    assert _err(xh.grad, xr.grad) < 1e-4
    assert _err(gh.grad, gr.grad) < 1e-4


@pytest.mark.parametrize('exact', [False, True])
@pytest.mark.parametrize('b,h,w', [(1, 6, 7), (2, 10, 12), (1, 5, 32), (2, 54, 54), (1, 27, 27)])
def test_sgam_flash_kernels_against_fp64(b, h, w, exact):
    """exact=False: the default kernels (split-bf16 products for P.V / dP / dV / dQ / dK, fp32 energies and soft-max);
    exact=True: the all-fp32-MFMA kernels (srhip_debug_set(4, 1); what SRHIP_MATH_FP32 runs).  Same bars."""
    from sradsgan_amd import _hip, ops
    _hip.lib().srhip_debug_set(4, 1 if exact else 0)
    try:
        _sgam_case(ops, b, h, w)
    finally:
        _hip.lib().srhip_debug_set(4, 0)


def _sgam_case(ops, b, h, w):
    x, q, k = _rand((b, 64, h, w), 3), _rand((b, 8, h, w), 4, 1.5), _rand((b, 8, h, w), 5, 1.5)
    v, dy = _rand((b, 64, h, w), 6), _rand((b, 64, h, w), 7)
    gamma = torch.tensor([0.6])
    ref = [t.double().requires_grad_(True) for t in (x, q, k, v, gamma)]
    yr = _sgam_ref(*ref)
    yr.backward(dy.double())
    hip = [t.to(DEV).requires_grad_(True) for t in (x, q, k, v, gamma)]
    yh = ops.sgam(*hip)
    yh.backward(dy.to(DEV))
    errs = [_err(yh.detach(), yr.detach())] + [_err(a.grad, r.grad) for a, r in zip(hip, ref)]
    print('sgam %dx%dx%d: y %.2e  dx %.2e dq %.2e dk %.2e dv %.2e dgamma %.2e' % ((b, h, w) + tuple(errs)))
    assert errs[0] < 2e-5
    for name, e in zip(('dx', 'dq', 'dk', 'dv', 'dgamma'), errs[1:]):
        assert e < 2e-4, (name, e)


def test_sgam_flash_forces_the_online_softmax_rescale():
    """A key that dominates one query's row only in a LATE tile makes the running maximum jump there: the online
    rescale of the accumulator must fire (cdna guide rule 26: a data-dependent branch needs an input that forces it)."""
    from sradsgan_amd import ops
    b, h, w = 1, 10, 12                                       # 120 keys = 4 tiles
    x, q, k = _rand((b, 64, h, w), 13), _rand((b, 8, h, w), 14, 0.5), _rand((b, 8, h, w), 15, 0.5)
    v = _rand((b, 64, h, w), 16)
    q[0, :, 0, 3] = 4.0                                        # query 3 ...
    k[0, :, 9, 5] = 4.0                                        # ... against key 113 (last tile): score 128, everything else O(1)
    k[0, :, 4, 0] = 2.5                                        # and a smaller spike at key 48 (tile 1): score 80
    gamma = torch.tensor([1.0])
    yr = _sgam_ref(x.double(), q.double(), k.double(), v.double(), gamma.double())
    yh = ops.sgam(x.to(DEV), q.to(DEV), k.to(DEV), v.to(DEV), gamma.to(DEV))
    assert _err(yh, yr) < 2e-5


def test_sgam_flash_x2_tile_never_materialises_n_squared():
    """x2's real tile (LR 108 x 108, N = 11664): forward + backward of the attention core at B = 2 stay far below the
    2 x 544 MB per image the reference's bmm + softmax allocate, and match a float64 evaluation on one image."""
    from sradsgan_amd import ops
    b, h, w = 2, 108, 108
    x, q, k = _rand((b, 64, h, w), 23), _rand((b, 8, h, w), 24), _rand((b, 8, h, w), 25)
    v, dy = _rand((b, 64, h, w), 26), _rand((b, 64, h, w), 27)
    gamma = torch.tensor([0.5])
    hip = [t.to(DEV).requires_grad_(True) for t in (x, q, k, v, gamma)]
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    yh = ops.sgam(*hip)
    yh.backward(dy.to(DEV))
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    n2 = (h * w) ** 2 * 4
    print('sgam x2 tile: peak extra memory %.1f MB (one N x N fp32 matrix would be %.1f MB per image)' % (peak / 2 ** 20, n2 / 2 ** 20))
    assert peak < 0.25 * n2
    ref = [t[:1].double().to(DEV).requires_grad_(True) if t.dim() == 4 else t.double().to(DEV).requires_grad_(True)
           for t in (x, q, k, v, gamma)]
    yr = _sgam_ref(*ref)
    yr.backward(dy[:1].double().to(DEV))
    assert _err(yh.detach()[:1], yr.detach()) < 2e-5
    for name, a, r in zip(('dx', 'dq', 'dk', 'dv'), hip, ref):
        assert _err(a.grad[:1], r.grad) < 2e-4, name


def test_loss_reduction_kernels_against_torch():
    from sradsgan_amd import ops
    a, bt = _rand((3, 3, 40, 52), 31), _rand((3, 3, 40, 52), 32)
    bt[0, 0, :4] = a[0, 0, :4]                                # exact ties: sign(0) = 0 like torch
    ar = a.double().requires_grad_(True)
    lr_ = (ar - bt.double()).abs().mean()
    (3.0 * lr_).backward()
    ah = a.to(DEV).requires_grad_(True)
    lh = ops.l1_mean(ah, bt.to(DEV))
    (3.0 * lh).backward()
    assert abs(float(lh.detach()) - float(lr_.detach())) < 1e-6 and _err(ah.grad, ar.grad) < 1e-6
    # odd element count (tail path) and 2-d inputs
    c, d = _rand((7, 33), 33), _rand((7, 33), 34)
    assert abs(float(ops.l1_mean(c.to(DEV), d.to(DEV))) - float((c.double() - d.double()).abs().mean())) < 1e-6
    # critic mean
    s = _rand((5, 1, 14, 14), 35)
    sh = s.to(DEV).requires_grad_(True)
    m = ops.mean(sh)
    (-m).backward()
    assert abs(float(m) - float(s.double().mean())) < 1e-6
    assert torch.allclose(sh.grad.cpu(), torch.full_like(s, -1.0 / s.numel()))
    # gradient-penalty reduction incl. a zero-norm pixel
    g = _rand((2, 3, 24, 20), 36, 1.5)
    g[0, :, 0, 0] = 0.0
    gr = g.double().requires_grad_(True)
    pr = (gr.norm(2, 1) - 1).pow(2).mean()
    (11.0 * pr).backward()
    gh = g.to(DEV).requires_grad_(True)
    ph = ops.gp_penalty(gh)
    (11.0 * ph).backward()
    assert abs(float(ph) - float(pr)) < 1e-6 * max(1.0, float(pr))
    assert _err(gh.grad, gr.grad) < 1e-5
    assert float(gh.grad[0, :, 0, 0].abs().max()) == 0.0


def _cbam_ref(x, fc1, fc2, w7):
    """ChannelAttention -> SpatialAttention of the discriminator (base_networks.py:387-403, 440-457) in plain torch."""
    import torch.nn.functional as F
    mlp = lambda v: F.conv2d(F.relu(F.conv2d(v, fc1)), fc2)
    s = torch.sigmoid(mlp(F.adaptive_avg_pool2d(x, 1)) + mlp(F.adaptive_max_pool2d(x, 1)))
    y = s * x
    pooled = torch.cat([y.mean(dim=1, keepdim=True), y.max(dim=1, keepdim=True)[0]], dim=1)
    return torch.sigmoid(F.conv2d(pooled, w7, padding=3)) * y


@pytest.mark.parametrize('b,c,h,w', [(2, 256, 27, 27), (3, 64, 10, 12)])
def test_discriminator_attention_pair_first_and_second_order(b, c, h, w):
    """The srhip_cbam_* composition at the discriminator's shape (C = 256 @ 27 x 27): forward, first-order gradients and the
    gradient-penalty pattern (a function of d out / d x differentiated again w.r.t. x and the weights) against float64."""
    from sradsgan_amd import ops
    x, dy, r = _rand((b, c, h, w), 41), _rand((b, c, h, w), 42), _rand((b, c, h, w), 43)
    fc1, fc2, w7 = _rand((c // 16, c, 1, 1), 44, 0.3), _rand((c, c // 16, 1, 1), 45, 0.3), _rand((1, 2, 7, 7), 46, 0.3)

    def run(dev, dtype):
        xs, f1, f2, k7 = (t.to(dev, dtype).requires_grad_(True) for t in (x, fc1, fc2, w7))
        if dev == 'cpu':
            out = _cbam_ref(xs, f1, f2, k7)
        else:
            out = ops.slam(ops.clam(xs, f1, f2), k7)
        (g1,) = torch.autograd.grad(out, xs, dy.to(dev, dtype), retain_graph=True)
        (gx,) = torch.autograd.grad(out, xs, dy.to(dev, dtype), create_graph=True)
        pen = (gx * r.to(dev, dtype)).sum() + 0.1 * (gx * gx).sum()
        g2 = torch.autograd.grad(pen, [xs, f1, f2, k7])
        return [out.detach(), g1, gx.detach()] + list(g2)
    ref = run('cpu', torch.float64)
    got = run(DEV, torch.float32)
    for name, a_, r_ in zip(('out', 'dx', 'dx(graph)', 'pen/dx', 'pen/dfc1', 'pen/dfc2', 'pen/dw7'), got, ref):
        assert _err(a_, r_) < 5e-4, (name, _err(a_, r_))
