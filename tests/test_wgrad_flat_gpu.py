"""The flat weight-gradient kernel on padded split-bf16 planes (csrc/conv_wgrad_flat.hip, ABI 9) against fp64: both tile shapes
(dy planes + x fp32 for Cout >= 128; dy fp32 + x planes for Cout = 64), ragged image sizes (the padded grid's row / image wraps
fall anywhere inside a 16-pixel chunk), channel counts that do not fill the last tile, grouped launches, the bias gradient, and
the plane format itself.  Replaces the autograd of sradsgan.py:222-223 like srhip_conv2d_wgrad."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _rel(got, ref):
    ref = ref.double()
    return float((got.detach().cpu().double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30))


def test_padded_planes_round_trip_and_zero_padding():
    from sradsgan_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 64, 5, 7, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    pp = ops.pp_from_f32(x)
    back = ops.pp_to_f32(pp)
    assert _rel(back, x.cpu()) < 2 ** -15                                   # hi + lo carries 16 significand bits
    planes = pp.buf.view(-1, 8, 2, 8).float()                               # [row][octet][hi | lo][8 channels]
    hi = planes[:, :, 0, :].reshape(-1, 64)
    lo = planes[:, :, 1, :].reshape(-1, 64)
    guard = __import__('sradsgan_amd')._hip.lib().srhip_pp_guard(7)
    grid = hi[guard:guard + 3 * 6 * 8].view(3, 6, 8, 64)
    assert float(grid[:, 5].abs().max()) == 0.0 and float(grid[:, :, 7].abs().max()) == 0.0    # pad row / pad column
    assert float(hi[:guard].abs().max()) == 0.0 and float(hi[guard + 3 * 6 * 8:].abs().max()) == 0.0
    assert float(lo[:guard].abs().max()) == 0.0 and float(lo[guard + 3 * 6 * 8:].abs().max()) == 0.0
    assert torch.equal(grid[:, :5, :7].permute(0, 3, 1, 2), x.to(torch.bfloat16).float())     # hi = round-to-nearest bf16
    # a buffer is reusable: converting another tensor of the same geometry leaves the padding untouched
    ops.pp_from_f32(-x, out=pp)
    hi2 = pp.buf.view(-1, 8, 2, 8)[:, :, 0, :].reshape(-1, 64).float()
    assert float(hi2[guard:guard + 144].view(3, 6, 8, 64)[:, 5].abs().max()) == 0.0


CASES = [(2, 64, 23, 37, 128), (1, 128, 54, 54, 64), (2, 256, 9, 20, 64), (2, 64, 17, 16, 256), (1, 64, 23, 23, 128),
         (3, 64, 23, 22, 128), (2, 128, 19, 40, 64), (1, 64, 2, 24, 256), (3, 128, 5, 17, 64), (2, 64, 8, 8, 576),
         (4, 64, 27, 27, 256), (2, 256, 24, 24, 64), (1, 64, 3, 5, 256), (1, 128, 1, 1, 64)]


def _operands(ops, mask, fmt, xg, dyg):
    """(x, dy) in the operand formats `fmt` ('dy', 'x' or 'both' = which of them are padded planes), None when not served."""
    want = {'dy': 1, 'x': 2, 'both': 4}[fmt]
    if not (mask & want):
        return None
    return (ops.pp_from_f32(xg) if fmt in ('x', 'both') else xg, ops.pp_from_f32(dyg) if fmt in ('dy', 'both') else dyg)


@pytest.mark.parametrize('case', CASES + [(2, 64, 23, 37, 256), (2, 256, 9, 20, 64), (3, 64, 5, 17, 576), (1, 512, 13, 11, 64), (2, 128, 12, 12, 320)])
def test_flat_wgrad_against_fp64(case):
    from sradsgan_amd import ops, _hip
    n, cin, h, w, cout = case
    g = torch.Generator().manual_seed(sum(case) + 11)
    x = torch.randn(n, cin, h, w, generator=g)
    dy = torch.randn(n, cout, h, w, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), padding=1)
    refb = dy.double().sum((0, 2, 3))
    xg = x.to(DEV).contiguous(memory_format=torch.channels_last)
    dyg = dy.to(DEV).contiguous(memory_format=torch.channels_last)
    with ops.conv_math('bf16x3'):
        mask = _hip.lib().srhip_conv2d_wgrad_pp_ok(n, h, w, cin, cout)
        assert mask & (1 if cout >= 128 else 2)
        assert bool(mask & 4) == ((cout >= 256 and cout % 8 == 0) or (cout == 64 and cin % 256 == 0))
        dw_rt, db_rt = ops.conv2d_wgrad_raw(xg, dyg, (cout, cin, 3, 3), 1, 1, True)
        for fmt in ('dy', 'x', 'both'):
            pair = _operands(ops, mask, fmt, xg, dyg)
            if pair is None:
                continue
            xo, dyo = pair
            dw = torch.full((cout, cin, 3, 3), 7.0, device=DEV)
            db = torch.full((cout,), -3.0, device=DEV)
            ops.conv2d_wgrad_pp_raw([(xo, dyo, dw, db)], accumulate=False)
            assert _rel(dw, ref) < 1.5e-5, fmt
            assert _rel(db, refb) < 5e-6, fmt
            # accumulate into the same buffers: twice the gradient
            ops.conv2d_wgrad_pp_raw([(xo, dyo, dw, db)], accumulate=True)
            assert _rel(dw, 2 * ref) < 1.5e-5 and _rel(db, 2 * refb) < 5e-6, fmt
            # and against the row-tap kernel on the same operands (different summation order, same products)
            assert _rel(dw, 2 * dw_rt.cpu().double()) < 5e-6, fmt


@pytest.mark.parametrize('case', [(2, 64, 23, 37, 128, 2, 'dy'), (3, 128, 19, 40, 64, 2, 'x'), (32, 64, 54, 54, 256, 2, 'dy'), (32, 256, 54, 54, 64, 2, 'x'),
                                  (4, 64, 27, 27, 256, 4, 'dy'), (2, 256, 24, 24, 64, 3, 'x'), (32, 64, 54, 54, 256, 2, 'both'),
                                  (32, 256, 54, 54, 64, 2, 'both'), (4, 64, 27, 27, 256, 4, 'both'), (2, 256, 24, 24, 64, 3, 'both'),
                                  (3, 64, 11, 9, 576, 2, 'both'), (32, 64, 54, 54, 256, 3, 'both'), (32, 256, 54, 54, 64, 3, 'both')])     # (the last two: the step's launches since round 6)
def test_grouped_flat_wgrad_matches_the_single_launches(case):
    """nprob weight gradients of one shape behind one launch (every problem with nsplit / nprob splits): the same products summed
    over fewer, longer splits -- equal to the single launches up to the summation order, with and without a bias gradient."""
    from sradsgan_amd import ops, _hip
    n, cin, h, w, cout, k, fmt = case
    g = torch.Generator().manual_seed(sum(case[:6]) + 17)
    xs = [torch.randn(n, cin, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last) for _ in range(k)]
    dys = [torch.randn(n, cout, h, w, generator=g).to(DEV).contiguous(memory_format=torch.channels_last) for _ in range(k)]
    with ops.conv_math('bf16x3'):
        mask = _hip.lib().srhip_conv2d_wgrad_pp_ok(n, h, w, cin, cout)
        ops_ = [_operands(ops, mask, fmt, x, dy) for x, dy in zip(xs, dys)]
        single = []
        for xo, dyo in ops_:
            dw, db = torch.zeros(cout, cin, 3, 3, device=DEV), torch.zeros(cout, device=DEV)
            ops.conv2d_wgrad_pp_raw([(xo, dyo, dw, db)], accumulate=False)
            single.append((dw, db))
        dws = [torch.zeros(cout, cin, 3, 3, device=DEV) for _ in range(k)]
        dbs = [torch.zeros(cout, device=DEV) if i != 1 else None for i in range(k)]      # one problem without a bias gradient
        ops.conv2d_wgrad_pp_raw([(xo, dyo, dws[i], dbs[i]) for i, (xo, dyo) in enumerate(ops_)], accumulate=False)
    for i in range(k):
        assert _rel(dws[i], single[i][0].cpu()) < 5e-6
        if dbs[i] is not None:
            assert _rel(dbs[i], single[i][1].cpu()) < 5e-6
    if n >= 32:                                                               # the bench shape also against fp64
        ref = torch.nn.grad.conv2d_weight(xs[0].cpu().double(), (cout, cin, 3, 3), dys[0].cpu().double(), padding=1)
        assert _rel(dws[0], ref) < 1.5e-5
