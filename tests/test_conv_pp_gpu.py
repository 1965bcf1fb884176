"""The persistent patch kernel with padded split-bf16 plane operands (csrc/conv_patch_pers.hip SRCPP / DSTPP, ABI 9) against the same
kernel family on fp32 tensors: a pp source carries exactly the hi | lo split the kernel forms itself, so forward / data-gradient
results are BIT-IDENTICAL to the fp32-operand calls; a pp destination holds the split of the fp32 result (hi bit-exact, hi + lo
within 2^-16).  Shapes: the RAB's four launches (conv1 fprop 64 -> 256 bias + LeakyReLU -> planes; conv2 fprop planes -> 64
(+ pooling partials); conv2 dgrad 64 -> 256 with the activation mask -> planes; conv1 dgrad planes -> 64 + residual) on ragged
images, several tiles per block, partial edge tiles.  Replaces sradsgan.py:222-223, 250-252 like srhip_conv2d_fwd / _dgrad."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _cl(t):
    return t.to(DEV).contiguous(memory_format=torch.channels_last)


CASES = [(2, 23, 37), (1, 54, 54), (3, 9, 20), (2, 17, 16), (5, 27, 27), (2, 24, 24), (16, 54, 54)]


@pytest.mark.parametrize('case', CASES)
def test_rab_convs_on_padded_planes_match_the_fp32_operand_kernels(case):
    from sradsgan_amd import ops, _hip
    n, h, w = case
    g = torch.Generator().manual_seed(sum(case) + 5)
    x = _cl(torch.randn(n, 64, h, w, generator=g))
    w1 = torch.nn.Parameter((torch.randn(256, 64, 3, 3, generator=g) * 0.05).to(DEV))
    b1 = (torch.randn(256, generator=g) * 0.1).to(DEV)
    w2 = torch.nn.Parameter((torch.randn(64, 256, 3, 3, generator=g) * 0.05).to(DEV))
    b2 = (torch.randn(64, generator=g) * 0.1).to(DEV)
    du = _cl(torch.randn(n, 64, h, w, generator=g))
    gres = _cl(torch.randn(n, 64, h, w, generator=g))
    lib = _hip.lib()
    with ops.conv_math('bf16x3'):
        assert ops.conv2d_pp_ok(n, 64, h, w, 256) and ops.conv2d_pp_ok(n, 256, h, w, 64)
        for grid in (0, 5):                                  # 5: several tiles per block on small images (srhip_debug_set(5, n))
            lib.srhip_debug_set(5, grid)
            lib.srhip_debug_set(0, -2)                       # the fp32-tensor reference through the patch kernels at any size (small
            try:                                             # problems otherwise take the exact-fp32 register-staged kernel)
                # reference: the fp32-tensor path
                t = ops.conv2d_fwd_raw(x, w1, b1, 1, 1, 0.2)
                u = ops.conv2d_fwd_raw(t, w2, b2, 1, 1)
                dt = ops.conv2d_dgrad_raw(du, w2, tuple(t.shape), 1, 1, None, t, 0.2)
                dx = ops.conv2d_dgrad_raw(dt, w1, tuple(x.shape), 1, 1, gres)
                # planes
                t_pp = ops.conv2d_fwd_pp_raw(x, w1, b1, 0.2, out_pp=ops.pp_empty(n, 256, h, w, DEV))
                t_ref = ops.pp_from_f32(t)
                assert torch.equal(t_pp.buf, t_ref.buf)                        # the split of the fp32 result, pads untouched (zero)
                u_pp = ops.conv2d_fwd_pp_raw(t_pp, w2, b2)
                assert torch.equal(u_pp, u)
                u_pool, (pool, sec, nseg) = ops.conv2d_fwd_pp_raw(t_pp, w2, b2, pool=True)
                assert torch.equal(u_pool, u) and nseg > 0
                if ops.pool_epilogue_ok(t, w2):
                    u2, (pool2, sec2, nseg2) = ops.conv2d_fwd_pool_raw(t, w2, b2)
                    assert nseg2 == nseg and sec2 == sec
                    for k in range(3):
                        a = pool[k * sec // 4:k * sec // 4 + n * nseg * 64]
                        b = pool2[k * sec // 4:k * sec // 4 + n * nseg * 64]
                        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
                dt_pp = ops.conv2d_dgrad_pp_raw(du, w2, actmask=t_pp, slope=0.2, out_pp=ops.pp_empty(n, 256, h, w, DEV))
                assert torch.equal(dt_pp.buf, ops.pp_from_f32(dt).buf)
                dx_pp = ops.conv2d_dgrad_pp_raw(dt_pp, w1, residual=gres)
                assert torch.equal(dx_pp, dx)
                # the LeakyReLU mask as sign words (srhip_conv2d_fwd_pp_signs / _dgrad_pp_signs): the forward leaves 1 bit per element, the
                # masked data gradient reads those instead of t's hi plane -- same planes out, from fp32 and from plane sources
                x_pl, du_pl = ops.pp_from_f32(x), ops.pp_from_f32(du)
                words = []
                for xs, dus in ((x, du), (x_pl, du_pl)):
                    signs = ops.pp_sign_words(n, h, w, 256, DEV)
                    assert signs is not None and signs.numel() * 8 == lib.srhip_conv2d_pp_sign_bytes(n, h, w, 256)
                    signs.fill_(0x5555555555555555)
                    t_s = ops.conv2d_fwd_pp_raw(xs, w1, b1, 0.2, out_pp=ops.pp_empty(n, 256, h, w, DEV), signs=signs)
                    assert torch.equal(t_s.buf, t_ref.buf)
                    dt_s = ops.conv2d_dgrad_pp_raw(dus, w2, slope=0.2, out_pp=ops.pp_empty(n, 256, h, w, DEV), signs=signs)
                    assert torch.equal(dt_s.buf, dt_pp.buf)
                    words.append(signs)
                assert torch.equal(words[0], words[1])
                with pytest.raises(RuntimeError):                              # a buffer smaller than the walk's tiles need
                    ops.conv2d_fwd_pp_raw(x, w1, b1, 0.2, out_pp=ops.pp_empty(n, 256, h, w, DEV), signs=words[0][:-2])
            finally:
                lib.srhip_debug_set(5, 0)
                lib.srhip_debug_set(0, 0)
        # a reused buffer keeps its zero padding: write other data into the same planes, pads still zero
        ops.conv2d_fwd_pp_raw(-x, w1, b1, 0.2, out_pp=t_pp)
        guard = lib.srhip_pp_guard(w)
        grid_ = t_pp.buf[guard:guard + n * (h + 1) * (w + 1)].view(n, h + 1, w + 1, 512).float()
        assert float(grid_[:, h].abs().max()) == 0.0 and float(grid_[:, :, w].abs().max()) == 0.0
        assert float(t_pp.buf[:guard].float().abs().max()) == 0.0


def test_tiny_image_on_padded_planes_against_fp64():
    """3 x 5 pixels: the fp32-tensor path has no patch kernel at this size (exact-fp32 register-staged kernel), the padded-plane path
    takes the patch kernel with a 12 %-filled tile: compared with fp64 instead of bit for bit."""
    from sradsgan_amd import ops
    n, h, w = 1, 3, 5
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n, 64, h, w, generator=g)
    w1 = (torch.randn(256, 64, 3, 3, generator=g) * 0.05)
    b1 = torch.randn(256, generator=g) * 0.1
    w2 = (torch.randn(64, 256, 3, 3, generator=g) * 0.05)
    du = torch.randn(n, 64, h, w, generator=g)
    F = torch.nn.functional
    t64 = F.leaky_relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1), 0.2)
    u64 = F.conv2d(t64, w2.double(), None, padding=1)
    dt64 = F.conv_transpose2d(du.double(), w2.double(), padding=1) * torch.where(t64 > 0, 1.0, 0.2)
    dx64 = F.conv_transpose2d(dt64, w1.double(), padding=1)
    rel = lambda a, b: float((a.cpu().double() - b).abs().max() / b.abs().max())
    with ops.conv_math('bf16x3'):
        w1p, w2p = torch.nn.Parameter(w1.to(DEV)), torch.nn.Parameter(w2.to(DEV))
        t_pp = ops.conv2d_fwd_pp_raw(_cl(x), w1p, b1.to(DEV), 0.2, out_pp=ops.pp_empty(n, 256, h, w, DEV))
        assert rel(ops.pp_to_f32(t_pp), t64) < 2e-5
        assert rel(ops.conv2d_fwd_pp_raw(t_pp, w2p, None), u64) < 2e-5
        dt_pp = ops.conv2d_dgrad_pp_raw(_cl(du), w2p, actmask=t_pp, slope=0.2, out_pp=ops.pp_empty(n, 256, h, w, DEV))
        assert rel(ops.pp_to_f32(dt_pp), dt64) < 2e-5
        assert rel(ops.conv2d_dgrad_pp_raw(dt_pp, w1p), dx64) < 3e-5


@pytest.mark.parametrize('case', [(32, 64, 54, 54, 256), (16, 64, 54, 54, 256), (12, 64, 54, 54, 256), (16, 64, 40, 37, 256), (8, 64, 54, 54, 576),
                                  (12, 128, 54, 54, 256), (4, 64, 108, 108, 256), (16, 32, 54, 54, 320)])
def test_eight_wave_patch_kernel_is_bit_identical_to_the_four_wave_kernels(case):
    """conv_patch8_kernel (csrc/conv_patch8.hip: one 8-wave block per CU, two wave groups one barrier apart, >= 256 destination
    channels, >= one tile per CU) against conv_patch_pers_kernel (srhip_debug_set(15, 0)) on the same operands: forward with
    bias + LeakyReLU and the data gradient with an activation mask, to fp32 tensors and to padded planes, several tiles per block,
    ragged images, a channel count that does not fill the last 256-wide tile, 2 / 4 / 8 chunks."""
    from sradsgan_amd import ops, _hip
    n, cin, h, w, cout = case
    lib = _hip.lib()
    g = torch.Generator().manual_seed(sum(case) + 23)
    x = _cl(torch.randn(n, cin, h, w, generator=g))
    w1 = torch.nn.Parameter((torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV))
    b1 = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    w2 = torch.nn.Parameter((torch.randn(cin, cout, 3, 3, generator=g) * 0.05).to(DEV))     # dgrad: cin -> cout channels
    du = _cl(torch.randn(n, cin, h, w, generator=g))
    with ops.conv_math('bf16x3'):
        res = {}
        for k8 in (0, 1):
            lib.srhip_debug_set(15, k8)
            try:
                t = ops.conv2d_fwd_raw(x, w1, b1, 1, 1, 0.2)
                dt = ops.conv2d_dgrad_raw(du, w2, tuple(t.shape), 1, 1, None, t, 0.2)
                out = [t, dt]
                if ops.conv2d_pp_ok(n, cin, h, w, cout):
                    t_pp = ops.conv2d_fwd_pp_raw(x, w1, b1, 0.2, out_pp=ops.pp_empty(n, cout, h, w, DEV))
                    dt_pp = ops.conv2d_dgrad_pp_raw(du, w2, actmask=t_pp, slope=0.2, out_pp=ops.pp_empty(n, cout, h, w, DEV))
                    out += [t_pp.buf, dt_pp.buf]
                res[k8] = out
            finally:
                lib.srhip_debug_set(15, 0)
        for a, b in zip(res[0], res[1]):
            assert torch.equal(a, b)


@pytest.mark.parametrize('case', [(2, 23, 37), (32, 54, 54), (3, 9, 20)])
def test_tail_conv_leaves_its_output_also_as_padded_planes(case):
    """srhip_conv2d_fwd_dual: the attention tail's 1x1 conv (bias + residual + row / channel scales, the row-group epilogue of
    conv_fast.hip) writes y as fp32 AND as padded planes = exactly pp_from_f32(y); a conv whose kernel has no second destination
    (3x3 patch kernel) falls back to the conversion pass behind the same call."""
    from sradsgan_amd import ops
    n, h, w = case
    g = torch.Generator().manual_seed(sum(case) + 31)
    u, skip = _cl(torch.randn(n, 64, h, w, generator=g)), _cl(torch.randn(n, 64, h, w, generator=g))
    wc = torch.nn.Parameter((torch.randn(64, 64, 1, 1, generator=g) * 0.1).to(DEV))
    bc = torch.randn(64, generator=g).to(DEV)
    m, s = torch.rand(n * h * w, generator=g).to(DEV), torch.rand(n, 64, generator=g).to(DEV)
    with ops.conv_math('bf16x3'):
        y0 = ops.conv2d_fwd_raw(u, wc, bc, 1, 0, None, skip, m, s)
        pp = ops.pp_empty(n, 64, h, w, DEV)
        y1 = ops.conv2d_fwd_raw(u, wc, bc, 1, 0, None, skip, m, s, out_pp=pp)
        assert torch.equal(y0, y1) and torch.equal(pp.buf, ops.pp_from_f32(y0).buf)
        w3 = torch.nn.Parameter((torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(DEV))
        pp3 = ops.pp_empty(n, 64, h, w, DEV)
        y3 = ops.conv2d_fwd_raw(u, w3, bc, 1, 1, out_pp=pp3)
        assert torch.equal(pp3.buf, ops.pp_from_f32(y3).buf)


def test_rab_chain_with_handed_over_input_planes_is_bit_identical():
    """Three RABs in a row at the bench's tile size: with the block outputs handed over as padded planes (the tail conv's second
    destination, conv1 and its weight gradient reading them) the output, the input gradient and every parameter gradient equal the
    chain in which every block converts for itself (SRHIP_X_PP=0 behaviour)."""
    from sradsgan_amd import ops, model as M
    torch.manual_seed(5)
    blocks = torch.nn.ModuleList([M.RAB(64, 64) for _ in range(3)]).to(DEV)
    x0 = _cl(torch.randn(8, 64, 54, 54, device=DEV))
    res = []
    with ops.conv_math('bf16x3'):
        for hand_over in (False, True):
            for p in blocks.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(True)
            out = x
            for i, b in enumerate(blocks):
                b._next_is_rab = hand_over and i < 2
                out = b(out)
            out.square().mean().backward()
            torch.cuda.synchronize()
            res.append([out.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in blocks.parameters()])
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize('case', [(32, 54, 54), (16, 54, 54), (2, 23, 37), (1, 9, 20)])
@pytest.mark.parametrize('nextra', [1, 2])
def test_data_gradient_with_three_residuals_equals_the_chained_adds(case, nextra):
    """srhip_conv2d_dgrad_res3 / _pp_res3 (ABI 9): dx = dgrad + residual + residual2 + residual3 in that order -- the persistent patch
    kernel's epilogue at the bench shapes, one srhip_sum_n pass where another kernel serves the launch -- against the plain call
    followed by torch adds in the same order: bit-identical (each add is one fp32 rounding either way)."""
    from sradsgan_amd import ops
    n, h, w = case
    g = torch.Generator().manual_seed(sum(case) + 77 + nextra)
    dt = _cl(torch.randn(n, 256, h, w, generator=g))
    w1 = torch.nn.Parameter((torch.randn(256, 64, 3, 3, generator=g) * 0.05).to(DEV))
    res = _cl(torch.randn(n, 64, h, w, generator=g))
    extra = [_cl(torch.randn(n, 64, h, w, generator=g)) for _ in range(nextra)]
    with ops.conv_math('bf16x3'):
        want = ops.conv2d_dgrad_raw(dt, w1, (n, 64, h, w), 1, 1, res)
        for e in extra:
            want = want + e
        got = ops.conv2d_dgrad_raw(dt, w1, (n, 64, h, w), 1, 1, res, extra=extra)
        assert torch.equal(got, want)
        dt_pp = ops.pp_from_f32(dt)
        want_pp = ops.conv2d_dgrad_pp_raw(dt_pp, w1, residual=res)
        for e in extra:
            want_pp = want_pp + e
        got_pp = ops.conv2d_dgrad_pp_raw(dt_pp, w1, residual=res, extra=extra)
        assert torch.equal(got_pp, want_pp)
    with ops.conv_math('fp32'):                          # another kernel family: the library adds the extras itself
        want = ops.conv2d_dgrad_raw(dt, w1, (n, 64, h, w), 1, 1, res)
        for e in extra:
            want = want + e
        assert torch.equal(ops.conv2d_dgrad_raw(dt, w1, (n, 64, h, w), 1, 1, res, extra=extra), want)
