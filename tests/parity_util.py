"""Shared by the GPU parity tests and __graft_entry__.smoke(): builds the HIP model and the CPU
oracle with identical deterministic weights and compares them."""
import numpy as np
import torch

from oracle import sradsgan_ref as O


def build_pair(n_groups, n_blocks, scale, device):
    """(G, D, F) on the HIP path and the oracle's (G, D, F) on CPU, same state_dicts."""
    from sradsgan_amd import model as M
    og = O.GeneratorResNet(O.ResGroup, n_residual_blocks=n_groups, n_basic_blocks=n_blocks, upscale_factor=scale)
    od, of = O.Discriminator(), O.FeatureExtractor()
    O.det_init_(og, prefix='G.'), O.det_init_(od, prefix='D.'), O.det_init_(of, prefix='F.')
    hg = M.GeneratorResNet(M.ResGroup, n_residual_blocks=n_groups, n_basic_blocks=n_blocks, upscale_factor=scale)
    hd, hf = M.Discriminator(), M.FeatureExtractor()
    hg.load_state_dict(og.state_dict(), strict=True)
    hd.load_state_dict(od.state_dict(), strict=True)
    hf.load_state_dict(of.state_dict(), strict=True)
    return (hg.to(device), hd.to(device), hf.to(device)), (og, od, of)


def rel_err(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def train_parity(device, tag, n_groups, n_blocks, batch, lr_side, scale, iters, golden=None):
    """Runs `iters` training iterations on both paths; returns max abs scalar diff and max rel
    weight diff.  With `golden` (npz from the reference) the HIP scalars are also checked against it."""
    from sradsgan_amd.train_step import TrainStep
    (hg, hd, hf), (og, od, of) = build_pair(n_groups, n_blocks, scale, device)
    step = TrainStep(hg, hd, hf)
    oG = torch.optim.Adam(og.parameters(), lr=2e-4, betas=(0.9, 0.999))
    oD = torch.optim.Adam(od.parameters(), lr=2e-4, betas=(0.9, 0.999))
    worst = 0.0
    names = ['loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'gp']
    for it in range(iters):
        lr_img = O.det_fill('%s.lr.%d' % (tag, it), (batch, 3, lr_side, lr_side), 0.5, 0.5)
        hr_img = O.det_fill('%s.hr.%d' % (tag, it), (batch, 3, lr_side * scale, lr_side * scale), 0.5, 0.5)
        if golden is not None:
            alpha = torch.from_numpy(golden['alpha%d' % it])
        else:
            alpha = O.det_fill('%s.alpha.%d' % (tag, it), (batch, 1, 1, 1), 0.5, 0.5)
        want = O.train_step(og, od, of, oG, oD, lr_img, hr_img, alpha)
        got = step(lr_img.to(device), hr_img.to(device), alpha.to(device))
        gv = np.array([float(got[k]) for k in names])
        wv = np.array([want[k] for k in names])
        worst = max(worst, float(np.abs(gv - wv).max()))
        if golden is not None:
            worst = max(worst, float(np.abs(gv - golden['scalars%d' % it]).max()))
    wdiff = 0.0
    for (k, a), (_, b) in zip(list(hg.state_dict().items()) + list(hd.state_dict().items()),
                              list(og.state_dict().items()) + list(od.state_dict().items())):
        # SGAM's key bias has a mathematically zero gradient (softmax shift invariance): Adam turns
        # its pure-roundoff gradient into +-lr steps on ANY two platforms, so it is not comparable.
        if a.dtype.is_floating_point and not k.endswith('key_conv.bias'):
            wdiff = max(wdiff, rel_err(a, b))
    return worst, wdiff
