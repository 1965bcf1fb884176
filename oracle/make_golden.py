"""Generate tests/golden/*.npz by running the REFERENCE modules (run in the build container only).

    python oracle/make_golden.py            # needs /root/reference; never runs on the GPU box
    python oracle/make_golden.py gabup_x8 gen_small_x8 ...   # only the named fixtures (the others stay as committed)

The reference (Meng-333/SRADSGAN) ships no tests or golden vectors, so the
vectors that pin the oracle are produced here by importing reference/SRADSGAN/model/sradsgan.py
itself (third-party modules that are not installed are stubbed), filling its
parameters with the deterministic hash filler of oracle/sradsgan_ref.py (`det_init_`, keyed by
state_dict name) and recording outputs.  Only data (inputs are re-derivable, outputs stored) is
written; no reference source text is copied.
"""
import importlib
import importlib.machinery as mach
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
from oracle import sradsgan_ref as O  # noqa: E402

REF = '/root/reference/SRADSGAN'


def import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)

    class _Any:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return _Any()

        def __getattr__(self, k):
            return _Any()

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        m.__spec__ = mach.ModuleSpec(name, None)
        sys.modules[name] = m
        return m

    tv = stub('torchvision')
    names = ['ToPILImage', 'ToTensor', 'Compose', 'CenterCrop', 'Resize', 'Normalize', 'RandomCrop',
             'RandomHorizontalFlip']
    tr = stub('torchvision.transforms', **{n: _Any for n in names})
    tr.__all__ = names
    tr.functional = stub('torchvision.transforms.functional')
    tv.transforms = tr
    tv.utils = stub('torchvision.utils', save_image=_Any, make_grid=_Any)
    tv.datasets = stub('torchvision.datasets')
    tv.models = stub('torchvision.models', vgg19=_Any)
    stub('skimage')
    stub('skimage.measure', compare_ssim=_Any, compare_mse=_Any, compare_psnr=_Any, compare_nrmse=_Any)
    for n in ('tensorflow', 'cv2', 'sewar', 'imageio', 'thop', 'scipy.misc'):
        stub(n)
    return importlib.import_module('model.sradsgan')


def vgg_standin():
    """vgg19.features[:12] layer structure with plain torch.nn (torchvision is absent); key names
    under `feature_extractor.` as in the reference FeatureExtractor."""
    import torch.nn as nn

    class FE(nn.Module):
        def __init__(self):
            super().__init__()
            self.feature_extractor = nn.Sequential(
                nn.Conv2d(3, 64, 3, 1, 1), nn.ReLU(True), nn.Conv2d(64, 64, 3, 1, 1), nn.ReLU(True),
                nn.MaxPool2d(2, 2), nn.Conv2d(64, 128, 3, 1, 1), nn.ReLU(True), nn.Conv2d(128, 128, 3, 1, 1),
                nn.ReLU(True), nn.MaxPool2d(2, 2), nn.Conv2d(128, 256, 3, 1, 1), nn.ReLU(True))

        def forward(self, x):
            return self.feature_extractor(x)
    return FE()


def np32(t):
    return O.digest(t)


ONLY = set(sys.argv[1:])                    # fixture names to (re)write; empty = all of them


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('%-28s %7.1f KB' % (name, os.path.getsize(path) / 1024.0))


def grads_of(module, keys):
    sd = dict(module.named_parameters())
    return {('grad__' + k.replace('.', '__')): np32(sd[k].grad) for k in keys}


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = import_reference()
    from utils import utils as srutils  # reference utils (stubs make it importable)

    # ---- 1. pixel-shuffle exact index tables (sradsgan.py:382,385) -------------------------- #
    tabs = {}
    for r in (2, 3):
        c, h, w = 2, 3, 4
        src = torch.arange(c * r * r * h * w, dtype=torch.float32).reshape(1, c * r * r, h, w)
        tabs['r%d' % r] = torch.nn.PixelShuffle(r)(src).to(torch.int64).numpy()
    save('pixel_shuffle_index', **tabs)

    # ---- 2. per-module forward + input-grad + a few param grads ------------------------------ #
    def run_module(tag, mod, x, grad_keys):
        O.det_init_(mod, prefix=tag + '.')
        x = x.clone().requires_grad_(True)
        y = mod(x)
        dy = O.det_fill(tag + '.dy', tuple(y.shape), 1.0)
        y.backward(dy)
        save(tag, y=np32(y), dx=np32(x.grad), **grads_of(mod, grad_keys))

    x64 = O.det_fill('x64', (2, 64, 10, 12), 1.0)
    run_module('clam', R.CLAM(64), x64, ['fc1.weight', 'fc2.weight'])
    run_module('slam', R.SLAM(7), x64, ['conv1.weight'])
    run_module('cgam', R.CGAM(64), x64 * 0.3, ['gamma'])
    run_module('sgam', R.SGAM(64), x64, ['gamma', 'query_conv.weight', 'key_conv.bias', 'value_conv.weight'])
    run_module('rab', R.RAB(64, 64), x64, ['conv1.weight', 'conv2.bias', 'ca.fc1.weight', 'sa.conv1.weight',
                                           'conv.weight'])
    run_module('resgroup', R.ResGroup(R.RAB, n_blocks=2), x64,
               ['RG.1.conv2.weight', 'ca.fc2.weight', 'sa.conv1.weight', 'conv.bias'])
    x3 = O.det_fill('x3', (2, 3, 10, 12), 0.5, 0.5)
    run_module('msb', R.MSB(3, 64), x3, ['conv1.weight', 'conv2.0.weight', 'conv2.1.bias', 'conv.weight'])
    for s in (2, 3, 4, 8, 9):                 # x8 = three tied x2 stages (sradsgan.py:388-392), x9 = two tied x3 stages
        run_module('gabup_x%d' % s, R.GAB_UP(upscale_factor=s), x64[:1, :, :6, :7] * 0.3,
                   ['upsampling.0.weight', 'upsampling.0.bias', 'conv.weight', 'ca.gamma', 'sa.gamma'])
    for s in (2, 3, 4, 8, 9):
        g = R.GeneratorResNet(R.ResGroup, n_residual_blocks=2, n_basic_blocks=1, upscale_factor=s)
        run_module('gen_small_x%d' % s, g, x3[:1],
                   ['conv1.0.weight', 'res_groups.0.RG.0.conv1.weight', 'res_groups.1.conv.weight',
                    'GAB_UP.upsampling.0.weight', 'MSB.conv.weight', 'conv3.0.bias'])

    # ---- 3. discriminator: forward, BN running stats, input grad ----------------------------- #
    d = R.Discriminator()
    O.det_init_(d, prefix='D.')
    img = O.det_fill('dimg', (2, 3, 32, 32), 0.5, 0.5).requires_grad_(True)
    out = d(img)
    out.backward(O.det_fill('D.dy', tuple(out.shape), 1.0))
    sd = d.state_dict()
    save('disc', y=np32(out), dx=np32(img.grad),
         rm3=np32(sd['model.3.running_mean']), rv3=np32(sd['model.3.running_var']),
         rm23=np32(sd['model.23.running_mean']), rv23=np32(sd['model.23.running_var']),
         nbt=sd['model.3.num_batches_tracked'].numpy(),
         **grads_of(d, ['model.0.weight', 'model.3.weight', 'model.17.fc1.weight', 'model.18.conv1.weight',
                        'model.25.weight', 'model.22.bias']))

    # ---- 4. gradient penalty through the reference's own method (np.random alpha) ------------- #
    d = R.Discriminator()
    O.det_init_(d, prefix='D.')
    real = O.det_fill('gp.real', (2, 3, 32, 32), 0.5, 0.5)
    fake = O.det_fill('gp.fake', (2, 3, 32, 32), 0.5, 0.5)

    class _Self:
        gpu_mode = False
    np.random.seed(123)
    alpha = np.random.random((2, 1, 1, 1))
    np.random.seed(123)
    gp = R.SRADSGAN.gradient_penalty(_Self(), d, real, fake, 'L2', 'LS')
    save('gradient_penalty', alpha=alpha.astype(np.float32), gp=np.float64(gp.item()),
         **grads_of(d, ['model.0.weight', 'model.3.weight', 'model.3.bias', 'model.11.weight',
                        'model.17.fc2.weight', 'model.18.conv1.weight', 'model.25.weight']))

    # ---- 5. two full training iterations, reference lines 829-892 replayed by hand ----------- #
    def ref_train(tag, n_groups, n_blocks, batch, lr_side, scale, iters, thr):
        torch.set_num_threads(thr)
        G = R.GeneratorResNet(R.ResGroup, n_residual_blocks=n_groups, n_basic_blocks=n_blocks, upscale_factor=scale)
        D = R.Discriminator()
        Fx = vgg_standin()
        O.det_init_(G, prefix='G.')
        O.det_init_(D, prefix='D.')
        O.det_init_(Fx, prefix='F.')
        crit = torch.nn.L1Loss()
        gan = R.GANLoss('wgan-gp')
        oG = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.9, 0.999))
        oD = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.9, 0.999))
        rec = {}
        for it in range(iters):
            lr_img = O.det_fill('%s.lr.%d' % (tag, it), (batch, 3, lr_side, lr_side), 0.5, 0.5)
            hr_img = O.det_fill('%s.hr.%d' % (tag, it), (batch, 3, lr_side * scale, lr_side * scale), 0.5, 0.5)
            oG.zero_grad()
            gen = G(lr_img)
            pixel = crit(gen, hr_img)
            content = crit(Fx(gen), Fx(hr_img).detach())
            lgan = gan(D(gen), True)
            loss_G = pixel + 1e-2 * content + 1e-3 * lgan
            loss_G.backward()
            if it == 0:
                rec.update({'it0_' + k: v for k, v in grads_of(
                    G, ['conv1.0.weight', 'res_groups.0.RG.0.conv1.weight', 'GAB_UP.upsampling.0.weight',
                        'GAB_UP.sa.gamma', 'conv3.0.weight']).items()})
            oG.step()
            oD.zero_grad()
            loss_D = gan(D(hr_img), True) + gan(D(gen.detach()), False)
            np.random.seed(1000 + it)
            alpha = np.random.random((batch, 1, 1, 1)).astype(np.float32)
            np.random.seed(1000 + it)
            gp = R.SRADSGAN.gradient_penalty(_Self(), D, hr_img.data, gen.detach().data, 'L2', 'LS')
            loss_D = loss_D + 10.0 * gp
            loss_D.backward()
            if it == 0:
                rec.update({'it0_D_' + k: v for k, v in grads_of(D, ['model.0.weight', 'model.25.weight']).items()})
            oD.step()
            for p in D.parameters():
                p.data.clamp_(-0.01, 0.01)
            rec['alpha%d' % it] = alpha
            rec['scalars%d' % it] = np.array([loss_G.item(), loss_D.item(), pixel.item(), content.item(),
                                              lgan.item(), gp.item()], dtype=np.float64)
            if it == 0:
                rec['gen0_crop'] = np32(gen[:, :, :8, :8])
                u8 = O.to_uint8_hwc(gen[0])
                gt = O.to_uint8_hwc(hr_img[0])
                rec['psnr0'] = np.float64(srutils.psnr(u8, gt))
            print(tag, it, rec['scalars%d' % it])
        gs, ds = G.state_dict(), D.state_dict()
        for k in ['conv1.0.weight', 'res_groups.0.RG.0.conv2.bias', 'GAB_UP.sa.gamma', 'GAB_UP.upsampling.0.weight',
                  'conv3.0.weight']:
            rec['G_after__' + k.replace('.', '__')] = np32(gs[k]).ravel()[:64]
        for k in ['model.0.weight', 'model.3.weight', 'model.3.running_mean', 'model.25.weight']:
            rec['D_after__' + k.replace('.', '__')] = np32(ds[k]).ravel()[:64]
        save(tag, **rec)

    if ONLY and not ({'train_small', 'train_full'} & ONLY):
        return
    ref_train('train_small', n_groups=2, n_blocks=1, batch=2, lr_side=8, scale=4, iters=2, thr=8)
    ref_train('train_full', n_groups=12, n_blocks=3, batch=2, lr_side=54, scale=4, iters=2, thr=8)


if __name__ == '__main__':
    main()
