"""Generate tests/golden/srgan_x{2,3,4}.npz and srgan_step.npz by running the REFERENCE model/srgan.py classes (build
container only; same stub import as oracle/make_golden.py).  Parameters come from the deterministic filler keyed by
state_dict name, inputs from det_fill.  Stored per scale: generator output, digests of the parameter gradients of an MSE
loss, BatchNorm running statistics after the call.  srgan_step.npz: the scalars of two training iterations driven with
the reference's GeneratorResNet / Discriminator, torch's MSELoss and Adam in the order of srgan.py:335-365 (the loop
itself lives inside SRGAN.train() behind its dataloader and cannot be called), plus digests of weights afterwards."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import sradsgan_ref as O  # noqa: E402
from oracle.make_golden import import_reference  # noqa: E402

STEP_KEYS = ['conv1.0.weight', 'res_blocks.1.conv_block.3.weight', 'res_blocks.0.conv_block.1.weight',
             'upsampling.0.weight', 'upsampling.1.bias', 'conv3.0.weight']
STEP_KEYS_D = ['model.0.weight', 'model.3.weight', 'model.11.weight', 'model.23.weight']


def main():
    import_reference()
    # srgan.py:34 imports two names data/data.py no longer defines (the file only imports as shipped with an older
    # data.py); the model classes never touch them, so they are set to None on the already-imported module
    dd = sys.modules['data.data']
    for name in ('get_training_datasets', 'get_test_datasets'):
        if not hasattr(dd, name):
            setattr(dd, name, None)
    srgan = importlib.import_module('model.srgan')
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    for scale in (2, 3, 4):
        net = srgan.GeneratorResNet(3, 3, n_residual_blocks=2, upscale_factor=scale)
        O.det_init_(net, prefix='S.')
        x = O.det_fill('srgan.x.%d' % scale, (2, 3, 12, 10), 0.5, 0.5)
        tgt = O.det_fill('srgan.t.%d' % scale, (2, 3, 12 * scale, 10 * scale), 0.5, 0.5)
        y = net(x)
        loss = torch.nn.functional.mse_loss(y, tgt)
        loss.backward()
        out = {'y': y.detach().numpy(), 'loss': np.float32(loss.item()), 'keys': np.array(sorted(net.state_dict().keys()))}
        seen = set()
        for k, p in net.named_parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out['grad__' + k.replace('.', '__')] = O.digest(p.grad)
        for k, b in net.named_buffers():
            out['buf__' + k.replace('.', '__')] = O.digest(b.float())
        np.savez_compressed(os.path.join(out_dir, 'srgan_x%d.npz' % scale), **out)
        print('x%d: y %s loss %.6f, %d grads' % (scale, tuple(y.shape), loss.item(), len(seen)))

    # two training iterations, x4, LR 16x16 -> HR 64x64, D patch 4x4
    G = srgan.GeneratorResNet(3, 3, n_residual_blocks=2, upscale_factor=4)
    D = srgan.Discriminator()
    Fx = O.FeatureExtractor()
    O.det_init_(G, prefix='S.')
    O.det_init_(D, prefix='SD.')
    O.det_init_(Fx, prefix='V.')
    for p in Fx.parameters():
        p.requires_grad_(False)
    opt_G = torch.optim.Adam(G.parameters(), lr=2e-4, betas=(0.9, 0.999))
    opt_D = torch.optim.Adam(D.parameters(), lr=2e-4, betas=(0.9, 0.999))
    mse = torch.nn.MSELoss()
    rows = []
    for it in range(2):
        lr_img = O.det_fill('srgan.step.lr.%d' % it, (4, 3, 16, 16), 0.5, 0.5)
        hr_img = O.det_fill('srgan.step.hr.%d' % it, (4, 3, 64, 64), 0.5, 0.5)
        opt_G.zero_grad()                                              # srgan.py:335-349
        gen_hr = G(lr_img)
        validity = D(gen_hr)
        valid, fake = torch.ones_like(validity), torch.zeros_like(validity)
        loss_gan = mse(validity, valid)
        content = mse(Fx(gen_hr), Fx(hr_img).detach())
        pixel = mse(gen_hr, hr_img)
        loss_G = pixel + 6e-3 * content + 1e-3 * loss_gan
        loss_G.backward()
        opt_G.step()
        opt_D.zero_grad()                                              # :355-365
        loss_real = mse(D(hr_img), valid)
        loss_fake = mse(D(gen_hr.detach()), fake)
        loss_D = (loss_real + loss_fake) / 2
        loss_D.backward()
        opt_D.step()
        rows.append([loss_G.item(), loss_D.item(), pixel.item(), content.item(), loss_gan.item(), loss_real.item(),
                     loss_fake.item()])
        print('it%d' % it, rows[-1])
    out = {'scalars': np.array(rows, dtype=np.float64),
           'names': np.array(['loss_G', 'loss_D', 'pixel', 'content', 'loss_gan', 'loss_real', 'loss_fake']),
           'keys_D': np.array(sorted(D.state_dict().keys()))}
    gs, ds = G.state_dict(), D.state_dict()
    for k in STEP_KEYS:
        out['G__' + k.replace('.', '__')] = O.digest(gs[k])
    for k in STEP_KEYS_D:
        out['D__' + k.replace('.', '__')] = O.digest(ds[k])
    out['D__bn_running_var'] = O.digest(ds['model.3.running_var'])
    np.savez_compressed(os.path.join(out_dir, 'srgan_step.npz'), **out)


if __name__ == '__main__':
    main()
