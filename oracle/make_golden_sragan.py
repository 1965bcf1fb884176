"""Generate tests/golden/sragan_x{2,3,4}.npz by running the REFERENCE model/sragan.py GeneratorResNet built the way its
trainer builds it (sragan.py:465-467: ResidualBlock_Block_WithAttention of BasicBlocks, 'CA-SA', 'Avg|Max', addconv),
shortened to 2 residual blocks x 3 basic blocks.  Build container only; same stub import as oracle/make_golden.py.
Stored: output, digests of the parameter gradients of an L1 loss (the trainer's pixel criterion), BatchNorm running
statistics after the call, the state_dict key set."""
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


def main():
    import_reference()
    sragan = importlib.import_module('model.sragan')
    base = importlib.import_module('model.base_networks')
    for scale in (2, 3, 4):
        net = sragan.GeneratorResNet(base.ResidualBlock_Block_WithAttention, n_residual_blocks=2, n_basic_blocks=3,
                                     rla_mode='CA-SA', bla_mode='CA-SA', ga_mode='CA-SA', pool_mode='Avg|Max',
                                     addconv=True, upscale_factor=scale)
        O.det_init_(net, prefix='A.')
        x = O.det_fill('sragan.x.%d' % scale, (2, 3, 12, 10), 0.5, 0.5)
        tgt = O.det_fill('sragan.t.%d' % scale, (2, 3, 12 * scale, 10 * scale), 0.5, 0.5)
        y = net(x)
        loss = torch.nn.functional.l1_loss(y, tgt)
        loss.backward()
        out = {'y': y.detach().numpy(), 'loss': np.float32(loss.item()), 'keys': np.array(sorted(net.state_dict().keys()))}
        seen = set()
        for k, p in net.named_parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out['grad__' + k.replace('.', '__')] = O.digest(p.grad)
        for k, b in net.named_buffers():
            out['buf__' + k.replace('.', '__')] = O.digest(b.float())
        np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'sragan_x%d.npz' % scale), **out)
        print('x%d: y %s loss %.6f, %d grads, %d keys' % (scale, tuple(y.shape), loss.item(), len(seen), len(out['keys'])))


if __name__ == '__main__':
    main()
