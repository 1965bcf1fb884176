"""Generate tests/golden/edsr_x{2,3,4}.npz by running the REFERENCE model/edsr.py Net (build container only; same stub
import as oracle/make_golden.py).  Parameters come from the deterministic filler keyed by state_dict name, inputs from
det_fill; stored: the output and digests of the gradients of an L1 loss (the EDSR trainer's criterion, edsr.py)."""
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
    edsr = importlib.import_module('model.edsr')
    for scale in (2, 3, 4):
        torch.manual_seed(0)
        net = edsr.Net(num_channels=3, base_filter=256, num_residuals=2, upscale_factor=scale)
        O.det_init_(net, prefix='E.')
        x = O.det_fill('edsr.x.%d' % scale, (2, 3, 12, 10), 0.5, 0.5)
        tgt = O.det_fill('edsr.t.%d' % scale, (2, 3, 12 * scale, 10 * scale), 0.5, 0.5)
        y = net(x)
        loss = torch.nn.functional.l1_loss(y, tgt)
        loss.backward()
        out = {'y': y.detach().numpy(), 'loss': np.float32(loss.item()), 'keys': np.array(sorted(net.state_dict().keys()))}
        seen = set()
        for k, p in net.named_parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out['grad__' + k.replace('.', '__')] = O.digest(p.grad)
        np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'edsr_x%d.npz' % scale), **out)
        print('x%d: y %s loss %.6f, %d grads' % (scale, tuple(y.shape), loss.item(), len(seen)))


if __name__ == '__main__':
    main()
