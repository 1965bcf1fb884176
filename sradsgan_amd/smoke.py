"""One small invocation of the hot path on a real MI355X, checked against the CPU oracle.
Called by __graft_entry__.smoke(); the oracle is imported here only as the checker."""
import os
import sys

import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(device):
    if _ROOT not in sys.path:
        sys.path.insert(0, _ROOT)
    from . import _hip
    _hip.lib()                                   # fail loudly when the HIP library is not built
    if device.type != 'cuda' or not torch.cuda.is_available():
        raise RuntimeError('smoke: needs a HIP device (cuda:0); there is no CPU fallback')
    from tests.parity_util import train_parity
    # x4, 2 ResGroups x 1 RAB, batch 2, LR 8x8 -> HR 32x32, two full G+D iterations (incl. the
    # WGAN-GP double backward) on both paths with identical deterministic weights/inputs
    worst, wdiff = train_parity(device, 'smoke', 2, 1, 2, 8, 4, 2)
    torch.cuda.synchronize()
    print('smoke: max |scalar diff| vs oracle = %.3e, weight score = %.3e' % (worst, wdiff))
    if not (worst < 1e-3 and wdiff < 5e-3):
        raise AssertionError('smoke: HIP path differs from the oracle (%.3e, %.3e)' % (worst, wdiff))
