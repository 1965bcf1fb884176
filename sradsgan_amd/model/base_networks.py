"""The two blocks of the reference's model/base_networks.py that the SRADSGAN path reaches:
ChannelAttention (base_networks.py:366-403) and SpatialAttention (:424-457), used by the
discriminator (sradsgan.py:495-496).  They are arithmetically the generator's CLAM / SLAM."""
import torch.nn as nn

from .. import ops
from .layers import HipConv2d


class _Clam(nn.Module):
    """sigmoid(MLP(avgpool x) + MLP(maxpool x)) * x with a shared bias-free 1x1 MLP C -> C/ratio -> C."""

    def __init__(self, in_planes, ratio=16, pool_mode='Avg|Max'):
        super().__init__()
        if pool_mode != 'Avg|Max':               # fail at construction, not at the first forward (ops.clam has no single-pool variant)
            raise NotImplementedError("CLAM / ChannelAttention: only pool_mode 'Avg|Max' is built by the SRADSGAN path "
                                      "(sradsgan.py:669-671)")
        self.pool_mode = pool_mode
        self.fc1 = HipConv2d(in_planes, in_planes // ratio, 1, bias=False)
        self.fc2 = HipConv2d(in_planes // ratio, in_planes, 1, bias=False)

    def forward(self, x):
        return ops.clam(x, self.fc1.weight, self.fc2.weight, self.pool_mode)


class _Slam(nn.Module):
    """sigmoid(conv kxk([mean_c x, max_c x])) * x, k in (3, 7), no bias."""

    def __init__(self, kernel_size=7, pool_mode='Avg|Max'):
        super().__init__()
        assert kernel_size in (3, 7), 'kernel size must be 3 or 7'
        if pool_mode != 'Avg|Max':
            raise NotImplementedError("SLAM / SpatialAttention: only pool_mode 'Avg|Max' is built by the SRADSGAN path "
                                      "(sradsgan.py:669-671)")
        self.pool_mode = pool_mode
        self.conv1 = HipConv2d(2, 1, kernel_size, padding=3 if kernel_size == 7 else 1, bias=False)

    def forward(self, x):
        return ops.slam(x, self.conv1.weight, self.pool_mode)


class ChannelAttention(_Clam):
    pass


class SpatialAttention(_Slam):
    pass
