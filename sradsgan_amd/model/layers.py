"""Parameter containers whose forward runs on the HIP C ABI.  They subclass the torch.nn containers
only to keep constructor arguments, parameter names (`weight`, `bias`, BN buffers) and therefore
state_dict keys identical to the reference's nn.Conv2d / nn.BatchNorm2d."""
import torch
import torch.nn as nn

from .. import ops


class HipConv2d(nn.Conv2d):
    """nn.Conv2d replacement: implicit-GEMM MFMA conv with the call site's elementwise tail fused."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        k, s, p, d = self.kernel_size, self.stride, self.padding, self.dilation
        if k[0] != k[1] or s[0] != s[1] or p[0] != p[1] or d != (1, 1) or self.groups != 1:
            raise NotImplementedError('HipConv2d: square kernel/stride/pad, dilation 1, groups 1 only')

    def forward(self, x, act_slope=None, residual=None):
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], act_slope, residual)


class HipBatchNorm2d(nn.BatchNorm2d):
    """Train-mode BatchNorm2d (+ optional fused LeakyReLU) used by the discriminator
    (reference sradsgan.py:478-479).  Differentiable twice (gradient penalty)."""

    def forward(self, x, act_slope=None):
        return ops.batch_norm_act(x, self, act_slope)
