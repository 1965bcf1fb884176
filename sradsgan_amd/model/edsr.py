"""EDSR generator on the same HIP kernels (BASELINE configs[0]).  Mirrors
SRADSGAN/model/edsr.py:23-75 (`Net`) with the building blocks of model/base_networks.py it uses (ConvBlock :170-208
without norm/activation, ResnetBlock :246-298 with norm=None, activation='relu'): same constructor, same state_dict
keys (`input_conv.conv.*`, `residual_layers.N.conv{1,2}.*`, `mid_conv.conv.*`, `upsampling.{0,3}.*`,
`output_conv.conv.*`), weight-tied upsampler stages as in the reference."""
import math

import torch
import torch.nn as nn

from .layers import HipConv2d
from .sradsgan import _ShuffleAct


class ConvBlock(nn.Module):
    """base_networks.py:170-208 for norm=None and activation in (None, 'relu', 'lrelu')."""

    def __init__(self, input_size, output_size, kernel_size=4, stride=2, padding=1, bias=True, activation=None, norm=None):
        super().__init__()
        if norm is not None or activation not in (None, 'relu', 'lrelu'):
            raise NotImplementedError('ConvBlock: only norm=None and activation None/relu/lrelu are on the HIP path')
        self.conv = HipConv2d(input_size, output_size, kernel_size, stride, padding, bias=bias)
        self.slope = {None: None, 'relu': 0.0, 'lrelu': 0.2}[activation]

    def forward(self, x, residual=None):
        return self.conv(x, self.slope, residual)


class ResnetBlock(nn.Module):
    """base_networks.py:246-298 with norm=None: conv1 -> ReLU -> conv2 -> + x (the add rides in conv2's epilogue)."""

    def __init__(self, num_filter, kernel_size=3, stride=1, padding=1, bias=True, activation='relu', norm=None):
        super().__init__()
        if norm is not None or activation != 'relu':
            raise NotImplementedError('ResnetBlock: norm=None, activation="relu" only')
        self.conv1 = HipConv2d(num_filter, num_filter, kernel_size, stride, padding, bias=bias)
        self.conv2 = HipConv2d(num_filter, num_filter, kernel_size, stride, padding, bias=bias)

    def forward(self, x):
        return self.conv2(self.conv1(x, 0.0), None, x)


class Net(nn.Module):
    """edsr.py:23-75.  The reference hard-codes 256 channels in the upsampler, so base_filter must be 256."""

    def __init__(self, num_channels, base_filter, num_residuals, upscale_factor=3):
        super().__init__()
        self.input_conv = ConvBlock(num_channels, base_filter, 3, 1, 1, activation=None, norm=None)
        self.residual_layers = nn.Sequential(*[ResnetBlock(base_filter, norm=None) for _ in range(num_residuals)])
        self.mid_conv = ConvBlock(base_filter, base_filter, 3, 1, 1, activation=None, norm=None)
        if (upscale_factor & (upscale_factor - 1)) == 0:
            r, stages = 2, int(math.log(upscale_factor, 2))
        elif upscale_factor % 3 == 0:
            r, stages = 3, int(math.log(upscale_factor, 3))
        else:
            r, stages = 1, 0
        # one conv object repeated => tied weights (edsr.py:45-58); slots 1, 2 keep the reference's indices
        stage = [HipConv2d(256, 256 * r * r, 3, 1, 1), _ShuffleAct(r), nn.Identity()]
        self.upsampling = nn.Sequential(*(stage * stages))
        self.output_conv = ConvBlock(base_filter, num_channels, 3, 1, 1, activation=None, norm=None)

    def forward(self, x):
        residual = self.input_conv(x)
        out = self.residual_layers(residual)
        out = self.mid_conv(out, residual)            # torch.add(out, residual), edsr.py:70
        out = self.upsampling(out)
        return self.output_conv(out)
