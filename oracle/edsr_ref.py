"""TEST INFRASTRUCTURE (oracle) -- not product code.  CPU restatement in stock torch ops of the reference's EDSR
generator: SRADSGAN/model/edsr.py:23-75 (`Net`) and the two blocks it takes from model/base_networks.py (ConvBlock
:170-208, ResnetBlock :246-298) for the arguments EDSR passes (norm=None; activation None / 'relu').
Pinned by tests/golden/edsr_x{2,3}.npz, produced by importing the reference itself (oracle/make_golden_edsr.py)."""
import math

import torch
import torch.nn as nn


class ConvBlock(nn.Module):                                   # base_networks.py:170-208, norm=None, activation=None
    def __init__(self, input_size, output_size, kernel_size=4, stride=2, padding=1):
        super().__init__()
        self.conv = nn.Conv2d(input_size, output_size, kernel_size, stride, padding, bias=True)

    def forward(self, x):
        return self.conv(x)


class ResnetBlock(nn.Module):                                 # base_networks.py:246-298, norm=None, activation='relu'
    def __init__(self, num_filter):
        super().__init__()
        self.conv1 = nn.Conv2d(num_filter, num_filter, 3, 1, 1, bias=True)
        self.conv2 = nn.Conv2d(num_filter, num_filter, 3, 1, 1, bias=True)

    def forward(self, x):
        return torch.add(self.conv2(torch.relu(self.conv1(x))), x)


class Net(nn.Module):                                         # edsr.py:23-75
    def __init__(self, num_channels, base_filter, num_residuals, upscale_factor=3):
        super().__init__()
        self.input_conv = ConvBlock(num_channels, base_filter, 3, 1, 1)
        self.residual_layers = nn.Sequential(*[ResnetBlock(base_filter) for _ in range(num_residuals)])
        self.mid_conv = ConvBlock(base_filter, base_filter, 3, 1, 1)
        two = [nn.Conv2d(256, 256 * 4, 3, 1, 1), nn.PixelShuffle(2), nn.LeakyReLU()]
        three = [nn.Conv2d(256, 256 * 9, 3, 1, 1), nn.PixelShuffle(3), nn.LeakyReLU()]
        up = []
        if (upscale_factor & (upscale_factor - 1)) == 0:
            for _ in range(int(math.log(upscale_factor, 2))):
                up += two                                     # same module objects again: tied weights (:51-56)
        elif upscale_factor % 3 == 0:
            for _ in range(int(math.log(upscale_factor, 3))):
                up += three
        self.upsampling = nn.Sequential(*up)
        self.output_conv = ConvBlock(base_filter, num_channels, 3, 1, 1)

    def forward(self, x):
        out = self.input_conv(x)
        residual = out
        out = self.mid_conv(self.residual_layers(out))
        out = torch.add(out, residual)
        return self.output_conv(self.upsampling(out))
