"""TEST INFRASTRUCTURE (oracle) -- not product code.  CPU restatement in stock torch ops of the reference's SRAGAN
generator: SRADSGAN/model/sragan.py:147-237 (GeneratorResNet) with the blocks it is built from in the only
configuration the trainer instantiates (sragan.py:465-467: ResidualBlock_Block_WithAttention x12 of BasicBlock x5,
norm_type=None, act 'lrelu', 'CA-SA' / 'Avg|Max' / addconv): model/base_networks.py ConvBlock :170-208, BasicBlock
:958-1070, ResidualBlock_Block_WithAttention :1505-1595, ChannelAttention :366-403, SpatialAttention :424-457,
PAM_Module :480-511, CAM_Module :513-554.  The attention arithmetic is oracle/sradsgan_ref's CLAM / SLAM / SGAM / CGAM
(the same formulas under other class names).  SRAGAN's discriminator (sragan.py:239-277), GANLoss (:42-74), gradient
penalty (:372-418) and loop (:539-575) are those of SRADSGAN: oracle/sradsgan_ref.{Discriminator, GANLoss,
gradient_penalty, train_step}.  Pinned by tests/golden/sragan_x{2,3,4}.npz (oracle/make_golden_sragan.py)."""
import math

import torch
import torch.nn as nn

from . import sradsgan_ref as O


class ConvBlock(nn.Module):                                   # base_networks.py:170-208, norm=None
    def __init__(self, cin, cout, kernel_size, stride, padding, bias, activation):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride, padding, bias=bias)
        self.act = {'lrelu': nn.LeakyReLU(0.2), None: None}[activation]

    def forward(self, x):
        y = self.conv(x)
        return y if self.act is None else self.act(y)


def _tail(mod, out):                                          # the 'CA-SA' + addconv branch of :1036-1040 / :1568-1572
    return mod.conv(mod.sa(mod.ca(out)))


class BasicBlock(nn.Module):                                  # base_networks.py:958-1070
    def __init__(self, inplanes, planes, act_type):
        super().__init__()
        self.conv1 = ConvBlock(inplanes, planes, 3, 1, 1, True, act_type)
        self.conv2 = ConvBlock(planes, planes, 3, 1, 1, True, None)
        self.ca, self.sa = O.CLAM(planes), O.SLAM(7)
        self.conv = nn.Conv2d(planes, planes, 1, bias=True)
        self.act = nn.LeakyReLU(0.2) if act_type == 'lrelu' else None

    def forward(self, x):
        out = _tail(self, self.conv2(self.conv1(x))) + x      # inplanes == planes: residual = x (:1023-1026)
        return out if self.act is None else self.act(out)


class ResidualBlock(nn.Module):                               # ResidualBlock_Block_WithAttention, :1505-1595, mode 'CNA'
    def __init__(self, n_blocks, nc=64):
        super().__init__()
        self.blocks = nn.Sequential(*[BasicBlock(nc, nc, 'lrelu') for _ in range(n_blocks - 1)])
        self.last_conv = BasicBlock(nc, nc, None)
        self.ca, self.sa = O.CLAM(nc), O.SLAM(7)
        self.conv = nn.Conv2d(nc, nc, 1, bias=True)

    def forward(self, x):
        return _tail(self, self.last_conv(self.blocks(x))) + x


class GeneratorResNet(nn.Module):                             # sragan.py:147-237, ga_mode 'CA-SA', addconv
    def __init__(self, n_residual_blocks=12, n_basic_blocks=1, upscale_factor=3):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1), nn.LeakyReLU())
        self.res_blocks = nn.Sequential(*[ResidualBlock(n_basic_blocks) for _ in range(n_residual_blocks)])
        self.conv2 = nn.Sequential(nn.Conv2d(64, 64, 3, 1, 1), nn.BatchNorm2d(64))
        self.ca, self.sa = O.CGAM(64), O.SGAM(64)
        self.conv = nn.Conv2d(64, 64, 1, bias=True)
        if (upscale_factor & (upscale_factor - 1)) == 0:
            r, stages = 2, int(math.log(upscale_factor, 2))
        elif upscale_factor % 3 == 0:
            r, stages = 3, int(math.log(upscale_factor, 3))
        else:
            r, stages = 1, 0
        stage = [nn.Conv2d(64, 64 * r * r, 3, 1, 1), nn.BatchNorm2d(64 * r * r), nn.PixelShuffle(r), nn.LeakyReLU()]
        self.upsampling = nn.Sequential(*(stage * stages))    # conv and BatchNorm tied across stages (:190-204)
        self.conv3 = nn.Sequential(nn.Conv2d(64, 3, 3, 1, 1), nn.Tanh())

    def forward(self, x):
        out1 = self.conv1(x)
        out = torch.add(out1, self.conv2(self.res_blocks(out1)))
        out = self.conv(self.sa(self.ca(out)))
        return self.conv3(self.upsampling(out))
