"""SRAGAN generator on the same HIP kernels.  Mirrors SRADSGAN/model/sragan.py:147-237
(`GeneratorResNet`) and the blocks of model/base_networks.py its trainer builds it from (sragan.py:465-467):
ConvBlock :170-208 (norm=None), BasicBlock :958-1070, ResidualBlock_Block_WithAttention :1505-1595, CAM_Module /
PAM_Module :480-554 -- same constructors for the arguments that path passes, same state_dict keys.  A BasicBlock is the
fused residual-attention node of the SRADSGAN generator (`ops.rab_block`: conv-LeakyReLU-conv, CLAM, SLAM, 1x1, +x)
with 64 mid channels; BatchNorm (conv2, tied up-sampler stages) and PixelShuffle+LeakyReLU run through the C ABI.
SRAGAN's Discriminator (:239-277), GANLoss (:42-74), gradient penalty (:372-418) and training iteration (:539-575) are
identical to SRADSGAN's: use `model.sradsgan.Discriminator / GANLoss` and `train_step.TrainStep` with this generator."""
import math

import torch
import torch.nn as nn

from .. import ops
from .base_networks import ChannelAttention, SpatialAttention
from .layers import HipBatchNorm2d, HipConv2d
from .sradsgan import CGAM as CAM_Module  # noqa: N811  (base_networks.py:513-554, same arithmetic)
from .sradsgan import SGAM as PAM_Module  # noqa: N811  (base_networks.py:480-511)
from .sradsgan import Discriminator, FeatureExtractor, GANLoss, _ShuffleAct  # noqa: F401


class ConvBlock(nn.Module):
    """base_networks.py:170-208 for norm=None and activation in (None, 'relu', 'lrelu')."""

    def __init__(self, input_size, output_size, kernel_size=4, stride=2, padding=1, dilation=1, bias=True, activation=None,
                 norm=None):
        super().__init__()
        if norm is not None or dilation != 1 or activation not in (None, 'relu', 'lrelu'):
            raise NotImplementedError('ConvBlock: norm=None, dilation 1, activation None/relu/lrelu on the HIP path')
        self.conv = HipConv2d(input_size, output_size, kernel_size, stride, padding, bias=bias)
        self.norm, self.activation = norm, activation
        self.slope = {None: None, 'relu': 0.0, 'lrelu': 0.2}[activation]

    def forward(self, x):
        return self.conv(x, self.slope)


class _LocalTail:
    """The 'CA-SA' + addconv tail shared by BasicBlock (:1036-1040) and the residual block (:1568-1572)."""

    def _build_tail(self, la_mode, pool_mode, planes, addconv):
        if la_mode != 'CA-SA' or not addconv or pool_mode != 'Avg|Max':
            raise NotImplementedError('SRAGAN blocks: la_mode "CA-SA", pool_mode "Avg|Max", addconv=True only '
                                      '(what sragan.py:465-467 builds)')
        self.la_mode, self.addconv = la_mode, addconv
        self.ca = ChannelAttention(planes, pool_mode=pool_mode)
        self.sa = SpatialAttention(kernel_size=7, pool_mode=pool_mode)
        self.conv = HipConv2d(planes, planes, kernel_size=1, bias=True)

    def _tail(self, out, skip):
        if ops.attention_tail_supported(out, self.ca.fc1.weight, self.sa.conv1.weight, self.conv.weight):
            return ops.attention_tail(out, skip, self.ca.fc1.weight, self.ca.fc2.weight, self.sa.conv1.weight,
                                      self.conv.weight, self.conv.bias)
        return self.conv(self.sa(self.ca(out)), residual=skip)


class BasicBlock(nn.Module, _LocalTail):
    """base_networks.py:958-1070 with norm_type=None, inplanes == planes: act(tail(conv2(conv1(x))) + x)."""
    expansion = 1

    def __init__(self, inplanes, planes, kernel_size=3, stride=1, padding=1, bias=True, dilation=1, norm_type='batch',
                 act_type=None, la_mode='CA-SA', pool_mode='Avg|Max', addconv=True, downsample=None):
        super().__init__()
        if norm_type is not None or inplanes != planes or act_type not in (None, 'lrelu'):
            raise NotImplementedError('BasicBlock: norm_type=None, inplanes == planes, act_type None/"lrelu" only')
        self.inplanes, self.planes = inplanes, planes
        self.conv1 = ConvBlock(inplanes, planes, kernel_size, stride, padding, dilation, bias, act_type, None)
        self.conv2 = ConvBlock(planes, planes, kernel_size, stride, padding, dilation, bias, None, None)
        self._build_tail(la_mode, pool_mode, planes, addconv)
        self.act_slope = 0.2 if act_type == 'lrelu' else None

    def forward(self, x):
        c1, c2 = self.conv1.conv, self.conv2.conv
        if (self.act_slope == 0.2 and c1.kernel_size == (3, 3) and c1.stride == (1, 1) and c1.padding == (1, 1)
                and ops.attention_tail_supported(x, self.ca.fc1.weight, self.sa.conv1.weight, self.conv.weight)):
            out = ops.rab_block(x, c1.weight, c1.bias, c2.weight, c2.bias, self.ca.fc1.weight, self.ca.fc2.weight,
                                self.sa.conv1.weight, self.conv.weight, self.conv.bias)
        else:
            out = self._tail(self.conv2(self.conv1(x)), x)
        return out if self.act_slope is None else torch.nn.functional.leaky_relu(out, self.act_slope)


class ResidualBlock_Block_WithAttention(nn.Module, _LocalTail):  # noqa: N801  (the reference's class name)
    """base_networks.py:1505-1595: n_blocks-1 activated blocks, a last block without activation (mode 'CNA'), the
    attention tail, + x."""

    def __init__(self, block, n_blocks=1, nc=64, gc=32, kernel_size=3, stride=1, bias=True, padding=1, norm_type='batch',
                 act_type='relu', mode='CNA', rla_mode='CA-SA', bla_mode='CA-SA', pool_mode='Avg|Max', addconv=True):
        super().__init__()
        mk = lambda act: block(nc, nc, kernel_size=kernel_size, bias=bias, stride=stride, padding=padding,
                               norm_type=norm_type, act_type=act, la_mode=bla_mode, pool_mode=pool_mode, addconv=addconv)
        self.blocks = nn.Sequential(*[mk(act_type) for _ in range(n_blocks - 1)])
        self.last_conv = mk(None if mode == 'CNA' else act_type)
        self._build_tail(rla_mode, pool_mode, nc, addconv)

    def forward(self, x):
        return self._tail(self.last_conv(self.blocks(x)), x)


class GeneratorResNet(nn.Module):
    """sragan.py:147-237 for ga_mode 'CA-SA' + addconv.  LeakyReLU(inplace=True) there is the default slope 0.01."""

    def __init__(self, buildingblock, in_channels=3, out_channels=3, n_residual_blocks=12, n_basic_blocks=1,
                 rla_mode='CA-SA', bla_mode='CA-SA', ga_mode='CA-SA', pool_mode='Avg|Max', addconv=True, upscale_factor=3):
        super().__init__()
        if ga_mode != 'CA-SA' or not addconv:
            raise NotImplementedError('GeneratorResNet: ga_mode "CA-SA" with addconv=True only (sragan.py:465-467)')
        self.ga_mode, self.addconv = ga_mode, addconv
        self.conv1 = nn.Sequential(HipConv2d(in_channels, 64, 3, 1, 1), nn.Identity())
        self.res_blocks = nn.Sequential(*[
            buildingblock(BasicBlock, n_blocks=n_basic_blocks, nc=64, gc=32, kernel_size=3, stride=1, padding=1,
                          norm_type=None, act_type='lrelu', mode='CNA', rla_mode=rla_mode, bla_mode=bla_mode,
                          pool_mode=pool_mode, addconv=addconv) for _ in range(n_residual_blocks)])
        self.conv2 = nn.Sequential(HipConv2d(64, 64, 3, 1, 1), HipBatchNorm2d(64))
        self.ca = CAM_Module(64)
        self.sa = PAM_Module(64)
        self.conv = HipConv2d(64, 64, kernel_size=1, bias=True)
        if (upscale_factor & (upscale_factor - 1)) == 0:
            r, stages = 2, int(math.log(upscale_factor, 2))
        elif upscale_factor % 3 == 0:
            r, stages = 3, int(math.log(upscale_factor, 3))
        else:
            r, stages = 1, 0
        stage = [HipConv2d(64, 64 * r * r, 3, 1, 1), HipBatchNorm2d(64 * r * r), _ShuffleAct(r), nn.Identity()]
        self.upsampling = nn.Sequential(*(stage * stages))
        self.conv3 = nn.Sequential(HipConv2d(64, out_channels, 3, 1, 1), nn.Identity())

    def forward(self, x):
        out1 = self.conv1[0](ops.nhwc(x), 0.01)
        out = torch.add(out1, self.conv2[1](self.conv2[0](self.res_blocks(out1))))
        out = self.conv(self.sa(self.ca(out)))
        up = self.upsampling
        for i in range(0, len(up), 4):
            out = up[i + 2](up[i + 1](up[i](out)))
        return torch.tanh(self.conv3[0](out))
