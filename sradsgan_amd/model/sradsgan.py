"""MI355X-native mirror of the reference's model graph (SRADSGAN/model/sradsgan.py:35-508).

Same class names, constructor signatures and state_dict keys as the reference, so its checkpoints
load unchanged and `main_sradsgan.py`-style drivers keep working; every forward runs on the HIP
kernels of libsradsgan_hip.so (through sradsgan_amd.ops).  Inputs/outputs are fp32, logical NCHW;
intermediate activations are NHWC in memory (torch channels_last).

Behaviours of the reference that are reproduced on purpose (SURVEY.md 8(a) "quirks"):
  * the x4/x8/x9 upsampler stages share ONE conv (sradsgan.py:381-392) and state_dict lists it under
    every stage index;
  * LeakyReLU slopes: 0.2 in RAB and the discriminator, 0.01 in head conv / MSB / upsampler;
  * the discriminator's CAM/PAM tail is never built (sradsgan.py:497 compares a list with 8);
  * CGAM/SGAM gammas start at 0 and are not touched by weights_init_normal.
"""
import math

import torch
import torch.nn as nn

from .. import ops
from .base_networks import ChannelAttention, SpatialAttention, _Clam, _Slam
from .layers import HipBatchNorm2d, HipConv2d


class GANLoss(nn.Module):
    """sradsgan.py:35-67.  Only the 'wgan-gp' flavour is on the hot path: -mean / +mean."""

    def __init__(self, gan_type, real_label_val=1.0, fake_label_val=0.0):
        super().__init__()
        self.gan_type = gan_type.lower()
        self.real_label_val, self.fake_label_val = real_label_val, fake_label_val
        if self.gan_type != 'wgan-gp':
            raise NotImplementedError('GAN type [{:s}] is not found'.format(self.gan_type))

    def forward(self, input, target_is_real):
        m = ops.mean(input)                       # srhip_mean_fwd: deterministic two-stage sum, result stays on the device
        return -m if target_is_real else m


class CLAM(_Clam):
    """Channel local attention, sradsgan.py:101-127."""


class SLAM(_Slam):
    """Spatial local attention, sradsgan.py:129-151."""


class SGAM(nn.Module):
    """Position (spatial) global attention, sradsgan.py:153-176."""

    def __init__(self, in_dim):
        super().__init__()
        self.chanel_in = in_dim
        self.query_conv = HipConv2d(in_dim, in_dim // 8, kernel_size=1)
        self.key_conv = HipConv2d(in_dim, in_dim // 8, kernel_size=1)
        self.value_conv = HipConv2d(in_dim, in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return ops.sgam(x, self.query_conv(x), self.key_conv(x), self.value_conv(x), self.gamma)


class CGAM(nn.Module):
    """Channel global attention, sradsgan.py:178-213 (light=False is the only variant ever built)."""

    def __init__(self, in_dim, light=False):
        super().__init__()
        if light:
            raise NotImplementedError('CGAM(light=True) is not reachable from the SRADSGAN path')
        self.chanel_in, self.light = in_dim, light
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return ops.cgam(x, self.gamma)


class _AttentionTail:
    """The CLAM/SLAM/1x1 tail shared by RAB (sradsgan.py:254-274) and ResGroup (:303-323)."""

    def _build_tail(self, la_mode, pool_mode, planes, addconv):
        self.la_mode, self.addconv = la_mode, addconv
        if 'CA' in la_mode:
            self.ca = CLAM(planes, pool_mode=pool_mode)
        if 'SA' in la_mode:
            self.sa = SLAM(kernel_size=7, pool_mode=pool_mode)
        if '|' in la_mode:
            self.conv = HipConv2d(planes * 2, planes, kernel_size=1, bias=True)
        if '-' in la_mode and addconv:
            self.conv = HipConv2d(planes, planes, kernel_size=1, bias=True)
        if la_mode == '':
            self.last_conv = HipConv2d(64, 64, kernel_size=1, bias=True)

    def _tail(self, out, skip, emit_pp=False):
        """attention tail followed by `out += skip`; the add is fused into the closing 1x1 conv."""
        m = self.la_mode
        if (m == 'CA-SA' and self.addconv and self.ca.pool_mode == 'Avg|Max' and self.sa.pool_mode == 'Avg|Max'
                and ops.attention_tail_supported(out, self.ca.fc1.weight, self.sa.conv1.weight, self.conv.weight)):
            return ops.attention_tail(out, skip, self.ca.fc1.weight, self.ca.fc2.weight, self.sa.conv1.weight,
                                      self.conv.weight, self.conv.bias, emit_pp=emit_pp)
        if m in ('CA-SA', 'SA-CA'):
            first, second = (self.ca, self.sa) if m == 'CA-SA' else (self.sa, self.ca)
            out = second(first(out))
            return self.conv(out, residual=skip) if self.addconv else out + skip
        if m == 'CA':
            return self.ca(out) + skip
        if m == 'SA':
            return self.sa(out) + skip
        if m == 'CA|SA':
            return self.conv(torch.cat([self.ca(out), self.sa(out)], dim=1), residual=skip)
        if m == '':
            return self.last_conv(out, residual=skip)
        return out + skip


class RAB(nn.Module, _AttentionTail):
    """Residual attention block, sradsgan.py:215-275."""

    def __init__(self, inplanes, planes, kernel_size=3, stride=1, padding=1, bias=True, dilation=1,
                 act_type='lrelu', la_mode='CA-SA', pool_mode='Avg|Max', addconv=True):
        super().__init__()
        if act_type != 'lrelu':
            raise NotImplementedError('RAB: only act_type="lrelu" is built by the SRADSGAN path')
        self.inplanes, self.planes = inplanes, planes
        self.conv1 = HipConv2d(inplanes, 4 * planes, kernel_size, stride, padding, bias=bias, dilation=dilation)
        self.conv2 = HipConv2d(4 * planes, planes, kernel_size, stride, padding, bias=bias, dilation=dilation)
        self._build_tail(la_mode, pool_mode, planes, addconv)

    def _fusable(self, x):
        return (self.la_mode == 'CA-SA' and self.addconv and self.ca.pool_mode == 'Avg|Max'
                and self.sa.pool_mode == 'Avg|Max' and self.inplanes == 64 and self.planes == 64
                and self.conv1.kernel_size == (3, 3) and self.conv1.stride == (1, 1) and self.conv1.padding == (1, 1)
                and ops.attention_tail_supported(x, self.ca.fc1.weight, self.sa.conv1.weight, self.conv.weight))

    def forward(self, x):
        if self._fusable(x):
            return ops.rab_block(x, self.conv1.weight, self.conv1.bias, self.conv2.weight, self.conv2.bias,
                                 self.ca.fc1.weight, self.ca.fc2.weight, self.sa.conv1.weight, self.conv.weight,
                                 self.conv.bias, emit_pp=getattr(self, '_next_is_rab', False),
                                 group_first=getattr(self, '_group_first', False))
        out = self.conv2(self.conv1(x, act_slope=0.2))
        return self._tail(out, x)


class ResGroup(nn.Module, _AttentionTail):
    """n_blocks RABs + attention tail + skip, sradsgan.py:277-324."""

    def __init__(self, block, n_blocks=5, nc=64, kernel_size=3, stride=1, bias=True, padding=1,
                 act_type='lrelu', mode='CNA', rla_mode='CA-SA', bla_mode='CA-SA', pool_mode='Avg|Max',
                 addconv=True):
        super().__init__()
        self.RG = nn.Sequential(*[
            block(nc, nc, kernel_size=kernel_size, bias=bias, stride=stride, padding=padding, act_type='lrelu',
                  la_mode=bla_mode, pool_mode=pool_mode, addconv=addconv) for _ in range(n_blocks)])
        for blk, nxt in zip(list(self.RG)[:-1], list(self.RG)[1:]):      # a RAB feeding a RAB may hand its output over in both forms (ops.rab_block)
            blk._next_is_rab = isinstance(blk, RAB) and isinstance(nxt, RAB)
        if n_blocks and isinstance(self.RG[0], RAB):
            self.RG[0]._group_first = True       # its backward is the group's last: the group's weight gradients leave as ONE launch (ops.rab_block)
        self._build_tail(rla_mode, pool_mode, nc, addconv)

    def forward(self, x):
        first = self.RG[0]
        if isinstance(first, RAB) and first._fusable(x):
            ops.carry_open(x)          # x's other consumers (this group's skip, the trunk's bus) stash their gradients for the first RAB's backward
        nr = self.__dict__.get('_next_rab')            # the first RAB of the group this group feeds (GeneratorResNet.__init__), not a submodule
        emit = nr is not None and nr._fusable(x) and ops.rab_planes_ok(x, nr.conv1.weight, nr.conv2.weight)
        return self._tail(self.RG(x), x, emit_pp=emit)  # a group feeding a group: the output also as padded planes (ops.attention_tail)


class MSB(nn.Module):
    """Multi-scale block, sradsgan.py:326-345."""

    def __init__(self, inplanes, planes):
        super().__init__()
        self.inplanes, self.planes = inplanes, planes
        self.conv1 = HipConv2d(inplanes, planes, 3, 1, 1)
        self.conv2 = nn.Sequential(HipConv2d(inplanes, planes, kernel_size=1, bias=True),
                                   HipConv2d(planes, planes, 3, 1, 1))
        self.conv3 = HipConv2d(inplanes, planes, kernel_size=1, bias=True)
        self.conv = HipConv2d(planes * 3, planes, kernel_size=1, bias=True)

    def forward(self, x):
        cat = ops.cat_channels([self.conv1(x), self.conv2(x), self.conv3(x)])      # torch.cat(dim=1) in one pass (srhip_cat_channels)
        return self.conv(cat, act_slope=0.01)


class _ShuffleAct(nn.Module):
    """Stands in the Sequential slot of nn.PixelShuffle; also applies the LeakyReLU that follows."""

    def __init__(self, r):
        super().__init__()
        self.upscale_factor = r

    def forward(self, x):
        return ops.pixel_shuffle_act(x, self.upscale_factor, 0.01)


class GAB_UP(nn.Module):
    """Global attention block + weight-tied pixel-shuffle upsampler, sradsgan.py:365-418."""

    def __init__(self, ga_mode='CA-SA', addconv=True, upscale_factor=4):
        super().__init__()
        self.ga_mode, self.addconv = ga_mode, addconv
        if 'CA' in ga_mode:
            self.ca = CGAM(64)
        if 'SA' in ga_mode:
            self.sa = SGAM(64)
        if '-' in ga_mode and addconv:
            self.conv = HipConv2d(64, 64, kernel_size=1, bias=True)
        if '|' in ga_mode:
            self.conv = HipConv2d(64 * 2, 64, kernel_size=1, bias=True)
        if upscale_factor & (upscale_factor - 1) == 0:
            r, stages = 2, int(math.log(upscale_factor, 2))
        elif upscale_factor % 3 == 0:
            r, stages = 3, int(math.log(upscale_factor, 3))
        else:
            r, stages = 1, 0
        # one conv object repeated => tied weights; slots 1,2 (4,5,...) keep the reference's indices
        stage = [HipConv2d(64, 64 * r * r, 3, 1, 1), _ShuffleAct(r), nn.Identity()]
        self.upsampling = nn.Sequential(*(stage * stages))

    def forward(self, x):
        m, out = self.ga_mode, x
        if m == 'CA':
            out = self.ca(out)
        elif m == 'SA':
            out = self.sa(out)
        elif m in ('CA-SA', 'SA-CA'):
            first, second = (self.ca, self.sa) if m == 'CA-SA' else (self.sa, self.ca)
            out = second(first(out))
            if self.addconv:
                out = self.conv(out)
        elif m == 'CA|SA':
            out = self.conv(torch.cat([self.ca(out), self.sa(out)], dim=1))
        return self.upsampling(out)


class GeneratorResNet(nn.Module):
    """sradsgan.py:420-468."""

    def __init__(self, buildingblock, in_channels=3, out_channels=3, n_residual_blocks=12, n_basic_blocks=3,
                 rla_mode='CA-SA', bla_mode='CA-SA', ga_mode='CA-SA', pool_mode='Avg|Max', addconv=True,
                 upscale_factor=4):
        super().__init__()
        self.conv1 = nn.Sequential(HipConv2d(in_channels, 64, 3, 1, 1), nn.Identity())
        self.res_groups = nn.Sequential(*[
            buildingblock(RAB, n_blocks=n_basic_blocks, nc=64, kernel_size=3, stride=1, padding=1,
                          act_type='lrelu', mode='CNA', rla_mode=rla_mode, bla_mode=bla_mode, pool_mode=pool_mode,
                          addconv=addconv) for _ in range(n_residual_blocks)])
        groups = list(self.res_groups)
        for grp, nxt in zip(groups[:-1], groups[1:]):        # a group whose output is the input of a group that starts with a RAB
            if isinstance(grp, ResGroup) and isinstance(nxt, ResGroup) and isinstance(nxt.RG[0], RAB):
                grp.__dict__['_next_rab'] = nxt.RG[0]        # (plain attribute: the block stays a submodule of ITS group only)
        self.GAB_UP = GAB_UP(ga_mode=ga_mode, addconv=addconv, upscale_factor=upscale_factor)
        self.MSB = MSB(inplanes=in_channels, planes=64)
        self.conv3 = nn.Sequential(HipConv2d(64, out_channels, 3, 1, 1))

    def forward(self, x):
        x = ops.nhwc(x)
        out = self.conv1[0](x, act_slope=0.01)
        terms = [self.MSB(x), out]
        for group in self.res_groups:
            out = group(out)
            terms.append(out)                # stratified dense sampling bus, sradsgan.py:455-460: bus = bus + out per group,
        bus = ops.sum_tensors(terms)         # summed once, in the same order (srhip_sum_n)
        return self.conv3[0](self.GAB_UP(bus))


class Discriminator(nn.Module):
    """sradsgan.py:470-508: 8 conv blocks (+BN from block 2, LeakyReLU .2), CBAM-style attention
    after block 6, 512->1 output conv.  Always in train mode (batch statistics)."""

    _PLAN = [(64, 1, False), (64, 2, True), (128, 1, True), (128, 2, True),
             (256, 1, True), (256, 2, True), (512, 1, True), (512, 2, True)]

    def __init__(self, in_channels=3, attention=True):
        super().__init__()
        layers, cin = [], in_channels
        self._blocks = []            # (conv idx, bn idx or None)
        for idx, (cout, stride, norm) in enumerate(self._PLAN, start=1):
            entry = [len(layers), None, []]
            layers.append(HipConv2d(cin, cout, 3, stride, 1))
            if norm:
                entry[1] = len(layers)
                layers.append(HipBatchNorm2d(cout))
            layers.append(nn.Identity())          # slot of LeakyReLU(0.2): fused into conv / BN
            if attention and idx == 6:
                entry[2] = [len(layers), len(layers) + 1]
                layers += [ChannelAttention(256), SpatialAttention()]
            self._blocks.append(tuple(entry))
            cin = cout
        layers.append(HipConv2d(cin, 1, 3, 1, 1))
        self.model = nn.Sequential(*layers)

    def forward(self, img):
        x = ops.nhwc(img)
        m = self.model
        for conv_i, bn_i, extra in self._blocks:
            if bn_i is None:
                x = m[conv_i](x, act_slope=0.2)
            else:
                x = m[bn_i](m[conv_i](x), act_slope=0.2)
            for e in extra:
                x = m[e](x)
        return m[len(m) - 1](x)


class FeatureExtractor(nn.Module):
    """vgg19.features[:12] structure (sradsgan.py:88-99): conv-relu x2, pool, conv-relu x2, pool,
    conv-relu; no input normalisation.  torchvision and its pretrained file are not available
    offline, so weights come from `load_state_dict` (torchvision key names `features.N.*` are
    accepted through `load_torchvision_vgg19`) or stay at their deterministic init."""

    _CFG = [(3, 64), 'R', (64, 64), 'R', 'P', (64, 128), 'R', (128, 128), 'R', 'P', (128, 256), 'R']

    def __init__(self):
        super().__init__()
        seq = []
        for item in self._CFG:
            seq.append(HipConv2d(item[0], item[1], 3, 1, 1) if isinstance(item, tuple) else nn.Identity())
        self.feature_extractor = nn.Sequential(*seq)

    def load_torchvision_vgg19(self, state_dict):
        own = {'feature_extractor.' + k[len('features.'):]: v for k, v in state_dict.items()
               if k.startswith('features.') and int(k.split('.')[1]) < 12}
        return self.load_state_dict(own, strict=True)

    def forward(self, img):
        x = ops.nhwc(img)
        fe = self.feature_extractor
        frozen = not any(p.requires_grad for p in self.parameters())
        if frozen and x.is_cuda and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0:
            # the training step's case: weights are in no optimiser (sradsgan.py:724-725), only d/d(img) is needed
            wb = []
            for i, item in enumerate(self._CFG):
                if isinstance(item, tuple):
                    wb += [fe[i].weight, fe[i].bias]
            return ops.vgg_features(x, wb)
        for i, item in enumerate(self._CFG):
            if isinstance(item, tuple):
                x = fe[i](x, act_slope=0.0)          # conv + ReLU fused
            elif item == 'P':
                x = ops.max_pool2x2(x)
        return x
