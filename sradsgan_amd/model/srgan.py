"""SRGAN on the same HIP kernels.  Mirrors SRADSGAN/model/srgan.py:57-155 (ResidualBlock,
GeneratorResNet, Discriminator) with the reference's constructors and state_dict keys, and one iteration of its training
loop (:335-365) as `train_step`.  Convolutions (9x9 head / tail included), train-mode BatchNorm (+ReLU / LeakyReLU) and
PixelShuffle(+ReLU) run through the C ABI; tanh, the residual adds and the three MSE reductions are torch element-wise
ops.  FeatureExtractor is the one of model/sradsgan.py (srgan.py:44-55 is the same vgg19.features[:12] slice)."""
import math

import torch
import torch.nn as nn

from .. import ops
from .layers import HipBatchNorm2d, HipConv2d
from .sradsgan import FeatureExtractor  # noqa: F401  (re-exported: srgan.py:44-55)


def weights_init_normal(m):
    """srgan.py:36-42: conv weights N(0, .02) (biases untouched), BatchNorm weights N(1, .02), biases 0."""
    name = m.__class__.__name__
    if name.find('Conv') != -1:
        torch.nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif name.find('BatchNorm') != -1:
        torch.nn.init.normal_(m.weight.data, 1.0, 0.02)
        torch.nn.init.constant_(m.bias.data, 0.0)


class _ShuffleRelu(nn.Module):
    """Sequential slot of nn.PixelShuffle; applies the ReLU of the following slot in the same pass."""

    def __init__(self, r):
        super().__init__()
        self.upscale_factor = r

    def forward(self, x):
        return ops.pixel_shuffle_act(x, self.upscale_factor, 0.0)


class ResidualBlock(nn.Module):
    """srgan.py:57-70: x + BN(conv(ReLU(BN(conv(x))))); slot 2 is the ReLU, fused into the first BN pass."""

    def __init__(self, in_features):
        super().__init__()
        self.conv_block = nn.Sequential(HipConv2d(in_features, in_features, 3, 1, 1), HipBatchNorm2d(in_features),
                                        nn.Identity(),
                                        HipConv2d(in_features, in_features, 3, 1, 1), HipBatchNorm2d(in_features))

    def forward(self, x):
        cb = self.conv_block
        t = cb[1](cb[0](x), act_slope=0.0)
        return x + cb[4](cb[3](t))


class GeneratorResNet(nn.Module):
    """srgan.py:72-121.  The up-sampler stage (conv, BatchNorm, PixelShuffle, ReLU) is ONE set of module objects
    repeated per stage, so conv and BatchNorm (running statistics included) are tied across stages as in the reference."""

    def __init__(self, in_channels=3, out_channels=3, n_residual_blocks=16, upscale_factor=3):
        super().__init__()
        self.conv1 = nn.Sequential(HipConv2d(in_channels, 64, 9, 1, 4), nn.Identity())
        self.res_blocks = nn.Sequential(*[ResidualBlock(64) for _ in range(n_residual_blocks)])
        self.conv2 = nn.Sequential(HipConv2d(64, 64, 3, 1, 1), HipBatchNorm2d(64))
        if (upscale_factor & (upscale_factor - 1)) == 0:
            r, stages = 2, int(math.log(upscale_factor, 2))
        elif upscale_factor % 3 == 0:
            r, stages = 3, int(math.log(upscale_factor, 3))
        else:
            r, stages = 1, 0
        stage = [HipConv2d(64, 64 * r * r, 3, 1, 1), HipBatchNorm2d(64 * r * r), _ShuffleRelu(r), nn.Identity()]
        self.upsampling = nn.Sequential(*(stage * stages))
        self.conv3 = nn.Sequential(HipConv2d(64, out_channels, 9, 1, 4), nn.Identity())

    def forward(self, x):
        out1 = self.conv1[0](ops.nhwc(x), 0.0)
        out2 = self.conv2[1](self.conv2[0](self.res_blocks(out1)))
        out = torch.add(out1, out2)
        up = self.upsampling
        for i in range(0, len(up), 4):
            out = up[i + 2](up[i + 1](up[i](out)))
        return torch.tanh(self.conv3[0](out))


class Discriminator(nn.Module):
    """srgan.py:123-155: the eight conv(+BN)+LeakyReLU(.2) blocks and the 512->1 conv, no attention."""

    _PLAN = [(64, 1, False), (64, 2, True), (128, 1, True), (128, 2, True),
             (256, 1, True), (256, 2, True), (512, 1, True), (512, 2, True)]

    def __init__(self, in_channels=3):
        super().__init__()
        layers, cin, self._blocks = [], in_channels, []
        for cout, stride, norm in self._PLAN:
            entry = (len(layers), len(layers) + 1 if norm else None)
            layers.append(HipConv2d(cin, cout, 3, stride, 1))
            if norm:
                layers.append(HipBatchNorm2d(cout))
            layers.append(nn.Identity())                 # slot of LeakyReLU(0.2): fused into conv / BN
            self._blocks.append(entry)
            cin = cout
        layers.append(HipConv2d(cin, 1, 3, 1, 1))
        self.model = nn.Sequential(*layers)

    def forward(self, img):
        x, m = ops.nhwc(img), self.model
        for conv_i, bn_i in self._blocks:
            x = m[conv_i](x, act_slope=0.2) if bn_i is None else m[bn_i](m[conv_i](x), act_slope=0.2)
        return m[len(m) - 1](x)


def train_step(G, D, Fx, opt_G, opt_D, lr_img, hr_img):
    """One iteration of srgan.py:335-365: loss_G = MSE(gen, hr) + 6e-3 * MSE(F(gen), F(hr)) + 1e-3 * MSE(D(gen), 1),
    Adam(G); loss_D = (MSE(D(hr), 1) + MSE(D(gen.detach()), 0)) / 2, Adam(D).  Returns the two logged scalars and the
    loss terms as 0-d device tensors (no host sync)."""
    mse = torch.nn.functional.mse_loss
    opt_G.zero_grad(set_to_none=True)
    gen_hr = G(lr_img)
    validity = D(gen_hr)
    valid, fake = torch.ones_like(validity), torch.zeros_like(validity)
    loss_gan = mse(validity, valid)
    with torch.no_grad():
        real_features = Fx(hr_img)
    content = mse(Fx(gen_hr), real_features)
    pixel = mse(gen_hr, hr_img)
    loss_G = pixel + 6e-3 * content + 1e-3 * loss_gan
    loss_G.backward()
    opt_G.step()
    ops.bump_weight_epoch()
    opt_D.zero_grad(set_to_none=True)                    # also drops what loss_G.backward() left in D (:355)
    loss_real = mse(D(hr_img), valid)
    loss_fake = mse(D(gen_hr.detach()), fake)
    loss_D = (loss_real + loss_fake) / 2
    loss_D.backward()
    opt_D.step()
    ops.bump_weight_epoch()
    return dict(loss_G=loss_G.detach(), loss_D=loss_D.detach(), pixel=pixel.detach(), content=content.detach(),
                loss_gan=loss_gan.detach(), loss_real=loss_real.detach(), loss_fake=loss_fake.detach())
