"""TEST INFRASTRUCTURE (oracle) -- not product code.  CPU restatement in stock torch ops of the reference's SRGAN
sibling: SRADSGAN/model/srgan.py:57-155 (ResidualBlock, GeneratorResNet, Discriminator) and one iteration of its
training loop (:335-365: MSE pixel + 6e-3 * MSE VGG-feature + 1e-3 * LSGAN generator loss; LSGAN discriminator loss).
Pinned by tests/golden/srgan_x{2,3,4}.npz, recorded from the reference's own classes (oracle/make_golden_srgan.py).
The VGG stand-in is oracle/sradsgan_ref.FeatureExtractor (same structure as srgan.py:44-55; weight values unpinned)."""
import math

import torch
import torch.nn as nn


class ResidualBlock(nn.Module):                               # srgan.py:57-70
    def __init__(self, in_features):
        super().__init__()
        self.conv_block = nn.Sequential(nn.Conv2d(in_features, in_features, 3, 1, 1), nn.BatchNorm2d(in_features),
                                        nn.ReLU(),
                                        nn.Conv2d(in_features, in_features, 3, 1, 1), nn.BatchNorm2d(in_features))

    def forward(self, x):
        return x + self.conv_block(x)


class GeneratorResNet(nn.Module):                             # srgan.py:72-121
    def __init__(self, in_channels=3, out_channels=3, n_residual_blocks=16, upscale_factor=3):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, 64, 9, 1, 4), nn.ReLU())
        self.res_blocks = nn.Sequential(*[ResidualBlock(64) for _ in range(n_residual_blocks)])
        self.conv2 = nn.Sequential(nn.Conv2d(64, 64, 3, 1, 1), nn.BatchNorm2d(64))
        if (upscale_factor & (upscale_factor - 1)) == 0:
            r, stages = 2, int(math.log(upscale_factor, 2))
        elif upscale_factor % 3 == 0:
            r, stages = 3, int(math.log(upscale_factor, 3))
        else:
            r, stages = 1, 0
        # the same four module objects once per stage: conv AND BatchNorm are tied across stages (:97-104)
        stage = [nn.Conv2d(64, 64 * r * r, 3, 1, 1), nn.BatchNorm2d(64 * r * r), nn.PixelShuffle(r), nn.ReLU()]
        self.upsampling = nn.Sequential(*(stage * stages))
        self.conv3 = nn.Sequential(nn.Conv2d(64, out_channels, 9, 1, 4), nn.Tanh())

    def forward(self, x):
        out1 = self.conv1(x)
        out2 = self.conv2(self.res_blocks(out1))
        return self.conv3(self.upsampling(torch.add(out1, out2)))


class Discriminator(nn.Module):                               # srgan.py:123-155
    def __init__(self, in_channels=3):
        super().__init__()
        layers, cin = [], in_channels
        for cout, stride, norm in [(64, 1, False), (64, 2, True), (128, 1, True), (128, 2, True),
                                   (256, 1, True), (256, 2, True), (512, 1, True), (512, 2, True)]:
            layers.append(nn.Conv2d(cin, cout, 3, stride, 1))
            if norm:
                layers.append(nn.BatchNorm2d(cout))
            layers.append(nn.LeakyReLU(0.2))
            cin = cout
        layers.append(nn.Conv2d(cin, 1, 3, 1, 1))
        self.model = nn.Sequential(*layers)

    def forward(self, img):
        return self.model(img)


def train_step(G, D, Fx, opt_G, opt_D, lr_img, hr_img):
    """One iteration of srgan.py:335-365.  `valid` / `fake` are the all-ones / all-zeros patch targets (:247-248,
    :280-281; their shape is D's output shape when crop_size is a multiple of 16)."""
    mse = nn.MSELoss()
    opt_G.zero_grad()
    gen_hr = G(lr_img)
    validity = D(gen_hr)
    valid, fake = torch.ones_like(validity), torch.zeros_like(validity)
    loss_gan = mse(validity, valid)
    content = mse(Fx(gen_hr), Fx(hr_img).detach())
    pixel = mse(gen_hr, hr_img)
    loss_G = pixel + 6e-3 * content + 1e-3 * loss_gan
    loss_G.backward()
    opt_G.step()
    opt_D.zero_grad()
    loss_real = mse(D(hr_img), valid)
    loss_fake = mse(D(gen_hr.detach()), fake)
    loss_D = (loss_real + loss_fake) / 2
    loss_D.backward()
    opt_D.step()
    return dict(loss_G=loss_G.item(), loss_D=loss_D.item(), pixel=pixel.item(), content=content.item(),
                loss_gan=loss_gan.item(), loss_real=loss_real.item(), loss_fake=loss_fake.item())
