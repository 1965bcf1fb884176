"""CPU oracle for the SRADSGAN generator/discriminator training step.

TEST INFRASTRUCTURE ONLY.  This file is a restatement, in stock PyTorch CPU ops,
of the arithmetic of the reference hot path (reference/SRADSGAN/model/sradsgan.py
lines 35-508 and the inner training loop 818-892).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it; the
product package `sradsgan_amd/` never does.

Parity status: PINNED.  `oracle/make_golden.py` imports the reference modules in
the build container (the reference has no tests or golden vectors of its own,
SURVEY.md section 4) and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks every function below against those vectors.  Two pieces are NOT pinned and
say so where they are defined: the VGG19 weights (pretrained file not available
offline; structure only) and SSIM (scikit-image 0.15 is not vendored in the
reference).

Module classes keep the reference's constructor signatures and state_dict keys so
the same weight dictionaries drive the reference, this oracle and the HIP path.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------- #
# local attention (reference sradsgan.py:101-151, base_networks.py:366-457)
# --------------------------------------------------------------------------- #


class CLAM(nn.Module):
    """sigmoid(MLP(avgpool x) + MLP(maxpool x)) * x  -- sradsgan.py:101-127."""

    def __init__(self, in_planes, ratio=16, pool_mode='Avg|Max'):
        super().__init__()
        self.pool_mode = pool_mode
        self.fc1 = nn.Conv2d(in_planes, in_planes // ratio, 1, bias=False)
        self.fc2 = nn.Conv2d(in_planes // ratio, in_planes, 1, bias=False)

    def _mlp(self, v):
        return self.fc2(F.relu(self.fc1(v)))

    def forward(self, x):
        logits = 0
        if 'Avg' in self.pool_mode:
            logits = logits + self._mlp(F.adaptive_avg_pool2d(x, 1))
        if 'Max' in self.pool_mode:
            logits = logits + self._mlp(F.adaptive_max_pool2d(x, 1))
        return torch.sigmoid(logits) * x


class SLAM(nn.Module):
    """sigmoid(conv7x7([mean_c x, max_c x])) * x  -- sradsgan.py:129-151."""

    def __init__(self, kernel_size=7, pool_mode='Avg|Max'):
        super().__init__()
        assert kernel_size in (3, 7)
        self.pool_mode = pool_mode
        cin = 2 if pool_mode == 'Avg|Max' else 1
        self.conv1 = nn.Conv2d(cin, 1, kernel_size, padding=kernel_size // 2, bias=False)

    def forward(self, x):
        maps = []
        if 'Avg' in self.pool_mode:
            maps.append(x.mean(dim=1, keepdim=True))
        if 'Max' in self.pool_mode:
            maps.append(x.max(dim=1, keepdim=True)[0])
        return torch.sigmoid(self.conv1(torch.cat(maps, dim=1))) * x


# the discriminator's attention pair is arithmetically the same pair
# (base_networks.py:366-403 and 424-457)
ChannelAttention = CLAM
SpatialAttention = SLAM

# --------------------------------------------------------------------------- #
# global attention (sradsgan.py:153-213)
# --------------------------------------------------------------------------- #


class SGAM(nn.Module):
    """Position self-attention, no 1/sqrt(d) scaling -- sradsgan.py:153-176."""

    def __init__(self, in_dim):
        super().__init__()
        self.query_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.key_conv = nn.Conv2d(in_dim, in_dim // 8, 1)
        self.value_conv = nn.Conv2d(in_dim, in_dim, 1)
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        b, c, h, w = x.shape
        q = self.query_conv(x).reshape(b, -1, h * w)          # [b, c/8, n]
        k = self.key_conv(x).reshape(b, -1, h * w)
        v = self.value_conv(x).reshape(b, -1, h * w)          # [b, c, n]
        att = torch.softmax(q.transpose(1, 2) @ k, dim=-1)    # [b, n, n], rows = query pixel
        out = (v @ att.transpose(1, 2)).reshape(b, c, h, w)
        return self.gamma * out + x


class CGAM(nn.Module):
    """Channel self-attention softmax(rowmax(E) - E) -- sradsgan.py:178-213 (light=False)."""

    def __init__(self, in_dim, light=False):
        super().__init__()
        if light:
            raise NotImplementedError('light CGAM is never built by the reference hot path')
        self.gamma = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        b, c, h, w = x.shape
        xf = x.reshape(b, c, h * w)
        energy = xf @ xf.transpose(1, 2)
        energy = energy.max(dim=-1, keepdim=True)[0] - energy
        out = (torch.softmax(energy, dim=-1) @ xf).reshape(b, c, h, w)
        return self.gamma * out + x


# --------------------------------------------------------------------------- #
# residual attention block / group (sradsgan.py:215-324)
# --------------------------------------------------------------------------- #


def _attention_tail(mod, la_mode, pool_mode, planes, addconv):
    if 'CA' in la_mode:
        mod.ca = CLAM(planes, pool_mode=pool_mode)
    if 'SA' in la_mode:
        mod.sa = SLAM(kernel_size=7, pool_mode=pool_mode)
    if '|' in la_mode:
        mod.conv = nn.Conv2d(planes * 2, planes, 1)
    if '-' in la_mode and addconv:
        mod.conv = nn.Conv2d(planes, planes, 1)
    if la_mode == '':
        mod.last_conv = nn.Conv2d(64, 64, 1)


def _apply_attention_tail(mod, out):
    mode = mod.la_mode
    if mode == 'CA':
        return mod.ca(out)
    if mode == 'SA':
        return mod.sa(out)
    if mode in ('CA-SA', 'SA-CA'):
        first, second = (mod.ca, mod.sa) if mode == 'CA-SA' else (mod.sa, mod.ca)
        out = second(first(out))
        return mod.conv(out) if mod.addconv else out
    if mode == 'CA|SA':
        return mod.conv(torch.cat([mod.ca(out), mod.sa(out)], dim=1))
    if mode == '':
        return mod.last_conv(out)
    return out


class RAB(nn.Module):
    """conv3x3(c->4c) LReLU(.2) conv3x3(4c->c) CLAM SLAM conv1x1 (+x) -- sradsgan.py:215-275."""

    def __init__(self, inplanes, planes, kernel_size=3, stride=1, padding=1, bias=True, dilation=1,
                 act_type='lrelu', la_mode='CA-SA', pool_mode='Avg|Max', addconv=True):
        super().__init__()
        if act_type != 'lrelu':
            raise NotImplementedError('reference hot path only builds act_type="lrelu"')
        self.conv1 = nn.Conv2d(inplanes, 4 * planes, kernel_size, stride, padding, dilation, bias=bias)
        self.conv2 = nn.Conv2d(4 * planes, planes, kernel_size, stride, padding, dilation, bias=bias)
        self.la_mode, self.addconv = la_mode, addconv
        _attention_tail(self, la_mode, pool_mode, planes, addconv)

    def forward(self, x):
        out = self.conv2(F.leaky_relu(self.conv1(x), 0.2))
        return _apply_attention_tail(self, out) + x


class ResGroup(nn.Module):
    """n_blocks x RAB, then the same attention tail, (+x) -- sradsgan.py:277-324."""

    def __init__(self, block, n_blocks=5, nc=64, kernel_size=3, stride=1, bias=True, padding=1,
                 act_type='lrelu', mode='CNA', rla_mode='CA-SA', bla_mode='CA-SA', pool_mode='Avg|Max',
                 addconv=True):
        super().__init__()
        self.RG = nn.Sequential(*[
            block(nc, nc, kernel_size=kernel_size, bias=bias, stride=stride, padding=padding,
                  act_type='lrelu', la_mode=bla_mode, pool_mode=pool_mode, addconv=addconv)
            for _ in range(n_blocks)])
        self.la_mode, self.addconv = rla_mode, addconv
        _attention_tail(self, rla_mode, pool_mode, nc, addconv)

    def forward(self, x):
        return _apply_attention_tail(self, self.RG(x)) + x


class MSB(nn.Module):
    """three branches -> cat(192) -> 1x1 -> LReLU(.01) -- sradsgan.py:326-345."""

    def __init__(self, inplanes, planes):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, 1, 1)
        self.conv2 = nn.Sequential(nn.Conv2d(inplanes, planes, 1), nn.Conv2d(planes, planes, 3, 1, 1))
        self.conv3 = nn.Conv2d(inplanes, planes, 1)
        self.conv = nn.Conv2d(planes * 3, planes, 1)

    def forward(self, x):
        cat = torch.cat([self.conv1(x), self.conv2(x), self.conv3(x)], dim=1)
        return F.leaky_relu(self.conv(cat), 0.01)


class GAB_UP(nn.Module):
    """CGAM, SGAM, 1x1, then k x [conv3x3 -> PixelShuffle(r) -> LReLU(.01)] with ONE shared conv
    -- sradsgan.py:365-418.  state_dict lists the tied conv under every stage index (0, 3, ...)."""

    def __init__(self, ga_mode='CA-SA', addconv=True, upscale_factor=4):
        super().__init__()
        self.ga_mode, self.addconv = ga_mode, addconv
        if 'CA' in ga_mode:
            self.ca = CGAM(64)
        if 'SA' in ga_mode:
            self.sa = SGAM(64)
        if '-' in ga_mode and addconv:
            self.conv = nn.Conv2d(64, 64, 1)
        if '|' in ga_mode:
            self.conv = nn.Conv2d(128, 64, 1)
        if upscale_factor & (upscale_factor - 1) == 0:
            r, stages = 2, int(math.log(upscale_factor, 2))
        elif upscale_factor % 3 == 0:
            r, stages = 3, int(math.log(upscale_factor, 3))
        else:
            r, stages = 0, 0
        stage = [nn.Conv2d(64, 64 * r * r, 3, 1, 1), nn.PixelShuffle(r), nn.LeakyReLU()] if stages else []
        self.upsampling = nn.Sequential(*(stage * stages))     # same module objects repeated => tied

    def forward(self, x):
        m = self.ga_mode
        out = x
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
    """sradsgan.py:420-468: MSB + head, dense-sampled residual groups, GAB_UP, tail conv."""

    def __init__(self, buildingblock, in_channels=3, out_channels=3, n_residual_blocks=12, n_basic_blocks=3,
                 rla_mode='CA-SA', bla_mode='CA-SA', ga_mode='CA-SA', pool_mode='Avg|Max', addconv=True,
                 upscale_factor=4):
        super().__init__()
        self.conv1 = nn.Sequential(nn.Conv2d(in_channels, 64, 3, 1, 1), nn.LeakyReLU())
        self.res_groups = nn.Sequential(*[
            buildingblock(RAB, n_blocks=n_basic_blocks, nc=64, kernel_size=3, stride=1, padding=1,
                          act_type='lrelu', mode='CNA', rla_mode=rla_mode, bla_mode=bla_mode,
                          pool_mode=pool_mode, addconv=addconv)
            for _ in range(n_residual_blocks)])
        self.GAB_UP = GAB_UP(ga_mode=ga_mode, addconv=addconv, upscale_factor=upscale_factor)
        self.MSB = MSB(inplanes=in_channels, planes=64)
        self.conv3 = nn.Sequential(nn.Conv2d(64, out_channels, 3, 1, 1))

    def forward(self, x):
        out = self.conv1(x)
        bus = self.MSB(x) + out
        for group in self.res_groups:
            out = group(out)
            bus = bus + out                      # stratified dense sampling, sradsgan.py:455-460
        return self.conv3(self.GAB_UP(bus))


class Discriminator(nn.Module):
    """sradsgan.py:470-508.  The CAM/PAM tail is never built (`layers == 8` compares a list)."""

    _PLAN = [(64, 1, False), (64, 2, True), (128, 1, True), (128, 2, True),
             (256, 1, True), (256, 2, True), (512, 1, True), (512, 2, True)]

    def __init__(self, in_channels=3, attention=True):
        super().__init__()
        layers, cin = [], in_channels
        for idx, (cout, stride, norm) in enumerate(self._PLAN, start=1):
            layers.append(nn.Conv2d(cin, cout, 3, stride, 1))
            if norm:
                layers.append(nn.BatchNorm2d(cout))
            layers.append(nn.LeakyReLU(0.2))
            if attention and idx == 6:
                layers += [ChannelAttention(256), SpatialAttention()]
            cin = cout
        layers.append(nn.Conv2d(cin, 1, 3, 1, 1))
        self.model = nn.Sequential(*layers)

    def forward(self, img):
        return self.model(img)


class FeatureExtractor(nn.Module):
    """Layer structure of torchvision vgg19().features[:12] (sradsgan.py:88-99); no input
    normalisation.  PARITY UNPINNED for the weight VALUES: the pretrained file cannot be
    fetched offline, so weights are whatever the caller loads (same arithmetic, any weights)."""

    def __init__(self):
        super().__init__()
        cfg = [(3, 64), 'R', (64, 64), 'R', 'P', (64, 128), 'R', (128, 128), 'R', 'P', (128, 256), 'R']
        seq = []
        for item in cfg:
            if item == 'R':
                seq.append(nn.ReLU())
            elif item == 'P':
                seq.append(nn.MaxPool2d(2, 2))
            else:
                seq.append(nn.Conv2d(item[0], item[1], 3, 1, 1))
        self.feature_extractor = nn.Sequential(*seq)

    def forward(self, img):
        return self.feature_extractor(img)


class GANLoss(nn.Module):
    """sradsgan.py:35-67, 'wgan-gp' flavour only: -mean(x) for a real target, +mean(x) otherwise."""

    def __init__(self, gan_type, real_label_val=1.0, fake_label_val=0.0):
        super().__init__()
        self.gan_type = gan_type.lower()
        if self.gan_type != 'wgan-gp':
            raise NotImplementedError('GAN type [{:s}] is not found'.format(self.gan_type))

    def forward(self, input, target_is_real):
        return -input.mean() if target_is_real else input.mean()


# --------------------------------------------------------------------------- #
# init, losses, one training iteration (utils/utils.py:97-114, sradsgan.py:595-641, 818-892)
# --------------------------------------------------------------------------- #


def weights_init_normal(m, mean=0.0, std=0.02):
    """utils/utils.py:97-114: conv/linear W~N(0,.02) b=0; BatchNorm W~N(1,.02) b=0."""
    name = m.__class__.__name__
    if 'Linear' in name or 'Conv2d' in name or 'ConvTranspose2d' in name:
        m.weight.data.normal_(mean, std)
        if m.bias is not None:
            m.bias.data.zero_()
    elif 'BatchNorm' in name:
        m.weight.data.normal_(1.0, 0.02)
        if m.bias is not None:
            m.bias.data.zero_()


def gradient_penalty(discriminator, real, fake, alpha, grad_penalty_Lp_norm='L2', penalty_type='LS'):
    """sradsgan.py:595-641 with `alpha` ([B,1,1,1]) injected instead of np.random.
    NB the norm is over dim=1 (channels) => per-pixel, and backward() is called inside."""
    interp = (alpha * real + (1 - alpha) * fake).requires_grad_(True)
    d_out = discriminator(interp)
    grads = torch.autograd.grad(d_out, interp, torch.ones_like(d_out), create_graph=True, retain_graph=True)[0]
    if grad_penalty_Lp_norm == 'Linf':
        norm = grads.abs().max(dim=1)[0]
    elif grad_penalty_Lp_norm == 'L1':
        norm = grads.norm(1, 1)
    else:
        norm = grads.norm(2, 1)
    cons = (norm - 1).pow(2) if penalty_type == 'LS' else F.relu(norm - 1)
    gp = cons.mean()
    gp.backward(retain_graph=True)
    return gp


def train_step(G, D, Fx, opt_G, opt_D, lr_img, hr_img, alpha, weight_content=1e-2, weight_gan=1e-3,
               lambda_gp=10.0, clip_value=0.01, use_gp=True):
    """One iteration of sradsgan.py:829-892 (non-relativistic, L1 content, wgan-gp).
    Returns the scalars the reference logs plus the three G-loss terms."""
    crit = nn.L1Loss()
    gan = GANLoss('wgan-gp')
    # ---- generator ----
    opt_G.zero_grad()
    gen_hr = G(lr_img)
    pixel = crit(gen_hr, hr_img)
    content = crit(Fx(gen_hr), Fx(hr_img).detach())
    loss_gan = gan(D(gen_hr), True)
    loss_G = pixel + weight_content * content + weight_gan * loss_gan
    loss_G.backward()
    opt_G.step()
    # ---- discriminator ----
    opt_D.zero_grad()
    loss_D = gan(D(hr_img), True) + gan(D(gen_hr.detach()), False)
    gp = torch.zeros(())
    if use_gp:
        gp = gradient_penalty(D, hr_img.detach(), gen_hr.detach(), alpha)
        loss_D = loss_D + lambda_gp * gp
    loss_D.backward()
    opt_D.step()
    with torch.no_grad():
        for p in D.parameters():
            p.clamp_(-clip_value, clip_value)
    return dict(loss_G=loss_G.item(), loss_D=loss_D.item(), pixel=pixel.item(), content=content.item(),
                loss_gan=loss_gan.item(), gp=float(gp.detach()), gen_hr=gen_hr.detach())


# --------------------------------------------------------------------------- #
# metric quantisation (sradsgan.py:1314-1325; utils/utils.py:923-962)
# --------------------------------------------------------------------------- #


def to_uint8_hwc(img_chw):
    """torchvision ToPILImage on a float CHW tensor: mul(255).byte() -- truncation toward zero and
    wrap modulo 256 out of range (documented in the vendored copy reference model/util.py:63-123)."""
    v = img_chw.detach().cpu().to(torch.float32).mul(255)
    v = torch.trunc(v).to(torch.int64).remainder(256).to(torch.uint8)
    return v.permute(1, 2, 0).contiguous().numpy()


def mse_u8(a, b):
    return float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))


def psnr_u8(a, b):
    """skimage compare_psnr on uint8 == utils/utils.py:923-930."""
    m = mse_u8(a, b)
    return float('inf') if m == 0 else 10.0 * math.log10(255.0 ** 2 / m)


def ergas2(img1, img2, scale=4):
    """utils/utils.py:954-962 (compare_ergas2), img1 = ground truth."""
    mean2 = np.mean(img1, dtype=np.float64) ** 2
    return 100.0 * math.sqrt(mse_u8(img1, img2) / mean2 / img1.shape[2]) / scale


def ssim_u8(a, b):
    """scikit-image 0.15 compare_ssim(multichannel=True) restated: 7x7 uniform window, K1=.01,
    K2=.03, sample covariance, data_range 255, mean over the valid interior and channels.
    PARITY UNPINNED: skimage is absent here and not vendored by the reference."""
    from scipy.ndimage import uniform_filter
    win, L = 7, 255.0
    c1, c2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    npix = win * win
    cov_norm = npix / (npix - 1.0)
    vals = []
    for ch in range(a.shape[2]):
        x = a[:, :, ch].astype(np.float64)
        y = b[:, :, ch].astype(np.float64)
        ux, uy = uniform_filter(x, win), uniform_filter(y, win)
        uxx, uyy, uxy = uniform_filter(x * x, win), uniform_filter(y * y, win), uniform_filter(x * y, win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        pad = (win - 1) // 2
        vals.append(s[pad:-pad, pad:-pad].mean())
    return float(np.mean(vals))


# --------------------------------------------------------------------------- #
# deterministic, RNG-library-independent tensor filler shared by goldens and tests
# --------------------------------------------------------------------------- #


def det_fill(name, shape, scale=1.0, offset=0.0):
    """Uniform(-scale, scale)+offset values from a splitmix-style integer hash of (name, index);
    pure integer numpy so it is reproducible on any box / numpy / torch version."""
    import zlib
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode()) * 2654435761 % (1 << 32))
    with np.errstate(over='ignore'):
        z = (np.arange(n, dtype=np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15) + (seed << np.uint64(32))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) / float(1 << 53)          # [0,1)
    return torch.from_numpy(((u * 2 - 1) * scale + offset).astype(np.float32).reshape(shape))


def det_init_(module, prefix='', gamma=0.5):
    """Fill every parameter of `module` deterministically by state_dict key (tied params once):
    conv weights U(+-0.02*sqrt3) (std .02), biases U(+-0.01) so bias paths are live, BN weights ~1,
    attention gammas = `gamma` so CGAM/SGAM are live.  VGG convs use a He-style scale."""
    seen = set()
    mods = dict(module.named_modules())
    with torch.no_grad():
        for key, p in module.named_parameters(remove_duplicate=False):
            if id(p) in seen:
                continue
            seen.add(id(p))
            owner = mods[key.rsplit('.', 1)[0]] if '.' in key else module
            leaf = key.rsplit('.', 1)[-1]
            full = prefix + key
            if leaf == 'gamma':
                p.fill_(gamma)
            elif isinstance(owner, nn.BatchNorm2d):
                p.copy_(det_fill(full, p.shape, 0.05, 1.0 if leaf == 'weight' else 0.0))
            elif leaf == 'bias':
                p.copy_(det_fill(full, p.shape, 0.01))
            elif 'feature_extractor' in full:
                fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                p.copy_(det_fill(full, p.shape, math.sqrt(6.0 / fan_in)))
            else:
                p.copy_(det_fill(full, p.shape, 0.02 * math.sqrt(3.0)))
    return module


def digest(t, full_max=4096, nsample=2048):
    """Fixture-size reducer shared by make_golden.py and the tests: small tensors whole, large ones as
    an evenly strided sample followed by [sum, l2 norm] (float64 -> float32)."""
    a = t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    a = a.astype(np.float32).ravel()
    if a.size <= full_max:
        return a
    stride = a.size // nsample
    a64 = a.astype(np.float64)
    return np.concatenate([a[::stride][:nsample], np.array([a64.sum(), np.sqrt((a64 ** 2).sum())], dtype=np.float32)])
