"""Input pipeline on the device: what the reference's dataset workers do per sample through
PIL -- RGB_TrainDatasetFromFolder.__getitem__, SRADSGAN/data/dataset.py:403-438 -- done per BATCH of uint8 HR tiles
already resident on the GPU: LR = bicubic down-sampling, BC = bicubic up-sampling of the LR tile back to HR size,
all three converted like torchvision's to_tensor (uint8 / 255 in float32).  The resampling is Pillow's 8-bit
algorithm bit for bit (csrc/resample.hip), so a batch built here equals the one the reference's DataLoader yields.
Weights per output coordinate are computed once per (in, out, filter) on the host and cached on the device."""
import ctypes

import numpy as np
import torch

from . import _hip

FILTERS = {'bilinear': 2, 'bicubic': 3}
_coeff_cache = {}


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _require_gpu_u8(t, what):
    if not t.is_cuda:
        raise RuntimeError('%s: the input pipeline runs on the MI355X HIP path only (got a %s tensor); there is no CPU '
                           'fallback' % (what, t.device.type))
    if t.dtype != torch.uint8 or t.dim() != 4:
        raise TypeError('%s expects a [N, H, W, C] uint8 tensor, got %s %s' % (what, t.dtype, tuple(t.shape)))


def _coeffs(in_size, out_size, filt, device):
    key = (in_size, out_size, filt, str(device))
    if key not in _coeff_cache:
        lib = _hip.lib()
        ksize = lib.srhip_resample_ksize(in_size, out_size, FILTERS[filt])
        bounds = np.empty(2 * out_size, np.int32)
        coeffs = np.empty(out_size * ksize, np.int32)
        _hip.check(lib.srhip_resample_coeffs(in_size, out_size, FILTERS[filt], ctypes.c_void_p(bounds.ctypes.data),
                                             ctypes.c_void_p(coeffs.ctypes.data)), 'resample_coeffs')
        _coeff_cache[key] = (ksize, torch.from_numpy(bounds).to(device), torch.from_numpy(coeffs).to(device))
    return _coeff_cache[key]


def resize_u8(img, out_h, out_w, filt='bicubic'):
    """img: [N, H, W, C] uint8 on the HIP device (PIL's interleaved layout) -> [N, out_h, out_w, C] uint8, equal to
    PIL.Image.resize((out_w, out_h), resample=filt) of every image (horizontal pass first, as Pillow does)."""
    _require_gpu_u8(img, 'resize_u8')
    if filt not in FILTERS:
        raise ValueError('filter must be one of %s' % sorted(FILTERS))
    img = img.contiguous()
    lib = _hip.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, h, w, c = img.shape
    if w != out_w:
        ksize, bounds, coeffs = _coeffs(w, out_w, filt, img.device)
        tmp = torch.empty(n, h, out_w, c, device=img.device, dtype=torch.uint8)
        _hip.check(lib.srhip_resample_pass_u8(_p(img), _p(tmp), _p(bounds), _p(coeffs), ksize, n, h, w, c, 1, out_w, st),
                   'resample_pass_u8')
        img, w = tmp, out_w
    if h != out_h:
        ksize, bounds, coeffs = _coeffs(h, out_h, filt, img.device)
        out = torch.empty(n, out_h, w, c, device=img.device, dtype=torch.uint8)
        _hip.check(lib.srhip_resample_pass_u8(_p(img), _p(out), _p(bounds), _p(coeffs), ksize, n, h, w, c, 0, out_h, st),
                   'resample_pass_u8')
        img = out
    return img


def to_tensor(img_u8):
    """[N, H, W, C] uint8 -> float32 [N, C, H, W] (channels_last memory: no transpose) = value / 255, torchvision's
    functional.to_tensor."""
    _require_gpu_u8(img_u8, 'to_tensor')
    img_u8 = img_u8.contiguous()
    out = torch.empty(img_u8.shape, device=img_u8.device, dtype=torch.float32)
    _hip.check(_hip.lib().srhip_u8_to_float(_p(img_u8), _p(out), img_u8.numel(),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'u8_to_float')
    return out.permute(0, 3, 1, 2)


def training_batch(hr_u8, scale):
    """hr_u8: [N, H, W, 3] uint8 HR tiles (already cropped to a multiple of `scale`).  Returns (lr, hr, bc) float
    tensors exactly as the reference's training DataLoader collates them (dataset.py:418-436)."""
    n, h, w, c = hr_u8.shape
    if h % scale or w % scale:
        raise ValueError('HR tile %dx%d is not a multiple of scale %d (calculate_valid_crop_size)' % (h, w, scale))
    lr_u8 = resize_u8(hr_u8, h // scale, w // scale, 'bicubic')
    bc_u8 = resize_u8(lr_u8, h, w, 'bicubic')
    return to_tensor(lr_u8), to_tensor(hr_u8), to_tensor(bc_u8)
