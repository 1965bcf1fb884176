"""Generator inference + validation metrics on the device: the arithmetic of the reference's
mfeNew_validate / validate loops (SRADSGAN/model/sradsgan.py:1258-1391, 1058-1194) without the per-image
device->host->PIL->numpy round trip.  Metrics follow the reference bit for bit where it is integer work
(uint8 quantisation with wrap, MSE, PSNR, ERGAS of utils/utils.py:923-962) and to fp64 roundoff for SSIM
(scikit-image 0.15 algorithm, parity unpinned: skimage is not vendored by the reference).  LPIPS is not
reproduced (its AlexNet weights are a download)."""
import ctypes
import math
import os

import torch

from . import _hip, ops


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def quantized_metrics(sr, hr, scale):
    """sr, hr: [N,3,H,W] float tensors on the HIP device (any memory format).  Returns a dict of float64
    tensors [N]: mse, psnr, ssim, ergas -- per image, exactly what the reference computes after ToPILImage."""
    ops._require_gpu(sr, 'quantized_metrics')
    ops._require_gpu(hr, 'quantized_metrics')
    if sr.shape != hr.shape:
        raise ValueError('quantized_metrics: shape mismatch %s vs %s' % (tuple(sr.shape), tuple(hr.shape)))
    sr, hr = ops.nhwc(sr.detach()), ops.nhwc(hr.detach())
    n, c, h, w = sr.shape
    lib = _hip.lib()
    nb = lib.srhip_metric_blocks()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    part = torch.empty(n, nb, 2, device=sr.device, dtype=torch.int64)
    _hip.check(lib.srhip_quant_sse(_p(sr), _p(hr), _p(part), n, c * h * w, st), 'quant_sse')
    spart = torch.empty(n, nb, device=sr.device, dtype=torch.float64)
    _hip.check(lib.srhip_ssim_u8(_p(sr), _p(hr), _p(spart), n, h, w, c, st), 'ssim_u8')
    if os.environ.get('SRHIP_METRIC_FINISH', '1') != '1':                  # A/B knob: the scalar tail as ~22 element-wise torch launches (rounds 2-4)
        sums = part.sum(1).to(torch.float64)
        count = float(c * h * w)
        mse = sums[:, 0] / count
        psnr = torch.where(mse > 0, 10.0 * torch.log10(255.0 ** 2 / mse.clamp_min(1e-300)), torch.full_like(mse, math.inf))
        mean_gt = sums[:, 1] / count
        ergas = 100.0 * torch.sqrt(mse / (mean_gt * mean_gt) / c) / scale
        ssim = spart.sum(1) / float((h - 6) * (w - 6) * c)
        return dict(mse=mse, psnr=psnr, ssim=ssim, ergas=ergas)
    out = torch.empty(4, n, device=sr.device, dtype=torch.float64)          # rows: mse, psnr, ssim, ergas (one launch for the scalar tail)
    _hip.check(lib.srhip_metric_finish(_p(part), _p(spart), _p(out), n, h, w, c, float(scale), st), 'metric_finish')
    return dict(mse=out[0], psnr=out[1], ssim=out[2], ergas=out[3])


@torch.no_grad()
def evaluate(generator, lr, hr, scale, bicubic=None):
    """One validation batch: recon = G(lr) (sradsgan.py:1305), metrics of recon vs hr and -- when the
    bicubic-upsampled input is given -- of bicubic vs hr (:1328-1331).  Returns per-image metric dicts."""
    recon = generator(lr)
    out = {'recon': recon, 'sr': quantized_metrics(recon, hr, scale)}
    if bicubic is not None:
        out['bicubic'] = quantized_metrics(bicubic, hr, scale)
    return out


class GraphedEvaluator:
    """evaluate() replayed from a captured hipGraph.  Generator inference at validation batch sizes is launch-bound
    (about 500 kernels of ~10 us for a batch of 16: the GPU waits for the Python launch loop), so the whole
    forward + metric pass is captured once per input shape and replayed; inputs are copied into the capture's static
    buffers, results live in static buffers that the next call overwrites (clone what must survive).
    Weights are read at replay time, so an evaluator stays valid across training steps as long as the parameter
    storage does not move (ParamArena guarantees that) and ops.repack_all() has run after the last update."""

    def __init__(self, generator, scale, warmup=2):
        self.generator, self.scale, self.warmup = generator, scale, warmup
        self._graphs = {}

    def _capture(self, key, lr, hr, bicubic):
        static = dict(lr=lr.clone(), hr=hr.clone(), bicubic=None if bicubic is None else bicubic.clone())
        side = torch.cuda.Stream(device=lr.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # library / allocator / packed-weight warm-up outside the capture
            for _ in range(self.warmup):
                evaluate(self.generator, static['lr'], static['hr'], self.scale, static['bicubic'])
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = evaluate(self.generator, static['lr'], static['hr'], self.scale, static['bicubic'])
        self._graphs[key] = (graph, static, out)

    def __call__(self, lr, hr, bicubic=None):
        ops._require_gpu(lr, 'GraphedEvaluator')
        key = (tuple(lr.shape), tuple(hr.shape), bicubic is not None, ops.get_conv_math())
        if key not in self._graphs:
            self._capture(key, lr, hr, bicubic)
        graph, static, out = self._graphs[key]
        static['lr'].copy_(lr)
        static['hr'].copy_(hr)
        if bicubic is not None:
            static['bicubic'].copy_(bicubic)
        graph.replay()
        return out


def format_results(mode, rlt):
    """The log line of the reference's PrintLogger.print_format_results (SRADSGAN/utils/logger.py:117-147), as a
    string: '<epoch:  3, iter:   1,200, time:1.25, lr:2.0e-04> dataset: AID key: 1.23e-04 ...' -- '{:.2e}' per value
    in 'train' mode, '{:.4e}' in 'val' mode.  `rlt` must carry epoch, iters, time, model (and optionally lr); it is
    not modified."""
    rlt = dict(rlt)
    epoch, iters, time_, model = rlt.pop('epoch'), rlt.pop('iters'), rlt.pop('time'), rlt.pop('model')
    if 'lr' in rlt:
        message = '<epoch:{:3d}, iter:{:8,d}, time:{:.2f}, lr:{:.1e}> '.format(epoch, iters, time_, rlt.pop('lr'))
    else:
        message = '<epoch:{:3d}, iter:{:8,d}, time:{:.2f}> '.format(epoch, iters, time_)
    message += '{:s}: {:s} '.format('dataset', model)
    for label, value in rlt.items():
        if mode == 'train':
            message += '{:s}: {:.2e} '.format(label, value)
        elif mode == 'val':
            message += '{:s}: {:.4e} '.format(label, value)
    return message


def append_log(path, mode, rlt):
    """Appends format_results(mode, rlt) to loss_log.txt / val_log.txt like the reference (logger.py:142-147)."""
    line = format_results(mode, rlt)
    with open(path, 'a') as f:
        f.write(line + '\n')
    return line
