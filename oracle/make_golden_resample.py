"""Generates tests/golden/pil_resample.npz with Pillow itself (run in the build container; Pillow 12.2 there).
Inputs are small deterministic uint8 images; outputs are Image.resize results for the sizes/filters the reference's
pipeline uses (x2/x3/x4 down with BICUBIC and BILINEAR, the bicubic up-sampling back) plus ragged sizes."""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def det_u8(tag, shape):
    rng = np.random.RandomState(sum(ord(c) for c in tag) * 7919 % (2 ** 31))
    base = rng.randint(0, 256, size=shape).astype(np.uint8)
    # mix smooth structure and noise so that overshoot (clipping) and flat regions both occur
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]]
    smooth = (127.5 + 127.5 * np.sin(xx / 5.0) * np.cos(yy / 7.0))[..., None]
    return np.where(rng.rand(*shape) < 0.5, base, smooth.astype(np.uint8)).astype(np.uint8)


CASES = [('t216', (216, 216, 3), (54, 54), 'bicubic'), ('t216', (216, 216, 3), (72, 72), 'bicubic'),
         ('t216', (216, 216, 3), (108, 108), 'bicubic'), ('t216', (216, 216, 3), (54, 54), 'bilinear'),
         ('t54', (54, 54, 3), (216, 216), 'bicubic'), ('rag', (37, 61, 3), (11, 23), 'bicubic'),
         ('rag', (37, 61, 3), (80, 45), 'bicubic'), ('rag', (37, 61, 3), (37, 20), 'bilinear'), ('gray', (40, 33, 1), (10, 11), 'bicubic')]


def main():
    out = {}
    for i, (tag, shape, (oh, ow), filt) in enumerate(CASES):
        img = det_u8(tag, shape)
        pil = Image.fromarray(img if shape[2] == 3 else img[..., 0])
        res = np.asarray(pil.resize((ow, oh), Image.BICUBIC if filt == 'bicubic' else Image.BILINEAR))
        if res.ndim == 2:
            res = res[..., None]
        out['img_' + tag] = img                       # each input stored once
        out['tag%d' % i] = np.array(tag)
        out['out%d' % i] = res
        out['filter%d' % i] = np.array(filt)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'pil_resample.npz'), **out)
    print('wrote', len(CASES), 'cases')


if __name__ == '__main__':
    sys.exit(main())
