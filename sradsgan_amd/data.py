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


# --------------------------------------------------------------------------------------------- #
# host side of the input pipeline: file listing + decode (data/dataset.py:88-101, 386-441) and the loader
# (sradsgan.py:643-656: shuffle, drop_last).  Decode stays on the host (Pillow, worker threads); everything after the
# uint8 HR tile -- LR / bicubic synthesis, to_tensor -- is `training_batch` / `test_batch` on the device.
# --------------------------------------------------------------------------------------------- #

IMG_EXTENSIONS = ('.png', '.jpg', '.jpeg', '.bmp', '.tif')                       # dataset.py:88-89 (case-sensitive there too)


def is_image_file(filename):
    return any(filename.endswith(ext) for ext in IMG_EXTENSIONS)


def calculate_valid_crop_size(crop_size, scale_factor):                         # dataset.py:100-101
    return crop_size - (crop_size % scale_factor)


def rgb_train_dirs(data_dir, datasets):
    """The class directories get_RGB_trainDataset walks (data/data.py:295-312): `SECOND` is one flat folder, AID / DOTA /
    LoveDA / RSSCN7_2800 contribute every sub-directory (os.listdir order, like the reference); other names are ignored
    there too."""
    import os
    dirs = []
    for name in datasets:
        if name == 'SECOND':
            dirs.append(os.path.join(data_dir, name))
        elif name in ('AID', 'DOTA', 'LoveDA', 'RSSCN7_2800'):
            root = os.path.join(data_dir, name)
            dirs.extend(os.path.join(root, d) for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
    return dirs


def rgb_test_dirs(data_dir, datasets):
    """get_RGB_testDataset's directory list (data/data.py:314-325): the sorted class folders of UCMerced_LandUse when that
    is the first entry, otherwise the entries themselves are the directories."""
    import os
    if datasets and datasets[0] == 'UCMerced_LandUse':
        root = os.path.join(data_dir, datasets[0])
        return [os.path.join(root, d) for d in sorted(os.listdir(root)) if os.path.isdir(os.path.join(root, d))]
    return list(datasets)


class TileFolder:
    """The file side of RGB_TrainDatasetFromFolder / RGB_TestDatasetFromFolder (dataset.py:386-441): every image file of
    `image_dirs`, each directory's names sorted, directories in the order given; item i = (uint8 [H, W, 3] RGB tensor,
    file name).  The reference applies no crop or flip to training tiles (the augmentation flags are accepted and
    ignored): tiles must already have the batch's common size."""

    def __init__(self, image_dirs, is_gray=False, crop_size=216, scale_factor=3, **ignored_augmentation_flags):
        import os
        if is_gray:
            raise NotImplementedError('TileFolder: RGB tiles only (is_gray=False is what the SRADSGAN trainer passes)')
        self.image_filenames = []
        for d in image_dirs:
            self.image_filenames.extend(os.path.join(d, x) for x in sorted(os.listdir(d)) if is_image_file(x))
        self.crop_size = calculate_valid_crop_size(crop_size, scale_factor)
        self.scale_factor = scale_factor

    def __len__(self):
        return len(self.image_filenames)

    def __getitem__(self, index):
        import numpy as np
        from PIL import Image
        fn = self.image_filenames[index]
        with Image.open(fn) as im:
            arr = np.asarray(im.convert('RGB'), dtype=np.uint8)                   # dataset.py:92-97
        return torch.from_numpy(arr.copy()), fn


class DevicePrefetcher:
    """DataLoader(shuffle, drop_last=True) (sradsgan.py:652, 656) for a dataset of uint8 tiles, delivering batches that
    are already resident on the GPU: worker THREADS decode batch k+1 (Pillow releases the GIL) into one of two pinned
    staging buffers while the device works on batch k, and the host-to-device copy runs on its own HIP stream; the
    consumer's stream waits on the copy's event only.  Yields (uint8 [B, H, W, 3] device tensor, [file names]).
    rank / world: this process's shard under data parallelism -- every rank draws the same permutation (same seed) and
    keeps every world-th index, all ranks the same number of batches."""

    def __init__(self, dataset, batch_size, device, shuffle=False, drop_last=True, num_workers=4, seed=None, rank=0, world=1):
        if torch.device(device).type != 'cuda' or not torch.cuda.is_available():
            raise RuntimeError('DevicePrefetcher stages through pinned memory to a GPU: needs a cuda device')
        self.dataset, self.batch_size, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.drop_last, self.num_workers = shuffle, drop_last, max(1, int(num_workers))
        self.rank, self.world = int(rank), int(world)
        if not 0 <= self.rank < self.world:
            raise ValueError('DevicePrefetcher: rank %d outside world %d' % (self.rank, self.world))
        self.generator = torch.Generator()
        if seed is None and self.world > 1:
            seed = 0                                  # ranks must draw the same permutation to get disjoint shards
        if seed is not None:
            self.generator.manual_seed(seed)

    def __len__(self):
        n = len(self.dataset) // self.world
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        from concurrent.futures import ThreadPoolExecutor
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        if self.world > 1:                            # data parallel: disjoint shards of the same permutation
            from .dp import shard_indices
            order = shard_indices(order, self.rank, self.world)
            n = len(order)
        batches = [order[i:i + self.batch_size] for i in range(0, n, self.batch_size)]
        if self.drop_last and batches and len(batches[-1]) < self.batch_size:
            batches.pop()
        if not batches:
            return
        copy_stream = torch.cuda.Stream(device=self.device)
        staging, staged_free = [None, None], [None, None]            # pinned buffers and "copy out of it finished" events

        def decode(indices, slot):
            items = list(pool.map(self.dataset.__getitem__, indices))
            shape = (len(items),) + tuple(items[0][0].shape)
            if staged_free[slot] is not None:
                staged_free[slot].synchronize()                       # the previous copy out of this buffer is done
            if staging[slot] is None or tuple(staging[slot].shape[1:]) != shape[1:] or staging[slot].shape[0] < shape[0]:
                staging[slot] = torch.empty(shape, dtype=torch.uint8).pin_memory()
            buf = staging[slot][:shape[0]]
            for j, (t, _) in enumerate(items):
                if tuple(t.shape) != shape[1:]:
                    raise ValueError('DevicePrefetcher: tiles of one batch differ in size (%s vs %s): %s'
                                     % (tuple(t.shape), shape[1:], items[j][1]))
                buf[j].copy_(t)
            return buf, [fn for _, fn in items]

        with ThreadPoolExecutor(self.num_workers + 1) as outer, ThreadPoolExecutor(self.num_workers) as pool:
            pending = outer.submit(decode, batches[0], 0)
            for k in range(len(batches)):
                buf, names = pending.result()
                if k + 1 < len(batches):
                    pending = outer.submit(decode, batches[k + 1], (k + 1) & 1)
                with torch.cuda.stream(copy_stream):
                    dev = buf.to(self.device, non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(copy_stream)
                staged_free[k & 1] = done
                torch.cuda.current_stream(self.device).wait_event(done)
                dev.record_stream(torch.cuda.current_stream(self.device))
                yield dev, names


def test_batch(hr_u8, scale):
    """The test dataset's triplet (data/data.py:329-343): LR by torchvision Resize's default BILINEAR, the bicubic
    image from that LR.  hr_u8: [N, H, W, 3] uint8 on the device.  Returns (lr, hr, bc) float32 NCHW."""
    n, h, w, c = hr_u8.shape
    lr_u8 = resize_u8(hr_u8, h // scale, w // scale, 'bilinear')
    bc_u8 = resize_u8(lr_u8, h, w, 'bicubic')
    return to_tensor(lr_u8), to_tensor(hr_u8), to_tensor(bc_u8)
