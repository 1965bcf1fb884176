"""Host side of the input pipeline (sradsgan_amd/data.py: TileFolder, DevicePrefetcher, test_batch) and the trainer fed
from a folder of PNG tiles end to end."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _write_tiles(root, n, side, seed):
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(root, exist_ok=True)
    tiles = {}
    for i in range(n):
        # smooth-ish content so PSNRs are finite and resampling is exercised on structure, not only noise
        base = rng.randint(0, 256, (side // 4, side // 4, 3)).astype(np.uint8)
        arr = np.asarray(Image.fromarray(base).resize((side, side), Image.BICUBIC))
        name = 'tile_%02d.png' % i
        Image.fromarray(arr).save(os.path.join(root, name))
        tiles[name] = arr
    open(os.path.join(root, 'notes.txt'), 'w').write('not an image')
    return tiles


def test_tile_folder_and_prefetcher(tmp_path):
    from sradsgan_amd import data as D
    a = _write_tiles(str(tmp_path / 'a'), 5, 24, 0)
    b = _write_tiles(str(tmp_path / 'b'), 4, 24, 1)
    ds = D.TileFolder([str(tmp_path / 'b'), str(tmp_path / 'a')], crop_size=24, scale_factor=3)
    assert len(ds) == 9 and [os.path.basename(f) for f in ds.image_filenames[:4]] == sorted(b)      # dirs in given order
    t, fn = ds[5]
    assert np.array_equal(t.numpy(), a[os.path.basename(fn)])
    # sequential, drop_last: 9 tiles, batch 4 -> 2 batches; contents and order as listed
    seq = list(D.DevicePrefetcher(ds, 4, DEV, shuffle=False, drop_last=True, num_workers=3))
    assert len(seq) == 2 and all(x.is_cuda and x.dtype == torch.uint8 and tuple(x.shape) == (4, 24, 24, 3) for x, _ in seq)
    flat = [f for _, names in seq for f in names]
    assert flat == ds.image_filenames[:8]
    for x, names in seq:
        for j, f in enumerate(names):
            src = b if os.sep + 'b' + os.sep in f else a
            assert np.array_equal(x[j].cpu().numpy(), src[os.path.basename(f)])
    # keep the tail batch when drop_last=False; shuffling is a seeded permutation, a different one each epoch
    assert [x.shape[0] for x, _ in D.DevicePrefetcher(ds, 4, DEV, drop_last=False)] == [4, 4, 1]
    p1 = D.DevicePrefetcher(ds, 3, DEV, shuffle=True, seed=11)
    e1 = [f for _, names in p1 for f in names]
    e2 = [f for _, names in p1 for f in names]
    again = [f for _, names in D.DevicePrefetcher(ds, 3, DEV, shuffle=True, seed=11) for f in names]
    assert sorted(e1) == sorted(ds.image_filenames) and e1 == again and e1 != e2 and e1 != ds.image_filenames
    with pytest.raises(RuntimeError):
        D.DevicePrefetcher(ds, 4, 'cpu')


def test_test_batch_matches_pillow_transforms(tmp_path):
    """data/data.py:329-343: LR = Resize(crop // s) (bilinear default), bicubic image = that LR resized back BICUBIC."""
    from PIL import Image
    from sradsgan_amd import data as D
    tiles = _write_tiles(str(tmp_path / 't'), 2, 48, 3)
    arrs = [tiles[k] for k in sorted(tiles)]
    lr, hr, bc = D.test_batch(torch.from_numpy(np.stack(arrs)).to(DEV), 4)
    for j, arr in enumerate(arrs):
        im = Image.fromarray(arr)
        want_lr = np.asarray(im.resize((12, 12), Image.BILINEAR))
        want_bc = np.asarray(Image.fromarray(want_lr).resize((48, 48), Image.BICUBIC))
        for got, want in ((lr, want_lr), (hr, arr), (bc, want_bc)):
            assert torch.equal(got[j].cpu(), torch.from_numpy(want.copy()).permute(2, 0, 1).float().div(255))


def test_trainer_from_a_folder_of_tiles(tmp_path):
    from sradsgan_amd import data as D
    from sradsgan_amd import trainer as T
    _write_tiles(str(tmp_path / 'train'), 5, 32, 5)
    _write_tiles(str(tmp_path / 'test'), 2, 32, 6)
    train = D.DevicePrefetcher(D.TileFolder([str(tmp_path / 'train')], crop_size=32, scale_factor=4), 2, DEV, shuffle=True, seed=1)
    test = D.DevicePrefetcher(D.TileFolder([str(tmp_path / 'test')], crop_size=32, scale_factor=4), 2, DEV)
    args = T.default_args(scale_factor=4, num_epochs=1, batch_size=2, save_dir=str(tmp_path / 'out'), crop_size=32, hr_height=32,
                          hr_width=32, sample_interval=1, n_residual_blocks=1, n_basic_blocks=1)
    net = T.SRADSGAN(args, train_loader=train, test_loader=test)
    hist = net.train()
    assert len(hist) == 1 and np.isfinite([hist[0]['loss_G'], hist[0]['loss_D'], hist[0]['psnr'], hist[0]['ssim']]).all()
    assert hist[0]['psnr'] > 5.0
