"""The reference's trainer surface on the HIP path (sradsgan_amd/trainer.py): two tiny epochs end to end."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _loaders():
    g = torch.Generator().manual_seed(21)
    train = [torch.randint(0, 256, (2, 32, 32, 3), generator=g, dtype=torch.uint8) for _ in range(2)]       # uint8 HR tiles
    hr = torch.rand(2, 3, 32, 32, generator=g)
    test = [(torch.nn.functional.avg_pool2d(hr, 4), hr, hr.clamp(0, 1), ['a', 'b'])]                          # reference-style tuple
    return train, test


def test_train_validate_and_checkpoints_end_to_end(tmp_path):
    from sradsgan_amd import trainer as T
    train, test = _loaders()
    args = T.default_args(scale_factor=4, num_epochs=2, batch_size=2, save_dir=str(tmp_path), crop_size=32, hr_height=32,
                          hr_width=32, sample_interval=1, n_residual_blocks=1, n_basic_blocks=1)
    assert args.lr == 0.0002 and args.lambda_gp == 10 and args.weight_gan == 1e-3          # reference defaults survive
    net = T.SRADSGAN(args, train_loader=train, test_loader=test)
    hist = net.train()
    assert len(hist) == 2 and all(torch.isfinite(torch.tensor([h['loss_G'], h['loss_D'], h['psnr'], h['ssim'], h['ergas']])).all() for h in hist)
    files = sorted(os.listdir(os.path.join(str(tmp_path), 'model')))
    assert files == ['discriminator_param.pkl', 'discriminator_param_epoch_1.pkl', 'discriminator_param_epoch_2.pkl',
                     'generator_param.pkl', 'generator_param_epoch_1.pkl', 'generator_param_epoch_2.pkl']
    val_lines = open(os.path.join(str(tmp_path), 'val_log.txt')).read().strip().splitlines()
    assert len(val_lines) == 2 and val_lines[0].startswith('<epoch:  0, iter:       0, time:') and 'srcnn_psnr:' in val_lines[0]
    assert 'loss_G:' in open(os.path.join(str(tmp_path), 'loss_log.txt')).read()
    # mfeNew_validate on the saved final generator reproduces the last in-training validation
    psnr, ssim, ergas, lpips = net.mfeNew_validate(epoch=2, modelpath=os.path.join(str(tmp_path), 'model', 'generator_param_epoch_2.pkl'))
    assert abs(psnr - hist[-1]['psnr']) < 1e-9 and abs(ssim - hist[-1]['ssim']) < 1e-12 and lpips != lpips
    assert 'sradsgan_psnr:' in open(os.path.join(str(tmp_path), 'val_log.txt')).read().splitlines()[-1]
    # resume path: epoch != 0 loads the epoch files strictly (sradsgan.py:705-711)
    args2 = T.default_args(**dict(vars(args), epoch=2, num_epochs=2))
    net2 = T.SRADSGAN(args2, train_loader=train, test_loader=test)
    net2._build()
    for (k, a), (_, b) in zip(net.generator.state_dict().items(), net2.generator.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k


def test_validate_by_class_and_single_image(tmp_path):
    """mfeNew_validateByClass (sradsgan.py:1393-1601): per-class and Total lines, Total = image-weighted mean of the
    classes; mfe_test_single (:1603-1641): the written PNGs hold save_img1's quantisation of the generator output and
    Pillow's own bicubic of the crop."""
    import numpy as np
    from collections import OrderedDict
    from PIL import Image
    from sradsgan_amd import trainer as T
    g = torch.Generator().manual_seed(5)

    def loader(n_batches):
        out = []
        for _ in range(n_batches):
            hr = torch.rand(2, 3, 32, 32, generator=g)
            out.append((torch.nn.functional.avg_pool2d(hr, 4), hr, hr.clamp(0, 1), ['x', 'y']))
        return out
    args = T.default_args(scale_factor=4, save_dir=str(tmp_path), crop_size=32, hr_height=32, hr_width=32, test_crop_size=12,
                          n_residual_blocks=1, n_basic_blocks=1)
    net = T.SRADSGAN(args)
    torch.manual_seed(3)
    gen = net._new_generator()
    gen.apply(T.weights_init_normal)
    path = os.path.join(str(tmp_path), 'g.pkl')
    torch.save(gen.state_dict(), path)
    net.class_loaders = OrderedDict([('airplane', loader(1)), ('beach', loader(2))])
    res = net.mfeNew_validateByClass(7, modelpath=path)
    assert list(res.keys()) == ['airplane', 'beach', 'Total']
    for k in ('sradsgan_psnr', 'bicubic_ssim', 'sradsgan_ergas', 'bicubic_mse'):
        assert abs(res['Total'][k] - (2 * res['airplane'][k] + 4 * res['beach'][k]) / 6) < 1e-9, k
    lines = open(os.path.join(str(tmp_path), 'val_log.txt')).read().strip().splitlines()
    assert len(lines) == 3 and all(ln.startswith('<epoch:  7, iter:       7, time:') for ln in lines)
    # single image: 20x16 PNG, centre crop 12 -> x4 = 48
    rgb = (torch.rand(16, 20, 3, generator=g) * 255).to(torch.uint8).numpy()
    fn = os.path.join(str(tmp_path), 'tile.png')
    Image.fromarray(rgb).save(fn)
    sr, bc = net.mfe_test_single(fn, modelpath=path)
    assert sr.shape == (48, 48, 3) and bc.shape == (48, 48, 3)
    crop = Image.fromarray(rgb).crop((4, 2, 16, 14))
    assert np.array_equal(bc, np.asarray(crop.resize((48, 48), Image.BICUBIC)))                 # bit-exact Pillow bicubic
    x = torch.from_numpy(np.asarray(crop, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255).unsqueeze(0).to(DEV)
    net.generator.eval()
    with torch.no_grad():
        want = (net.generator(x)[0] * 255.0).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(sr, want)
    for tag, arr in (('SRADSGAN', sr), ('Bicubic', bc)):
        assert np.array_equal(np.asarray(Image.open(os.path.join(str(tmp_path), 'SR_%s_tile.png' % tag))), arr)


def test_chain_training_two_scales(tmp_path):
    """x2 -> x3 with one tiny epoch each: stage 2 starts from stage 1's files; across 2^n -> 3^n only the up-sampler conv
    (64 -> 256 vs 64 -> 576) keeps its fresh init, the discriminator loads completely."""
    from sradsgan_amd import trainer as T
    g = torch.Generator().manual_seed(9)

    def loaders(scale):
        side = 12 * scale
        train = [torch.randint(0, 256, (2, side, side, 3), generator=g, dtype=torch.uint8) for _ in range(2)]
        test = [torch.randint(0, 256, (2, side, side, 3), generator=g, dtype=torch.uint8)]
        return train, test
    args = T.default_args(num_epochs=1, batch_size=2, save_dir=str(tmp_path), sample_interval=1, n_residual_blocks=1,
                          n_basic_blocks=1)
    res = T.chain_train(args, [2, 3], loaders)
    assert list(res.keys()) == [2, 3] and res[2][1] == {}
    loaded, missing, mismatch, unused = res[3][1]['generator']
    assert sorted(mismatch) == ['GAB_UP.upsampling.0.bias', 'GAB_UP.upsampling.0.weight'] and not missing and not unused
    assert len(loaded) == 38
    d_loaded, d_missing, d_mismatch, d_unused = res[3][1]['discriminator']
    assert not d_missing and not d_mismatch and not d_unused
    for scale in (2, 3):
        assert os.path.exists(os.path.join(str(tmp_path), 'x%d' % scale, 'model', 'generator_param.pkl'))
        assert all(torch.isfinite(torch.tensor([h['loss_G'], h['psnr']])).all() for h in res[scale][0])


def test_unsupported_reference_options_fail_loudly():
    from sradsgan_amd import trainer as T
    with pytest.raises(NotImplementedError):
        T.SRADSGAN(T.default_args(penalty_type='hinge'))


def _write_tiles(folder, n, size, seed):
    import numpy as np
    from PIL import Image
    os.makedirs(folder, exist_ok=True)
    rng = np.random.RandomState(seed)
    for i in range(n):
        Image.fromarray(rng.randint(0, 256, (size, size, 3), dtype=np.uint8)).save(os.path.join(folder, 'tile_%02d.png' % i))


def test_drop_in_train_from_data_dir_without_injected_loaders(tmp_path):
    """`net = SRADSGAN(args); net.train(); net.mfeNew_validate(...)` exactly as main_sradsgan.py:116-135 calls them: the
    loaders come from args.data_dir / train_dataset / test_dataset (sradsgan.py:643-656, data/data.py:295-346)."""
    from sradsgan_amd import trainer as T
    data = tmp_path / 'dataset'
    _write_tiles(str(data / 'AID' / 'Airport'), 3, 32, 1)
    _write_tiles(str(data / 'AID' / 'Beach'), 2, 32, 2)
    _write_tiles(str(data / 'SECOND'), 2, 32, 3)
    _write_tiles(str(data / 'UCMerced_LandUse' / 'agricultural'), 2, 32, 4)
    _write_tiles(str(data / 'UCMerced_LandUse' / 'airplane'), 1, 32, 5)
    args = T.default_args(scale_factor=4, num_epochs=1, batch_size=2, test_batch_size=1, save_dir=str(tmp_path / 'Result'),
                          data_dir=str(data), train_dataset=['AID', 'SECOND'], test_dataset=['UCMerced_LandUse'], crop_size=32,
                          hr_height=32, hr_width=32, sample_interval=1, num_threads=2, n_residual_blocks=1, n_basic_blocks=1)
    net = T.SRADSGAN(args)                                   # no loaders injected
    hist = net.train()
    assert len(net.train_loader) == 3 and len(net.test_loader) == 3          # 7 training tiles // 2, 3 test tiles
    assert len(hist) == 1 and all(v == v for v in (hist[0]['loss_G'], hist[0]['loss_D'], hist[0]['psnr']))
    model = os.path.join(str(tmp_path / 'Result'), 'model', 'generator_param_epoch_1.pkl')
    net2 = T.SRADSGAN(args)
    psnr, ssim, ergas, lpips = net2.mfeNew_validate(epoch=1, modelpath=model)
    assert abs(psnr - hist[0]['psnr']) < 1e-9 and lpips != lpips
    by_class = net2.mfeNew_validateByClass(epoch=1, modelpath=model)
    assert list(by_class.keys()) == ['agricultural', 'airplane', 'Total']
    tot = (2 * by_class['agricultural']['sradsgan_psnr'] + by_class['airplane']['sradsgan_psnr']) / 3
    assert abs(tot - by_class['Total']['sradsgan_psnr']) < 1e-9 and abs(by_class['Total']['sradsgan_psnr'] - psnr) < 1e-9
