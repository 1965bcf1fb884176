"""Checkpoint / chain-training host logic (sradsgan_amd/checkpoint.py) on CPU: file format the reference reads, the
shape-aware partial load between scales, and the plateau rule of sradsgan.py:985-1036 replayed step by step."""
import os

import torch

from oracle import sradsgan_ref as O
from sradsgan_amd import checkpoint as C
from sradsgan_amd import model as M


def _gen(scale, groups=2, blocks=1, mod=M):
    return mod.GeneratorResNet(mod.ResGroup, n_residual_blocks=groups, n_basic_blocks=blocks, upscale_factor=scale)


def test_epoch_files_round_trip_and_load_in_the_reference_layout(tmp_path):
    g = _gen(4)
    O.det_init_(g, prefix='G.')
    path = C.save_epoch_network(str(tmp_path), g, 'generator', 7)
    assert os.path.basename(path) == 'generator_param_epoch_7.pkl'
    raw = torch.load(path)                               # what the reference's load_epoch_network would read
    assert isinstance(raw, dict) and all(isinstance(v, torch.Tensor) and v.device.type == 'cpu' for v in raw.values())
    ref = _gen(4, mod=O)                                 # the oracle's generator carries the reference's key set
    ref.load_state_dict(raw, strict=True)
    g2 = _gen(4)
    C.load_epoch_network(path, g2, strict=True)
    for (k, a), (_, b) in zip(g.state_dict().items(), g2.state_dict().items()):
        assert torch.equal(a, b), k
    assert g2.GAB_UP.upsampling[0].weight is g2.GAB_UP.upsampling[3].weight       # tied stages stay tied after loading


def test_save_model_and_load_model(tmp_path):
    g, d = _gen(2), M.Discriminator()
    C.save_model(str(tmp_path), g, d)
    C.save_model(str(tmp_path), g, d, epoch=3)
    names = sorted(os.listdir(os.path.join(str(tmp_path), 'model')))
    assert names == ['discriminator_param.pkl', 'discriminator_param_epoch_3.pkl', 'generator_param.pkl', 'generator_param_epoch_3.pkl']
    g2 = _gen(2)
    assert C.load_model(str(tmp_path), g2) is True
    assert C.load_model(str(tmp_path / 'nothing'), g2) is False


def test_chain_training_partial_load_between_scales():
    x2 = _gen(2)
    O.det_init_(x2, prefix='G.')
    x4, x3 = _gen(4), _gen(3)
    loaded, missing, mismatch, unused = C.load_compatible(x4, x2.state_dict())
    assert not mismatch and not unused                   # x2 -> x4: same upsampler conv, one more (tied) stage
    assert sorted(missing) == ['GAB_UP.upsampling.3.bias', 'GAB_UP.upsampling.3.weight']
    assert torch.equal(x4.conv1[0].weight, x2.conv1[0].weight)
    assert torch.equal(x4.GAB_UP.upsampling[3].weight, x2.GAB_UP.upsampling[0].weight)   # tied: filled through stage 0
    loaded, missing, mismatch, unused = C.load_compatible(x3, x2.state_dict())
    assert sorted(mismatch) == ['GAB_UP.upsampling.0.bias', 'GAB_UP.upsampling.0.weight']   # 64 -> 256 vs 64 -> 576
    assert not missing and not unused and len(loaded) == len(x3.state_dict()) - 2
    assert torch.equal(x3.res_groups[1].conv.weight, x2.res_groups[1].conv.weight)


class _Lr:
    lr_G = 2e-4
    lr_D = 2e-4


def test_plateau_rule_replays_the_reference_sequence():
    ctl, step, reloaded = C.PlateauRollback(lr=2e-4), _Lr(), []
    epoch = 0
    # epochs 0..2 improve (psnr; then ssim only; then ergas only); epoch 3 still counts as an improvement because the
    # reference initialises lpips_max to 10000 and only the branch taken updates its own maximum (:985-1003); then five
    # epochs without any improvement
    seq = [(30.0, 0.80, 5.0, 0.5), (29.0, 0.81, 5.0, 0.5), (29.0, 0.80, 4.0, 0.5)] + [(28.0, 0.70, 6.0, 0.6)] * 6
    for i, m in enumerate(seq):
        epoch, rb = ctl.update(epoch, *m, step=step, on_rollback=reloaded.append)
        if i < 8:
            assert not rb and epoch == i + 1
    assert ctl.best_step == 3
    assert rb and reloaded == [4] and epoch == 4          # best epoch index 3 -> generator_param_epoch_4.pkl, resume at 4
    assert step.lr_G == 1e-4 and step.lr_D == 2e-4 and ctl.lr == 1e-4   # D halves only once lr < 1e-4 (checked before halving)
    assert len(ctl.history) == 4 and ctl.no_improve == 0
    for m in [(28.0, 0.70, 6.0, 0.6)] * 5:
        epoch, rb = ctl.update(epoch, *m, step=step)
    assert rb and step.lr_G == 5e-5 and step.lr_D == 2e-4 and ctl.lr == 5e-5
    for m in [(28.0, 0.70, 6.0, 0.6)] * 5:
        epoch, rb = ctl.update(epoch, *m, step=step)
    assert rb and step.lr_D == 1e-4                       # now lr (5e-5) < 1e-4
    assert ctl.keep_training(3, 100) and not C.PlateauRollback(lr=5e-6).keep_training(0, 100)


def test_val_log_line_format(tmp_path):
    """logger.py:117-147: field order, thousands separator, precision per mode."""
    from sradsgan_amd import validate
    rlt = dict(epoch=3, iters=1200, time=1.2549, model='AID', lr=2e-4, bicubic_psnr=27.123456, srwgan_psnr=29.5)
    assert validate.format_results('val', rlt) == \
        '<epoch:  3, iter:   1,200, time:1.25, lr:2.0e-04> dataset: AID bicubic_psnr: 2.7123e+01 srwgan_psnr: 2.9500e+01 '
    del rlt['lr']
    assert validate.format_results('train', rlt) == \
        '<epoch:  3, iter:   1,200, time:1.25> dataset: AID bicubic_psnr: 2.71e+01 srwgan_psnr: 2.95e+01 '
    p = tmp_path / 'val_log.txt'
    validate.append_log(str(p), 'val', rlt)
    validate.append_log(str(p), 'val', rlt)
    assert p.read_text().count('\n') == 2 and 'epoch' in rlt


def test_dataset_directory_walk_matches_the_reference(tmp_path):
    """data.rgb_train_dirs / rgb_test_dirs restate get_RGB_trainDataset / get_RGB_testDataset's directory lists
    (data/data.py:295-325): SECOND flat, the four class-folder datasets expanded, unknown names ignored, UCMerced's
    class folders sorted, other test entries taken as directories themselves."""
    import os
    from sradsgan_amd import data as D
    for d in ('AID/b', 'AID/a', 'SECOND', 'DOTA/x', 'Other/z', 'UCMerced_LandUse/river', 'UCMerced_LandUse/beach'):
        os.makedirs(os.path.join(str(tmp_path), d))
    open(os.path.join(str(tmp_path), 'AID', 'readme.txt'), 'w').close()
    got = D.rgb_train_dirs(str(tmp_path), ['AID', 'SECOND', 'Other', 'DOTA'])
    assert sorted(got[:2]) == [os.path.join(str(tmp_path), 'AID', 'a'), os.path.join(str(tmp_path), 'AID', 'b')]
    assert got[2:] == [os.path.join(str(tmp_path), 'SECOND'), os.path.join(str(tmp_path), 'DOTA', 'x')]
    assert D.rgb_test_dirs(str(tmp_path), ['UCMerced_LandUse']) == [os.path.join(str(tmp_path), 'UCMerced_LandUse', c) for c in ('beach', 'river')]
    assert D.rgb_test_dirs(str(tmp_path), ['/some/dir']) == ['/some/dir']
