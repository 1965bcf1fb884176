"""The reference's trainer surface on the HIP path: `SRADSGAN(args)` with `.train()`, `.validate()`,
`.mfeNew_validate()` and the save/load helpers of SRADSGAN/model/sradsgan.py:511-1391, so `main_sradsgan.py`'s
    net = SRADSGAN(args); net.train(); net.mfeNew_validate(epoch=.., modelpath=..)
keeps working with the import changed.  Same argparse field names and defaults
(main_sradsgan.py:16-61, `default_args`), same loss weights / optimiser settings / step order (TrainStep), same
checkpoint names and log lines, same epoch control (PlateauRollback).

The data side: like the reference (`load_dataset`, sradsgan.py:643-656) the trainer builds its loaders from
`args.data_dir / train_dataset / test_dataset` when none are injected -- the same directory walk and file order
(data/data.py:295-346), but as `data.TileFolder` + `data.DevicePrefetcher`: uint8 HR tiles decoded by worker threads,
copied on their own HIP stream, and turned into (lr, hr, bc) on the device (sradsgan_amd.data.training_batch /
test_batch, bit-exact with the reference's PIL transforms) instead of 16 PIL worker processes.  An injected iterable
(`train_loader`, `test_loader`: the reference's own DataLoader tuples `(lr, hr, bc, paths)` or uint8 HR tiles
`[N, H, W, 3]`) takes precedence.  Metrics are computed on the device (validate.py); LPIPS is not reproduced (AlexNet weights are a
download): the lpips slots of the returned tuples and log lines carry NaN.  PNG panels are not written."""
import argparse
import math
import os
import time
from collections import OrderedDict

import numpy as np
import torch

from . import checkpoint as ckpt
from . import data as sdata
from . import dp
from . import validate as sval
from .model import Discriminator, FeatureExtractor, GeneratorResNet, ResGroup
from .train_step import TrainStep


def default_args(**overrides):
    """argparse.Namespace with the reference's flag names and defaults (main_sradsgan.py:16-61)."""
    d = dict(model_name='SRADSGAN', root_dir='.', data_dir='.', train_dataset=['AID', 'DOTA', 'LoveDA', 'RSSCN7_2800', 'SECOND'],
             test_dataset=['UCMerced_LandUse'], crop_size=216, num_threads=16, num_channels=3, scale_factor=8, epoch=0,
             num_epochs=100, save_epochs=1, batch_size=16, test_batch_size=1, save_dir='Result', lr=0.0002, b1=0.9, b2=0.999,
             gpu_mode=True, test_crop_size=216, n_cpu=16, hr_height=216, hr_width=216, sample_interval=1000, clip_value=0.01,
             lambda_gp=10, gp=True, penalty_type='LS', grad_penalty_Lp_norm='L2', relativeGan=False, loss_Lp_norm='L1',
             weight_content=1e-2, weight_gan=1e-3, max_train_samples=40000, is_train=True)
    d.update(overrides)
    return argparse.Namespace(**d)


def weights_init_normal(m, mean=0.0, std=0.02):
    """utils/utils.py:97-114 (applied at sradsgan.py:713-714)."""
    name = m.__class__.__name__
    if name.find('Linear') != -1 or name.find('Conv2d') != -1 or name.find('ConvTranspose2d') != -1:
        m.weight.data.normal_(mean, std)
        if m.bias is not None:
            m.bias.data.zero_()
    elif name.find('BatchNorm') != -1:
        m.weight.data.normal_(1.0, 0.02)
        if m.bias is not None:
            m.bias.data.zero_()


class SRADSGAN(object):
    def __init__(self, args, train_loader=None, test_loader=None):
        for k in ('model_name', 'train_dataset', 'test_dataset', 'crop_size', 'test_crop_size', 'hr_height', 'hr_width',
                  'num_threads', 'num_channels', 'scale_factor', 'epoch', 'num_epochs', 'save_epochs', 'batch_size',
                  'test_batch_size', 'lr', 'b1', 'b2', 'data_dir', 'root_dir', 'save_dir', 'gpu_mode', 'n_cpu',
                  'sample_interval', 'clip_value', 'lambda_gp', 'gp', 'penalty_type', 'grad_penalty_Lp_norm',
                  'weight_gan', 'weight_content', 'max_train_samples'):                       # sradsgan.py:513-556
            setattr(self, k, getattr(args, k))
        self.relative = args.relativeGan
        self.loss_Lp_norm = args.loss_Lp_norm
        if self.penalty_type != 'LS' or self.grad_penalty_Lp_norm != 'L2' or self.loss_Lp_norm != 'L1' or self.relative:
            raise NotImplementedError('the HIP step implements the reference defaults: LS penalty, L2 gradient norm, L1 '
                                      'content loss, non-relativistic GAN (sradsgan.py:595-641, 829-892)')
        if not torch.cuda.is_available():
            raise Exception('No GPU found, please run without --gpu_mode=False')               # main_sradsgan.py:95-96
        # generator depth: the reference hard-codes 12 groups x 3 blocks (:669-671); overridable for tests
        self.n_residual_blocks = getattr(args, 'n_residual_blocks', 12)
        self.n_basic_blocks = getattr(args, 'n_basic_blocks', 3)
        self.pretrained_generator = getattr(args, 'pretrained_generator', None)      # chain training, see _build
        self.pretrained_discriminator = getattr(args, 'pretrained_discriminator', None)
        self.train_loader, self.test_loader = train_loader, test_loader
        # data parallel: one process per GPU under torch.distributed; replicas start identical (broadcast
        # from rank 0), gradients are averaged by TrainStep's GradSync, BatchNorm stays local, rank 0 owns files and
        # logs, validation results and with them every rollback decision are rank 0's
        self.rank, self.world = dp.rank_world()
        self.device = torch.device('cuda', torch.cuda.current_device())
        self.generator = self.discriminator = self.feature_extractor = None
        self.step = None
        self.log_dict = OrderedDict()
        self.loss_log_path = os.path.join(self.save_dir, 'loss_log.txt')
        self.val_log_path = os.path.join(self.save_dir, 'val_log.txt')

    # ------------------------------------------------------------------ networks ---------------- #
    def _new_generator(self):
        return GeneratorResNet(ResGroup, n_residual_blocks=self.n_residual_blocks, n_basic_blocks=self.n_basic_blocks,
                               rla_mode='CA-SA', bla_mode='CA-SA', ga_mode='CA-SA', pool_mode='Avg|Max', addconv=True,
                               upscale_factor=self.scale_factor)                               # :669-671, :1263-1265

    def _build(self):
        self.generator = self._new_generator()
        self.discriminator = Discriminator()
        self.feature_extractor = FeatureExtractor()
        model_dir = os.path.join(self.save_dir, 'model')
        if self.epoch != 0:                                                                     # :705-711
            self.load_epoch_network(model_dir + '/generator_param_epoch_%d.pkl' % self.epoch, self.generator, strict=True)
            self.load_epoch_network(model_dir + '/discriminator_param_epoch_%d.pkl' % self.epoch, self.discriminator, strict=True)
        else:
            self.generator.apply(weights_init_normal)                                           # :713-714
            self.discriminator.apply(weights_init_normal)
            # chain training (:716-721, commented out in the reference and edited by hand per scale): start from the
            # previous scale's checkpoints.  strict=False there still raises on shape mismatches, so the partial load
            # is shape-aware (checkpoint.load_compatible): the up-sampler conv keeps its fresh init across 2^n <-> 3^n
            self.chain_report = {}
            for label, net in (('generator', self.generator), ('discriminator', self.discriminator)):
                path = getattr(self, 'pretrained_' + label, None)
                if path:
                    self.chain_report[label] = ckpt.load_compatible(net, torch.load(path, map_location='cpu'))
                    print('%s initialised from %s: %d tensors loaded, %d kept their init' % (
                        label, path, len(self.chain_report[label][0]),
                        len(self.chain_report[label][1]) + len(self.chain_report[label][2])))
        self.generator.to(self.device), self.discriminator.to(self.device), self.feature_extractor.to(self.device)
        self.feature_extractor.eval()                                                           # :727
        if self.world > 1:
            dp.broadcast_module(self.generator, 0), dp.broadcast_module(self.discriminator, 0)
            dp.broadcast_module(self.feature_extractor, 0)
            if getattr(self, 'grad_sync', None) is None:
                self.grad_sync = dp.GradSync(self.world)
            self._alpha_rng = np.random.RandomState(1234 + self.rank)                           # alpha drawn per rank
        self.step = TrainStep(self.generator, self.discriminator, self.feature_extractor, lr=self.lr, b1=self.b1, b2=self.b2,
                              weight_content=self.weight_content, weight_gan=self.weight_gan, lambda_gp=self.lambda_gp,
                              clip_value=self.clip_value, use_gp=bool(self.gp), grad_sync=getattr(self, 'grad_sync', None))

    def _batch(self, item, test=False):
        """(lr, hr, bc) device float tensors from a loader item: the reference's 4-tuple, a uint8 HR batch [B,H,W,3],
        or data.DevicePrefetcher's (uint8 batch, file names).  uint8 tiles get the reference's per-sample transforms
        on the device: bicubic LR for training (dataset.py:418-436), bilinear LR for the test set (data.py:329-343)."""
        if isinstance(item, (tuple, list)) and len(item) == 2 and torch.is_tensor(item[0]) and item[0].dtype == torch.uint8:
            item = item[0]
        if torch.is_tensor(item) and item.dtype == torch.uint8:
            make = sdata.test_batch if test else sdata.training_batch
            return make(item.to(self.device), self.scale_factor)
        lr, hr, bc = item[0], item[1], item[2]
        return lr.to(self.device).float(), hr.to(self.device).float(), bc.to(self.device).float()

    # ------------------------------------------------------------------ datasets ---------------- #
    def load_dataset(self, dataset='train', max_samples=20000, dirs=None):
        """sradsgan.py:643-656: DataLoader(shuffle=True, drop_last=True) over the training folders resp.
        DataLoader(shuffle=False, drop_last=True) over the test folders of `args.data_dir`, as a device prefetcher of
        uint8 tiles.  `dirs` overrides the directory list (per-class validation)."""
        if self.num_channels == 1:
            raise NotImplementedError('load_dataset: RGB tiles only (num_channels=3 is what main_sradsgan.py passes)')
        if dataset == 'train':
            print('Loading train dct_datasets...')
            folders = sdata.rgb_train_dirs(self.data_dir, self.train_dataset) if dirs is None else dirs
            tiles = sdata.TileFolder(folders, crop_size=self.crop_size, scale_factor=self.scale_factor)
            return sdata.DevicePrefetcher(tiles, self.batch_size, self.device, shuffle=True, drop_last=True,
                                          num_workers=self.num_threads, rank=self.rank, world=self.world)
        if dataset == 'test':
            print('Loading test dct_datasets...')
            folders = sdata.rgb_test_dirs(self.data_dir, self.test_dataset) if dirs is None else dirs
            tiles = sdata.TileFolder(folders, crop_size=self.crop_size, scale_factor=self.scale_factor)
            return sdata.DevicePrefetcher(tiles, self.test_batch_size, self.device, shuffle=False, drop_last=True,
                                          num_workers=self.num_threads)
        raise ValueError("load_dataset: dataset must be 'train' or 'test'")

    # ------------------------------------------------------------------ training ---------------- #
    def train(self):
        """sradsgan.py:658-1036.  Returns the per-epoch history (avg losses and validation metrics)."""
        if self.train_loader is None:
            self.train_loader = self.load_dataset(dataset='train', max_samples=self.max_train_samples)    # :749
        if self.test_loader is None:
            self.test_loader = self.load_dataset(dataset='test')                                          # :750
        os.makedirs(self.save_dir, exist_ok=True)
        self._build()
        model_dir = os.path.join(self.save_dir, 'model')
        control = ckpt.PlateauRollback(self.lr)
        avg_loss_G, avg_loss_D, history = [], [], []
        step_count = 0
        epoch = self.epoch
        start_time = time.time()
        while control.keep_training(epoch, self.num_epochs):                                    # :803
            self.generator.train()
            self.discriminator.train()
            epoch_loss_G = epoch_loss_D = 0.0
            n_batches = 0
            for i, item in enumerate(self.train_loader):
                imgs_lr, imgs_hr, _ = self._batch(item)
                rng = getattr(self, '_alpha_rng', None) or np.random
                alpha = torch.from_numpy(rng.random_sample((imgs_hr.size(0), 1, 1, 1))).float().to(self.device)   # :609
                out = self.step(imgs_lr, imgs_hr, alpha)                                        # :829-892
                step_count += 1
                n_batches += 1
                if (i + 1) % self.sample_interval == 0 or i == 0:                               # host reads only at log cadence
                    lg, ld = float(out['loss_G']), float(out['loss_D'])
                    self.log_dict['loss_G'], self.log_dict['loss_D'] = lg, ld
                    rlt = OrderedDict(model=self.model_name, epoch=epoch, iters=step_count, time=time.time() - start_time)
                    rlt.update(self.log_dict)
                    if self.rank == 0:
                        print(sval.append_log(self.loss_log_path, 'train', rlt))               # :898-906, 963-969
                epoch_loss_G = epoch_loss_G + out['loss_G']
                epoch_loss_D = epoch_loss_D + out['loss_D']
            avg_loss_G.append(float(epoch_loss_G) / max(n_batches, 1))                          # :975-976
            avg_loss_D.append(float(epoch_loss_D) / max(n_batches, 1))
            val = self.validate(epoch=epoch, mode='train', save_img=((epoch + 1) % self.save_epochs == 0))   # :978
            val = tuple(dp.broadcast_floats(val, 0, self.device))                               # rank 0's numbers decide
            history.append(dict(epoch=epoch, loss_G=avg_loss_G[-1], loss_D=avg_loss_D[-1], psnr=val[0], ssim=val[1],
                                ergas=val[2], lpips=val[3]))
            if self.rank == 0:
                self.save_epoch_network(save_dir=model_dir, network=self.generator, network_label='generator', iter_label=epoch + 1)
                self.save_epoch_network(save_dir=model_dir, network=self.discriminator, network_label='discriminator', iter_label=epoch + 1)
            dp.barrier()                                                                        # files exist before any rank rolls back
            reload_g = lambda n: self.load_epoch_network(model_dir + '/generator_param_epoch_%d.pkl' % n, self.generator)
            epoch, rolled = control.update(epoch, val[0], val[1], val[2], val[3] if not math.isnan(val[3]) else 10000,
                                           step=self.step, on_rollback=reload_g)               # :985-1036
            self.lr = control.lr
            if rolled:
                print('optimizer_G_Learning rate decay: lr={}'.format(self.step.lr_G))
        if self.rank == 0:
            self.save_model(epoch=None)
        dp.barrier()
        return history

    # ------------------------------------------------------------------ validation -------------- #
    def _evaluate_loader(self, generator, label, loader=None, totals=None):
        if loader is None:
            if self.test_loader is None:
                self.test_loader = self.load_dataset(dataset='test')                              # :1074, :1277
            loader = self.test_loader
        sums = {k: 0.0 for k in ('bicubic_mse', 'bicubic_psnr', 'bicubic_ssim', 'bicubic_ergas', label + '_mse', label + '_psnr',
                                 label + '_ssim', label + '_ergas')}
        img_num = 0
        was_training = generator.training
        generator.eval()                                                                         # :1288
        start = time.time()
        for item in loader:
            imgs_lr, imgs_hr, imgs_bc = self._batch(item, test=True)
            out = sval.evaluate(generator, imgs_lr, imgs_hr, self.scale_factor, bicubic=imgs_bc)
            img_num += imgs_hr.size(0)
            for k in ('mse', 'psnr', 'ssim', 'ergas'):
                sums['bicubic_' + k] += float(out['bicubic'][k].sum())
                sums[label + '_' + k] += float(out['sr'][k].sum())
        generator.train(was_training)
        if totals is not None:                                                                   # running sums over classes
            totals['num'] = totals.get('num', 0) + img_num
            for k, v in sums.items():
                totals[k] = totals.get(k, 0.0) + v
        avg = {k: v / max(img_num, 1) for k, v in sums.items()}                                 # :1365-1374
        return avg, time.time() - start

    def _log_val(self, epoch, avg, elapsed, label, model=None):
        rlt = OrderedDict(model=self.model_name if model is None else model, epoch=epoch, iters=epoch, time=elapsed)
        for prefix in ('bicubic', label):                                                        # :1377-1390 key order
            for k in ('mse', 'psnr', 'ssim', 'ergas'):
                rlt['%s_%s' % (prefix, k)] = avg['%s_%s' % (prefix, k)]
            rlt['%s_lpips' % prefix] = float('nan')
        if self.rank != 0:
            return
        os.makedirs(self.save_dir, exist_ok=True)
        print(sval.append_log(self.val_log_path, 'val', rlt))

    def validate(self, epoch=0, mode='test', save_img=False):
        """sradsgan.py:1058-1194 -> (psnr, ssim, ergas, lpips) averaged over the test set."""
        if mode == 'test':
            self.generator = self._new_generator().to(self.device)                              # :1065-1067
            self.load_epoch_model(epoch)
        avg, elapsed = self._evaluate_loader(self.generator, 'srcnn')
        self._log_val(epoch, avg, elapsed, 'srcnn')
        return avg['srcnn_psnr'], avg['srcnn_ssim'], avg['srcnn_ergas'], float('nan')

    def mfeNew_validate(self, epoch=100, modelpath=None):
        """sradsgan.py:1258-1391: fresh generator, optional `modelpath` (strict=False), same averages and log line
        (keys bicubic_* / sradsgan_*)."""
        self.generator = self._new_generator().to(self.device)
        if modelpath is not None:
            self.generator.load_state_dict(torch.load(modelpath, map_location='cpu'), strict=False)   # :1270-1271
            ckpt._after_load()
        avg, elapsed = self._evaluate_loader(self.generator, 'sradsgan')
        self._log_val(epoch, avg, elapsed, 'sradsgan')
        return avg['sradsgan_psnr'], avg['sradsgan_ssim'], avg['sradsgan_ergas'], float('nan')

    def mfeNew_validateByClass(self, epoch, save_img=False, modelpath=None):
        """sradsgan.py:1393-1601: the validation of mfeNew_validate once per class folder of the test set, one
        `val_log.txt` line per class (model = class name) and a final "Total" line over all images.  Like the
        reference one loader per class directory of the test set is built (:1433-1447) unless `self.class_loaders` (an
        ordered mapping class name -> iterable of batches) was injected.  Returns {class or 'Total': averages}."""
        loaders = getattr(self, 'class_loaders', None)
        if not loaders:                                                                          # :1430-1447
            loaders = OrderedDict()
            for d in sdata.rgb_test_dirs(self.data_dir, self.test_dataset):
                loaders[os.path.split(d)[-1]] = self.load_dataset(dataset='test', dirs=[d])
                print('Number of val images in [{:s}]: {:d}'.format(d, len(loaders[os.path.split(d)[-1]])))
        self.generator = self._new_generator().to(self.device)
        if modelpath is not None:
            self.generator.load_state_dict(torch.load(modelpath, map_location='cpu'), strict=False)   # :1400-1401
            ckpt._after_load()
        start, totals, result = time.time(), {}, OrderedDict()
        for name, loader in loaders.items():
            avg, _ = self._evaluate_loader(self.generator, 'sradsgan', loader=loader, totals=totals)
            self._log_val(epoch, avg, time.time() - start, 'sradsgan', model=name)              # :1548-1564
            result[name] = avg
        num = max(totals.pop('num', 0), 1)
        result['Total'] = {k: v / num for k, v in totals.items()}                               # :1568-1577
        self._log_val(epoch, result['Total'], time.time() - start, 'sradsgan', model='Total')
        return result

    def mfe_test_single(self, img_fn, modelpath=None):
        """sradsgan.py:1603-1641: centre-crop `test_crop_size` of one image file, super-resolve it, and write
        `SR_SRADSGAN_<name>` and `SR_Bicubic_<name>` into save_dir with save_img1's quantisation (x255, clamp, truncate;
        utils/utils.py:169-187).  The bicubic image is img_interp's (utils.py:755-780: uint8 via ToPILImage, Pillow
        bicubic) on the device.  The comparison figure (matplotlib) is not drawn.  Returns the two uint8 HWC arrays."""
        from PIL import Image
        import numpy as np
        self.generator = self._new_generator().to(self.device)
        if modelpath is not None:
            self.generator.load_state_dict(torch.load(modelpath, map_location='cpu'), strict=False)   # :1609-1610
            ckpt._after_load()
        self.generator.eval()
        img = Image.open(img_fn).convert('RGB')
        c = self.test_crop_size
        left, top = int(round((img.width - c) / 2.0)), int(round((img.height - c) / 2.0))          # transforms.CenterCrop
        u8 = torch.from_numpy(np.asarray(img.crop((left, top, left + c, top + c)), dtype=np.uint8).copy())
        u8 = u8.unsqueeze(0).contiguous().to(self.device)                                        # [1,c,c,3] uint8
        x = sdata.to_tensor(u8)
        with torch.no_grad():
            sr = self.generator(x)
        bc_u8 = sdata.resize_u8(u8, c * self.scale_factor, c * self.scale_factor, 'bicubic')[0]  # [H,W,3]
        sr_u8 = (sr[0] * 255.0).clamp(0, 255).to(torch.uint8).permute(1, 2, 0)                   # save_img1
        os.makedirs(self.save_dir, exist_ok=True)
        name = os.path.basename(img_fn)
        out = []
        for tag, t in (('SRADSGAN', sr_u8), ('Bicubic', bc_u8)):
            arr = t.contiguous().cpu().numpy()
            Image.fromarray(arr).save(os.path.join(self.save_dir, 'SR_%s_%s' % (tag, name)))
            out.append(arr)
        return tuple(out)

    # ------------------------------------------------------------------ checkpoints ------------- #
    def save_epoch_network(self, save_dir, network, network_label, iter_label):
        return ckpt.save_epoch_network(save_dir, network, network_label, iter_label)

    def load_epoch_network(self, load_path, network, strict=True):
        ckpt.load_epoch_network(load_path, network, strict=strict)
        print('Trained model is loaded.')

    def save_model(self, epoch=None):
        ckpt.save_model(self.save_dir, self.generator, self.discriminator, epoch)
        print('Trained model is saved.')

    def load_model(self):
        ok = ckpt.load_model(self.save_dir, self.generator)
        print('Trained model is loaded.' if ok else 'No model exists to load.')
        return ok

    def load_epoch_model(self, epoch):
        path = os.path.join(self.save_dir, 'model', 'generator_param_epoch_%d.pkl' % epoch)
        if not os.path.exists(path):
            print('No model exists to load.')
            return False
        ckpt.load_epoch_network(path, self.generator)
        print('Trained model is loaded.')
        return True


def chain_train(args, scales, loaders, trainer_cls=SRADSGAN):
    """BASELINE configs[4]: the multi-scale chain x2 -> x3 -> x4 -> x8 -> x9 the reference runs by hand (re-launching
    main_sradsgan.py per --scale_factor after editing the paths at sradsgan.py:716-721).  One trainer per scale, results
    under <save_dir>/x<scale>; every stage after the first starts from the previous stage's final generator /
    discriminator files through the shape-aware partial load.  `loaders(scale)` -> (train_loader, test_loader) for
    that scale's tile size.  Returns {scale: (history, chain_report)}."""
    out, prev = OrderedDict(), None
    for scale in scales:
        a = argparse.Namespace(**vars(args))
        a.scale_factor, a.epoch = scale, 0
        a.save_dir = os.path.join(args.save_dir, 'x%d' % scale)
        if prev is not None:
            a.pretrained_generator = os.path.join(prev, 'model', 'generator_param.pkl')
            a.pretrained_discriminator = os.path.join(prev, 'model', 'discriminator_param.pkl')
        train_loader, test_loader = loaders(scale)
        net = trainer_cls(a, train_loader=train_loader, test_loader=test_loader)
        history = net.train()
        out[scale] = (history, getattr(net, 'chain_report', {}))
        prev = a.save_dir
    return out
