"""Checkpoint I/O and the plateau / chain-training control of the reference's trainer, host
logic only.  Mirrors SRADSGAN/model/sradsgan.py:
  save_epoch_network  :1041-1051   '<label>_param_epoch_<n>.pkl' = torch.save(state_dict on CPU)
  load_epoch_network  :1054-1058   load_state_dict(torch.load(path), strict)
  save_model / load_model :1060-1085  'generator_param.pkl' / 'discriminator_param.pkl'
  plateau rule        :985-1036    best-metric bookkeeping, roll back to the best epoch and halve the learning rates
and adds what chain training x2 -> x3 -> x4 -> x8 -> x9 needs but the reference does by hand (commented-out paths at
:716-721): a shape-aware partial load between generators of different scale, and the optimiser state (the reference
never saves Adam's moments; resuming there restarts them from zero).
Files written here load in the reference unchanged and vice versa (same keys, plain tensors)."""
import os

import torch


def _unwrap(network):
    return network.module if isinstance(network, torch.nn.DataParallel) else network


def epoch_path(save_dir, network_label, iter_label):
    return os.path.join(save_dir, '{}_param_epoch_{}.pkl'.format(network_label, iter_label))


def save_epoch_network(save_dir, network, network_label, iter_label):
    """sradsgan.py:1041-1051."""
    os.makedirs(save_dir, exist_ok=True)
    state = {k: v.detach().cpu().clone() for k, v in _unwrap(network).state_dict().items()}
    path = epoch_path(save_dir, network_label, iter_label)
    torch.save(state, path)
    return path


def _after_load():
    # parameters are views into flat arenas and convs cache re-packed weights: loading writes in place, then the packs
    # must be refreshed (sradsgan_amd.ops is only needed when the HIP library is present)
    try:
        from . import ops
        ops.bump_weight_epoch()
    except Exception:                                   # CPU-only use (tests of the host logic)
        pass


def load_epoch_network(load_path, network, strict=True):
    """sradsgan.py:1054-1058.  Copies into the existing parameter storage (arena views stay valid)."""
    state = torch.load(load_path, map_location='cpu')
    result = _unwrap(network).load_state_dict(state, strict=strict)
    _after_load()
    return result


def save_model(save_dir, generator, discriminator, epoch=None):
    """sradsgan.py:1060-1072."""
    model_dir = os.path.join(save_dir, 'model')
    os.makedirs(model_dir, exist_ok=True)
    suffix = '_param.pkl' if epoch is None else '_param_epoch_%d.pkl' % epoch
    for label, net in (('generator', generator), ('discriminator', discriminator)):
        torch.save({k: v.detach().cpu().clone() for k, v in _unwrap(net).state_dict().items()},
                   os.path.join(model_dir, label + suffix))


def load_model(save_dir, generator):
    """sradsgan.py:1074-1085: generator only, strict=False; returns False when the file does not exist."""
    path = os.path.join(save_dir, 'model', 'generator_param.pkl')
    if not os.path.exists(path):
        return False
    _unwrap(generator).load_state_dict(torch.load(path, map_location='cpu'), strict=False)
    _after_load()
    return True


def load_compatible(network, state_dict):
    """Chain training: initialise `network` from a checkpoint of a different scale.  Every key present in both with
    the SAME shape is copied; keys missing on either side or with another shape (the 64 -> 64 r^2 upsampler conv when
    r changes between 2 and 3, the extra tied stage 'GAB_UP.upsampling.3.*' of the 2^n/3^n scales) keep their
    initialisation.  Returns (loaded, missing_in_checkpoint, shape_mismatch, unused_in_checkpoint) key lists."""
    net = _unwrap(network)
    own = net.state_dict()
    loaded, missing, mismatch = [], [], []
    with torch.no_grad():
        for k, dst in own.items():
            if k not in state_dict:
                missing.append(k)
            elif tuple(state_dict[k].shape) != tuple(dst.shape):
                mismatch.append(k)
            else:
                dst.copy_(state_dict[k])
                loaded.append(k)
    unused = [k for k in state_dict if k not in own]
    _after_load()
    return loaded, missing, mismatch, unused


def save_optimizer_state(path, step):
    """Adam moments and step counters of a TrainStep (both arenas) -- not in the reference; lets a resumed or
    rolled-back run continue with the moments it had."""
    state = {}
    for name, arena in (('G', step.arena_G), ('D', step.arena_D)):
        state[name] = dict(exp_avg=arena.exp_avg.detach().cpu().clone(), exp_avg_sq=arena.exp_avg_sq.detach().cpu().clone(),
                           step_state=arena.step_state.detach().cpu().clone(), numel=arena.numel)
    state['lr_G'], state['lr_D'] = step.lr_G, step.lr_D
    torch.save(state, path)


def load_optimizer_state(path, step):
    state = torch.load(path, map_location='cpu')
    for name, arena in (('G', step.arena_G), ('D', step.arena_D)):
        s = state[name]
        if s['numel'] != arena.numel:
            raise ValueError('optimizer state of %s has %d elements, the arena %d' % (name, s['numel'], arena.numel))
        arena.exp_avg.copy_(s['exp_avg'])
        arena.exp_avg_sq.copy_(s['exp_avg_sq'])
        arena.step_state.copy_(s['step_state'])
    step.lr_G, step.lr_D = state['lr_G'], state['lr_D']


class PlateauRollback:
    """The per-epoch control of sradsgan.py:985-1036 as an object.  update(epoch, psnr, ssim, ergas, lpips) after each
    validation returns the epoch index training continues from and whether a rollback happened; on a rollback the
    caller reloads 'generator_param_epoch_<best+1>.pkl' (the reference reloads the generator only) -- `on_rollback`
    does that when given -- and the learning rates are halved exactly as the reference does: G always, D only while
    lr < 1e-4 (checked BEFORE halving the tracked lr, :1022-1029)."""

    def __init__(self, lr, max_no_improve=5):
        self.lr = lr
        self.max_no_improve = max_no_improve
        self.psnr_max, self.ssim_max, self.ergas_min, self.lpips_min = 0, 0, 10000, 10000     # :795-798
        self.no_improve = 0
        self.best_step = 0
        self.history = []                                # (psnr, ssim, ergas, lpips) per epoch, popped on rollback

    def keep_training(self, epoch, num_epochs):
        return epoch < num_epochs and self.lr >= 0.00001   # :803

    def update(self, epoch, psnr, ssim, ergas, lpips, step=None, on_rollback=None):
        self.history.append((psnr, ssim, ergas, lpips))
        # first criterion that improves wins, in the reference's order (:985-1003)
        if psnr - self.psnr_max > 0:
            self.psnr_max, self.no_improve, self.best_step = psnr, 0, epoch
        elif ssim - self.ssim_max > 0:
            self.ssim_max, self.no_improve, self.best_step = ssim, 0, epoch
        elif ergas - self.ergas_min < 0:
            self.ergas_min, self.no_improve, self.best_step = ergas, 0, epoch
        elif lpips - self.lpips_min < 0:
            self.lpips_min, self.no_improve, self.best_step = lpips, 0, epoch
        else:
            self.no_improve += 1
        epoch += 1                                        # :1010
        rolled_back = False
        if self.no_improve >= self.max_no_improve:
            rolled_back = True
            if on_rollback is not None:
                on_rollback(self.best_step + 1)          # generator_param_epoch_<best+1>.pkl, :1012-1013
            if step is not None:
                step.lr_G /= 2.0                          # :1022-1024
                if self.lr < 0.0001:                      # :1025-1028
                    step.lr_D /= 2.0
            self.lr /= 2.0                                # :1029
            epoch = self.best_step + 1                    # :1030
            self.no_improve = 0
            for _ in range(self.max_no_improve):          # :1032-1036
                self.history.pop()
        return epoch, rolled_back
