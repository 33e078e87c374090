"""utils/misc.py of the reference: fps(), set_random_seed(), the BatchNorm-momentum schedule."""
import random

import numpy as np
import torch
import torch.nn as nn

from . import pointnet2_utils


def fps(data, number):
    """data (B,N,3|6) -> (idx (B,number) i32, data[idx] (B,number,C)); utils/misc.py:13-20."""
    xyz = data[:, :, :3].contiguous()
    if data.shape[2] == 3:
        return pointnet2_utils.furthest_point_sample_with_centres(xyz, number)
    idx = pointnet2_utils.furthest_point_sample(xyz, number)
    out = pointnet2_utils.gather_operation(data.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
    return idx, out


def set_random_seed(seed, deterministic=False):
    """utils/misc.py:42-66: python, numpy, torch (CPU + all GPUs)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    if deterministic:
        # the reference's --deterministic pins cuDNN (utils/misc.py:56-62); here the hand-written kernels'
        # atomic reductions become ordered ones (include/pdae.h: pdae_set_deterministic)
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
        if torch.cuda.is_available():
            from . import _lib
            if not _lib.deterministic():
                _lib.set_deterministic(True)


def set_bn_momentum_default(bn_momentum):
    """utils/misc.py:91-95."""
    def fn(m):
        if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
            m.momentum = bn_momentum
    return fn


class BNMomentumScheduler(object):
    """utils/misc.py:97-127: momentum(epoch) = bn_lambda(epoch) written into every BatchNorm of the model by step().
    The captured step graphs hold the momentum as a kernel argument of their statistics kernels, so a step() that
    CHANGES the value calls `listeners` (the graphed steps register their invalidate(): the next step re-captures)."""

    def __init__(self, model, bn_lambda, last_epoch=-1, setter=set_bn_momentum_default):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.model, self.setter, self.lmbd = model, setter, bn_lambda
        self.listeners, self.current = [], None
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        value = self.lmbd(epoch)
        self.model.apply(self.setter(value))
        if value != self.current:
            self.current = value
            for fn in self.listeners:
                fn()

    def get_momentum(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        return self.lmbd(epoch)


def build_lambda_bnsche(model, config):
    """utils/misc.py:34-40: momentum(e) = max(bn_momentum * bn_decay ** (e / decay_step), lowest_decay)."""
    if config.get('decay_step') is None:
        raise NotImplementedError('bnmscheduler without decay_step')
    return BNMomentumScheduler(
        model, lambda e: max(config.bn_momentum * config.bn_decay ** (e / config.decay_step), config.lowest_decay))
