"""Model / optimiser / scheduler / checkpoint builders of the pretraining path
(tools/builder.py of the reference: model_builder :34-36, build_opti_sche
:38-153, resume_model :155-178, save_checkpoint :191-200, load_model :202-227).
"""
import math
import os

import torch

from . import point_cae_dgcnn, point_cae_pointnetv2, point_cae_transformer  # noqa: F401  (register the models)
from .registry import build_model_from_cfg


def model_builder(config):
    return build_model_from_cfg(config)


def _unwrap(model):
    return model.module if hasattr(model, 'module') else model


def add_weight_decay(model, weight_decay, part='all', lr=None, skip_list=()):
    """AdamW parameter groups (builder.py:41-98): 1-D tensors, '.bias' and names
    containing 'token' get no weight decay."""
    def no_wd(name, p):
        return p.dim() == 1 or name.endswith('.bias') or 'token' in name or name in skip_list

    def chosen(name):
        if part == 'all':
            return True
        if part == 'only_new':
            return 'cls' in name
        if part == 'decoder':
            return ('decoder_pos_embed' in name) or ('MAE_decoder' in name) or ('increase_dim' in name)
        if part == 'diff_lr':
            return True
        raise NotImplementedError(part)

    decay, no_decay, decay_pre, no_decay_pre = [], [], [], []
    for name, p in _unwrap(model).named_parameters():
        if not p.requires_grad or not chosen(name):
            continue
        pre = part == 'diff_lr' and 'cls' not in name
        if no_wd(name, p):
            (no_decay_pre if pre else no_decay).append(p)
        else:
            (decay_pre if pre else decay).append(p)
    groups = []
    if part == 'diff_lr':
        groups += [{'params': no_decay_pre, 'weight_decay': 0., 'lr': lr * 0.1},
                   {'params': decay_pre, 'weight_decay': weight_decay, 'lr': lr * 0.1}]
    groups += [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': weight_decay}]
    return groups


class CosineLRScheduler:
    """timm 0.4.5 CosineLRScheduler as the reference configures it
    (builder.py:120-130: t_mul 1, cycle_limit 1, decay_rate 0.1, epoch units):
    lr(t) = lr_min + (lr - lr_min)/2 (1 + cos(pi t / t_initial)) for t < t_initial,
    lr_min * decay_rate afterwards; linear warm-up from warmup_lr_init over
    warmup_t epochs (0 in every shipped config).  `step(epoch)`
    is called after each epoch with that epoch's number (runner :237-241)."""

    def __init__(self, optimizer, t_initial, lr_min=0., decay_rate=0.1, warmup_t=0, warmup_lr_init=0.):
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.decay_rate = t_initial, lr_min, decay_rate
        self.warmup_t, self.warmup_lr_init = warmup_t, warmup_lr_init
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_values = [g['initial_lr'] for g in optimizer.param_groups]
        if warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values]
            self._set([warmup_lr_init] * len(self.base_values))

    def _set(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g['lr'] = v

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        i = t // self.t_initial
        if i < 1:                          # cycle_limit = 1: one cosine cycle (gamma = decay_rate ** 0 = 1)
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial))
                    for v in self.base_values]
        return [self.lr_min * self.decay_rate for _ in self.base_values]      # lr_min * decay_rate ** cycle_limit

    def step(self, epoch):
        self._set(self._get_lr(epoch))


def build_opti_sche(base_model, config):
    oc = config.optimizer
    if oc.type != 'AdamW':
        raise NotImplementedError(oc.type)
    from .data_parallel import FlatDataParallel
    if (isinstance(base_model, FlatDataParallel) and base_model.flat_param.is_cuda
            and oc.get('part', 'all') == 'all'):
        from .optim import FlatAdamW            # two fused launches over the flat buffers
        optimizer = FlatAdamW(base_model, **oc.kwargs)
    else:
        groups = add_weight_decay(base_model, oc.kwargs.weight_decay, part=oc.get('part', 'all'), lr=oc.kwargs.lr)
        optimizer = torch.optim.AdamW(groups, **oc.kwargs)
    sc = config.scheduler
    kw = sc.kwargs
    if sc.type == 'CosLR':
        min_lr = kw.min_lr if kw.get('min_lr', False) else oc.kwargs.lr / 1000.
        scheduler = CosineLRScheduler(optimizer, t_initial=kw.get('t_max', kw.epochs), lr_min=min_lr,
                                      decay_rate=0.1, warmup_lr_init=kw.get('warmup_lr', 1.0e-6),
                                      warmup_t=kw.get('warmup_epochs', 0))
    elif sc.type == 'LambdaLR':
        # utils/misc.py:26-32 build_lambda_sche: lr x max(lr_decay ** (epoch / decay_step), lowest_decay)
        if kw.get('decay_step') is None:
            raise NotImplementedError('LambdaLR without decay_step')
        scheduler = torch.optim.lr_scheduler.LambdaLR(
            optimizer, lambda e: max(kw.lr_decay ** (e / kw.decay_step), kw.lowest_decay))
    elif sc.type == 'StepLR':
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, **kw)
    elif sc.type == 'function':
        scheduler = None
    else:
        raise NotImplementedError(sc.type)
    if config.get('bnmscheduler') is not None:
        # builder.py:147-151: a BatchNorm-momentum schedule beside the learning-rate one (utils/misc.py:34-40, :97-127);
        # the runner steps both per epoch and lets a changed momentum invalidate the captured step graphs
        from .misc import build_lambda_bnsche
        bc = config.bnmscheduler
        if bc.type != 'Lambda':
            raise NotImplementedError('bnmscheduler type %r' % (bc.type,))
        scheduler = [scheduler, build_lambda_bnsche(base_model, bc.kwargs)]
    return optimizer, scheduler


def save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, prefix, args, logger=None):
    """rank-0 torch.save of the reference's dict layout (builder.py:191-200)."""
    if getattr(args, 'local_rank', 0) != 0:
        return
    state = {'base_model': _unwrap(base_model).state_dict(), 'optimizer': optimizer.state_dict(),
             'epoch': epoch,
             'metrics': metrics.state_dict() if hasattr(metrics, 'state_dict') else (metrics or dict()),
             'best_metrics': best_metrics.state_dict() if hasattr(best_metrics, 'state_dict') else (best_metrics or dict())}
    os.makedirs(args.experiment_path, exist_ok=True)
    torch.save(state, os.path.join(args.experiment_path, prefix + '.pth'))


def _strip_module(sd):
    return {k.replace('module.', ''): v for k, v in sd.items()}


def load_model(base_model, ckpt_path, logger=None):
    """builder.py:202-227: accepts {'model': ...} or {'base_model': ...}, strips 'module.'."""
    if not os.path.exists(ckpt_path):
        raise NotImplementedError('no checkpoint file from path %s...' % ckpt_path)
    sd = torch.load(ckpt_path, map_location='cpu')
    if sd.get('model') is not None:
        base = _strip_module(sd['model'])
    elif sd.get('base_model') is not None:
        base = _strip_module(sd['base_model'])
    else:
        raise RuntimeError('mismatch of ckpt weight')
    _unwrap(base_model).load_state_dict(base, strict=True)
    return sd.get('epoch', -1)


def remap_pretrain_keys(state_dict):
    """The key surgery the reference's downstream models apply to a pretraining checkpoint
    (models/Point_MAE.py:643-656): strip 'module.', then 'MAE_encoder.<k>' -> '<k>' and
    'base_model.<k>' -> '<k>' (the encoder of the auto-encoder becomes the backbone)."""
    out = {}
    for k, v in _strip_module(state_dict).items():
        if k.startswith('MAE_encoder.'):
            k = k[len('MAE_encoder.'):]
        elif k.startswith('base_model.'):
            k = k[len('base_model.'):]
        out[k] = v
    return out


def load_pretrained_encoder(module, ckpt_path):
    """Load a `ckpt-*.pth` of the pretraining runner into a backbone (a MaskTransformer, or any module
    whose keys follow the remapped names) with strict=False, as load_model_from_ckpt does.
    -> the incompatible-keys record (missing_keys, unexpected_keys)."""
    ckpt = torch.load(ckpt_path, map_location='cpu')
    return module.load_state_dict(remap_pretrain_keys(ckpt['base_model']), strict=False)


def resume_model(base_model, args, logger=None):
    """builder.py:155-178: weights + epoch + best metric from ckpt-last.pth
    (the reference does not restore the optimiser state either, runner :92-93)."""
    ckpt_path = os.path.join(args.experiment_path, 'ckpt-last.pth')
    if not os.path.exists(ckpt_path):
        return 0, 0
    sd = torch.load(ckpt_path, map_location='cpu')
    _unwrap(base_model).load_state_dict(_strip_module(sd['base_model']), strict=True)
    best = sd.get('best_metrics', 0)
    if isinstance(best, dict):
        best = best.get('acc', 0)
    return sd['epoch'] + 1, best
