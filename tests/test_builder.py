"""Host-side builders against closed forms (CPU): the CosLR schedule of tools/builder.py:111-129 is timm 0.4.5's
CosineLRScheduler(t_initial, t_mul=1, lr_min, decay_rate=0.1, warmup_lr_init, warmup_t, cycle_limit=1,
t_in_epochs=True); timm is not in the image, so the expected values are its published formula written out
here independently of point_dae_amd.builder:

    t <  warmup_t              : warmup_lr_init + t (lr - warmup_lr_init) / warmup_t
    warmup_t <= t < t_initial  : lr_min + (lr - lr_min)/2 (1 + cos(pi t / t_initial))      (warmup_prefix False)
    t >= t_initial             : lr_min * decay_rate ** cycle_limit
"""
import math

import pytest
import torch

from point_dae_amd import builder


def _timm_cosine(t, lr, t_initial, lr_min, decay_rate=0.1, warmup_t=0, warmup_lr_init=0.0, cycle_limit=1):
    if t < warmup_t:
        return warmup_lr_init + t * (lr - warmup_lr_init) / warmup_t
    i = t // t_initial
    t_curr = t - t_initial * i
    gamma = decay_rate ** i
    if i < cycle_limit:
        return lr_min * gamma + 0.5 * (lr * gamma - lr_min * gamma) * (1 + math.cos(math.pi * t_curr / t_initial))
    return lr_min * decay_rate ** cycle_limit


def _opt(lrs=(1e-3, 1e-3)):
    ps = [torch.nn.Parameter(torch.zeros(1)) for _ in lrs]
    return torch.optim.AdamW([{'params': [p], 'lr': lr} for p, lr in zip(ps, lrs)])


@pytest.mark.parametrize('warmup_t', [0, 10])
def test_cosine_lr_matches_timm_closed_form(warmup_t):
    opt = _opt((1e-3, 1e-4))                                   # two groups (diff_lr): each follows its own base value
    sch = builder.CosineLRScheduler(opt, t_initial=300, lr_min=1e-6, decay_rate=0.1, warmup_t=warmup_t,
                                    warmup_lr_init=1e-6)
    if warmup_t:                                               # timm sets the warm-up start value at construction
        assert [g['lr'] for g in opt.param_groups] == [1e-6, 1e-6]
    else:
        assert [g['lr'] for g in opt.param_groups] == [1e-3, 1e-4]
    for t in (0, 1, 5, 9, 10, 11, 150, 299, 300, 301, 599, 600, 900):
        sch.step(t)
        for g, base in zip(opt.param_groups, (1e-3, 1e-4)):
            want = _timm_cosine(t, base, 300, 1e-6, 0.1, warmup_t, 1e-6)
            assert g['lr'] == pytest.approx(want, rel=1e-12, abs=0), (t, g['lr'], want)
    # the anchor values themselves: base lr at t = 0, the midpoint at t = 150, lr_min * 0.1 after the cycle
    sch.step(150)
    assert opt.param_groups[0]['lr'] == pytest.approx(1e-6 + 0.5 * (1e-3 - 1e-6), rel=1e-12)
    sch.step(300)
    assert opt.param_groups[0]['lr'] == pytest.approx(1e-7, rel=1e-12)


def test_cosine_lr_as_the_runner_steps_it():
    """runner_pretrain.py:237-241 calls scheduler.step(epoch) AFTER epoch `epoch`: epochs 0 and 1 both train at the
    base lr, epoch e >= 1 at lr(e - 1)."""
    opt = _opt((1e-3,))
    sch = builder.CosineLRScheduler(opt, t_initial=300, lr_min=1e-6)
    seen = []
    for epoch in range(0, 4):
        seen.append(opt.param_groups[0]['lr'])
        sch.step(epoch)
    assert seen[0] == seen[1] == 1e-3
    assert seen[2] == pytest.approx(_timm_cosine(1, 1e-3, 300, 1e-6), rel=1e-12)
    assert seen[3] == pytest.approx(_timm_cosine(2, 1e-3, 300, 1e-6), rel=1e-12)


def test_build_opti_sche_reads_the_reference_yaml(tmp_path):
    """builder.py:109-129: min_lr defaults to lr / 1000, t_max to scheduler.kwargs.epochs, no warm-up (F8)."""
    import os
    from point_dae_amd.config import cfg_from_yaml_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = cfg_from_yaml_file(os.path.join(
        root, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.LayerNorm(4))
    opt, sch = builder.build_opti_sche(net, config)
    assert isinstance(sch, builder.CosineLRScheduler)
    lr = config.optimizer.kwargs.lr
    assert sch.t_initial == config.scheduler.kwargs.epochs and sch.warmup_t == 0
    assert sch.lr_min == pytest.approx(lr / 1000.)
    # the two AdamW groups of builder.py:41-98: 1-D tensors and biases without weight decay
    wd = {g['weight_decay']: sum(p.numel() for p in g['params']) for g in opt.param_groups}
    assert wd == {0.0: 4 + 4 + 4, config.optimizer.kwargs.weight_decay: 16}
