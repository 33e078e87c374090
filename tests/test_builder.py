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


# ---- checkpoints against the LIVE reference's own writer (tests/golden/make_ckpt_fixtures.py) -------------------
import json   # noqa: E402
import os     # noqa: E402
import sys    # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
LAYOUT = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'ckpt_layout.json')))


def _published_model(seed=5):
    from weights import fill_state
    from point_dae_amd.config import cfg_from_yaml_file
    config = cfg_from_yaml_file(os.path.join(
        ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    config.model.NAME = 'PointCAE_transformer_fc_global_folding_local'
    config.model.transformer_config.depth = 2
    config.model.transformer_config.decoder_depth = 1
    torch.manual_seed(0)
    return fill_state(builder.model_builder(config.model), seed), config


def _checksum(t):
    t = t.detach().double().reshape(-1)
    return [float(t.sum()), float(t.abs().sum())]


def test_state_dict_is_the_reference_checkpoint_layout():
    """Every key the live reference's save_checkpoint wrote for this model (with nn.DataParallel's `module.` prefix,
    runner_pretrain.py:86-88) exists here with the same shape, dtype and -- filled by tests/golden/weights.py from
    (seed, key) -- the same checksum; nothing is missing, nothing is extra."""
    net, _ = _published_model()
    mine = net.state_dict()
    ref = {k[len('module.'):]: v for k, v in LAYOUT['base_model'].items()}
    assert all(k.startswith('module.') for k in LAYOUT['base_model'])
    assert sorted(mine) == sorted(ref)
    for k, rec in ref.items():
        assert list(mine[k].shape) == rec['shape'] and str(mine[k].dtype) == 'torch.' + rec['dtype'], k
        assert _checksum(mine[k]) == rec['checksum'], k


def test_reads_a_reference_written_checkpoint(tmp_path):
    """load_model / resume_model / load_pretrained_encoder on a file with the live reference's layout: top-level keys,
    `module.`-prefixed weights, dict-valued metric records, torch.optim.AdamW state (tools/builder.py:155-227)."""
    net, config = _published_model()
    opt_state = {'state': {}, 'param_groups': [dict(g, params=list(range(i * 47, i * 47 + g['params'])))
                                               for i, g in enumerate(LAYOUT['optimizer']['param_groups'])]}
    ckpt = {'base_model': {'module.' + k: v.clone() for k, v in net.state_dict().items()}, 'optimizer': opt_state,
            'epoch': LAYOUT['epoch'], 'metrics': LAYOUT['metrics'], 'best_metrics': LAYOUT['best_metrics']}
    assert sorted(ckpt) == LAYOUT['top_level_keys'] and sorted(ckpt['base_model']) == sorted(LAYOUT['base_model'])
    torch.save(ckpt, tmp_path / 'ckpt-last.pth')

    class Args:
        experiment_path, local_rank = str(tmp_path), 0
    fresh, _ = _published_model(seed=9)
    epoch, best = builder.resume_model(fresh, Args)
    assert [epoch, {'acc': best}] == LAYOUT['resume_model_returns']       # what the reference's resume_model returned
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), net.state_dict().values()))
    fresh2, _ = _published_model(seed=9)
    assert builder.load_model(fresh2, str(tmp_path / 'ckpt-last.pth')) == LAYOUT['epoch']
    assert torch.equal(fresh2.mask_token, net.mask_token)
    # the downstream remap (models/Point_MAE.py:643-656): MAE_encoder.* becomes the backbone
    from point_dae_amd.point_cae_transformer import MaskTransformer
    backbone = MaskTransformer(config.model)
    rec = builder.load_pretrained_encoder(backbone, str(tmp_path / 'ckpt-last.pth'))
    assert not rec.missing_keys
    assert torch.equal(backbone.encoder.first_conv[0].weight, net.MAE_encoder.encoder.first_conv[0].weight)


def test_writes_the_reference_checkpoint_layout(tmp_path):
    """save_checkpoint (builder.py:191-200): same top-level keys, metric records as dicts, torch.optim.AdamW layout
    with the reference's two groups (no-decay first) holding the same parameter NAMES in the same order."""
    from point_dae_amd.svm_probe import Acc_Metric
    net, config = _published_model()
    opt, _ = builder.build_opti_sche(net, config)
    out = net.coarse_pred[0].weight.sum() + sum(p.sum() for p in net.parameters()) * 0
    out.backward()
    opt.step()

    class Args:
        experiment_path, local_rank = str(tmp_path), 0
    builder.save_checkpoint(net, opt, 41, Acc_Metric(0.8125), Acc_Metric(0.875), 'ckpt-last', Args)
    sd = torch.load(tmp_path / 'ckpt-last.pth', map_location='cpu')
    assert sorted(sd) == LAYOUT['top_level_keys']
    assert sd['epoch'] == LAYOUT['epoch'] and sd['metrics'] == LAYOUT['metrics'] and sd['best_metrics'] == LAYOUT['best_metrics']
    assert sorted('module.' + k for k in sd['base_model']) == sorted(LAYOUT['base_model'])
    ref_groups = LAYOUT['optimizer']['param_groups']
    assert [len(g['params']) for g in sd['optimizer']['param_groups']] == [g['params'] for g in ref_groups]
    for g, r in zip(sd['optimizer']['param_groups'], ref_groups):
        for key in ('lr', 'weight_decay', 'eps', 'amsgrad'):
            assert g[key] == r[key], key
        assert list(g['betas']) == r['betas']
    names = {id(p): n for n, p in net.named_parameters()}
    mine = [['module.' + names[id(p)] for p in g['params']] for g in opt.param_groups]
    assert mine == LAYOUT['optimizer']['group_names']
    assert sorted({k for st in sd['optimizer']['state'].values() for k in st}) == LAYOUT['optimizer']['state_keys']
    assert len(sd['optimizer']['state']) == LAYOUT['optimizer']['state_entries']


def test_lambda_lr_schedule_and_bn_momentum_schedule():
    """builder.py:117-118 -> utils/misc.py:26-32: lr(e) = lr * max(lr_decay ** (e / decay_step), lowest_decay);
    builder.py:147-151 -> utils/misc.py:34-40, :97-127: the BatchNorm momentum schedule beside it, as a [lr, bn] list."""
    from point_dae_amd.config import cfg_from_yaml_file
    config = cfg_from_yaml_file(os.path.join(
        ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    config.scheduler.type = 'LambdaLR'
    config.scheduler.kwargs = type(config.scheduler.kwargs)(decay_step=21, lr_decay=0.76, lowest_decay=0.02)
    net = torch.nn.Linear(4, 4)
    opt, sch = builder.build_opti_sche(net, config)
    lr = config.optimizer.kwargs.lr
    for e in (0, 1, 21, 100, 400):
        sch.step(e)
        assert opt.param_groups[0]['lr'] == pytest.approx(lr * max(0.76 ** (e / 21), 0.02), rel=1e-12)
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4), torch.nn.Sequential(torch.nn.BatchNorm2d(2)))
    config.bnmscheduler = type(config.scheduler)(
        type='Lambda', kwargs=type(config.scheduler.kwargs)(decay_step=21, bn_decay=0.5, bn_momentum=0.9, lowest_decay=0.01))
    opt, sch = builder.build_opti_sche(net, config)
    assert isinstance(sch, list) and len(sch) == 2
    bns = [m for m in net.modules() if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d))]
    assert len(bns) == 2 and all(m.momentum == 0.9 for m in bns)           # the constructor steps to epoch 0 (:114)
    calls = []
    sch[1].listeners.append(lambda: calls.append(sch[1].current))
    for e in (0, 1, 21, 42, 400, 401):
        for item in sch:                                                       # runner_pretrain.py:237-241
            item.step(e)
        want = max(0.9 * 0.5 ** (e / 21), 0.01)
        assert all(m.momentum == pytest.approx(want, rel=1e-12) for m in bns)
        assert sch[1].get_momentum(e) == pytest.approx(want, rel=1e-12)
    assert len(calls) == 4 and calls[-1] == 0.01                              # epoch 0 and 401 changed nothing: no re-capture
    config.bnmscheduler.type = 'Step'
    with pytest.raises(NotImplementedError, match='bnmscheduler'):
        builder.build_opti_sche(net, config)
