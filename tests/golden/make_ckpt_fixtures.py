"""Reference-side evidence for checkpoints and the SVM probe from the LIVE reference.

Runs only where /root/reference exists.  Imports tools/builder.py and tools/runner_pretrain.py in place and lets
THEM do the work:

  1. builder.save_checkpoint (tools/builder.py:191-200) writes `ckpt-last.pth` for the published model
     (PointCAE_transformer_fc_global_folding_local, depth 2 + 1) wrapped in nn.DataParallel as the reference's
     non-distributed runner does (runner_pretrain.py:86-88: the saved keys carry a `module.` prefix), after one
     optimiser step of the reference's own AdamW groups (builder.build_opti_sche :38-101).  The weights come from
     tests/golden/weights.py (seed, key), so what is committed is the LAYOUT: key list, shapes, dtypes and a
     checksum per tensor (of the (seed, key) weights, restored after the step so that any machine can regenerate
     them), the optimiser's param_groups / state layout, epoch and metric records
     (tests/golden/ckpt_layout.json) -- not the 9 MB file.
  2. runner_pretrain.validate (:290-349) runs its FPS-resample -> return_feat -> LinearSVC loop on seeded labelled
     clouds; the feature matrices it fits and scores, and its accuracy, are stored (tests/golden/svm_probe_ref.npz).

    python tests/golden/make_ckpt_fixtures.py
"""
import importlib
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import as R  # noqa: E402
from weights import fill_state  # noqa: E402

CFG = 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'
MODEL = 'PointCAE_transformer_fc_global_folding_local'
SEED = 5


def checksum(t):
    t = t.detach().double().reshape(-1)
    return [float(t.sum()), float(t.abs().sum())]


def load_reference_tools():
    R.setup()
    R.cpu_cuda_noop()
    torch.Tensor.cuda = lambda self, *a, **k: self

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    sched = mod('timm.scheduler', CosineLRScheduler=lambda *a, **k: None)
    sys.modules['timm'].scheduler = sched
    mod('thop', profile=None, clever_format=None)
    mod('ptflops', get_model_complexity_info=None)
    import models.PointCAE_transformer as M
    sys.modules['models'].build_model_from_cfg = lambda cfg: getattr(M, cfg.NAME)(cfg)
    sys.modules['datasets'].build_dataset_from_cfg = lambda *a, **k: None
    tools = types.ModuleType('tools')
    tools.__path__ = [os.path.join(R.REF, 'tools')]
    sys.modules['tools'] = tools
    builder = importlib.import_module('tools.builder')
    tools.builder = builder
    runner = importlib.import_module('tools.runner_pretrain')
    return M, builder, runner


def main():
    import yaml
    M, builder, runner = load_reference_tools()
    from easydict import EasyDict                          # (the stub ref_import installs)
    from point_dae_amd.synthetic import labelled_clouds, shapenet_like_clouds
    full = EasyDict(yaml.safe_load(open(os.path.join(R.REF, CFG))))
    cfg = full.model
    cfg.NAME = MODEL
    cfg.transformer_config.depth = 2
    cfg.transformer_config.decoder_depth = 1
    R.seed_all(SEED)
    net = fill_state(builder.model_builder(cfg), SEED)

    # ---- 1. one optimiser step, then the reference's save_checkpoint / resume_model / load_model -----------
    wrapped = torch.nn.DataParallel(net)                   # runner_pretrain.py:86-88 (non-distributed launch)
    optimizer, _ = builder.build_opti_sche(wrapped, full)
    pts = torch.from_numpy(shapenet_like_clouds(2, 1024, seed=SEED))
    R.seed_all(SEED + 1)
    wrapped.train()
    l1, l2 = wrapped(pts, pts)
    (l1 + 0.005 * l2.sum()).backward()
    optimizer.step()
    fill_state(net, SEED)          # back to the (seed, key) weights: the stored checksums are then reproducible anywhere,
                                   # while the optimiser keeps the state entries its step created
    args = types.SimpleNamespace(local_rank=0, distributed=False, experiment_path=tempfile.mkdtemp())
    metrics, best = runner.Acc_Metric(0.8125), runner.Acc_Metric(0.875)
    builder.save_checkpoint(wrapped, optimizer, 41, metrics, best, 'ckpt-last', args)
    sd = torch.load(os.path.join(args.experiment_path, 'ckpt-last.pth'), map_location='cpu')
    layout = {
        'written_by': 'tools/builder.py:191-200 save_checkpoint (live reference), model %s depth 2+1, '
                      'weights tests/golden/weights.py seed %d + one AdamW step' % (MODEL, SEED),
        'top_level_keys': sorted(sd),
        'epoch': sd['epoch'], 'metrics': sd['metrics'], 'best_metrics': sd['best_metrics'],
        'base_model': {k: {'shape': list(v.shape), 'dtype': str(v.dtype).replace('torch.', ''), 'checksum': checksum(v)}
                       for k, v in sd['base_model'].items()},
        'optimizer': {
            'state_keys': sorted({k for st in sd['optimizer']['state'].values() for k in st}),
            'state_entries': len(sd['optimizer']['state']),
            'param_groups': [{k: (len(v) if k == 'params' else v) for k, v in g.items()}
                             for g in sd['optimizer']['param_groups']],
            # which parameter NAMES sit in which group, in group order (builder.py:41-98)
            'group_names': [],
        },
    }
    names = {id(p): n for n, p in wrapped.named_parameters()}
    for g in optimizer.param_groups:
        layout['optimizer']['group_names'].append([names[id(p)] for p in g['params']])
    # the reference's own readers on its own file
    fresh = builder.model_builder(cfg)
    start_epoch, best_metrics = builder.resume_model(fresh, args)
    assert start_epoch == 42 and best_metrics == {'acc': 0.875}
    assert all(torch.equal(a, b) for a, b in zip(fresh.state_dict().values(), net.state_dict().values()))
    layout['resume_model_returns'] = [start_epoch, best_metrics]
    json.dump(layout, open(os.path.join(HERE, 'ckpt_layout.json'), 'w'), indent=1, sort_keys=True)

    # ---- 2. validate(): FPS-resample -> return_feat -> LinearSVC on seeded labelled clouds -------------------
    net = fill_state(builder.model_builder(cfg), SEED)      # (the weights BEFORE the step: what weights.py gives)
    tr_x, tr_y = labelled_clouds(48, 1536, seed=31)        # 1536 -> FPS-resampled to 1024 by validate
    te_x, te_y = labelled_clouds(16, 1024, seed=32)        # 1024 -> 1024: FPS still re-ORDERS the cloud
    def loader(x, y, bs):
        return [('ModelNet', i, (torch.from_numpy(x[i:i + bs]), torch.from_numpy(y[i:i + bs])))
                for i in range(0, len(x), bs)]
    feats = []
    hook = net.register_forward_hook(lambda m, i, o: feats.append(o.detach().clone()))
    vcfg = EasyDict(dataset=EasyDict(extra_train=EasyDict(others=EasyDict(npoints=1024))))
    vargs = types.SimpleNamespace(distributed=False)
    R.seed_all(77)
    acc = runner.validate(net, loader(tr_x, tr_y, 16), loader(te_x, te_y, 16), 0, None, vargs, vcfg)
    hook.remove()
    feats = torch.cat(feats).numpy()
    assert feats.shape == (64, 384)
    np.savez_compressed(os.path.join(HERE, 'svm_probe_ref.npz'), train_features=feats[:48], test_features=feats[48:],
                        train_labels=tr_y, test_labels=te_y, acc=np.float64(acc.acc),
                        meta=np.array([31, 32, 48, 1536, 16, 1024, 77, SEED]))
    print('ckpt keys', len(layout['base_model']), 'groups', [g['params'] for g in layout['optimizer']['param_groups']],
          'svm acc', acc.acc, 'feature norm', float(np.abs(feats).mean()))


if __name__ == '__main__':
    main()
