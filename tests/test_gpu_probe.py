"""SVM-probe validation against the LIVE reference's validate() (tools/runner_pretrain.py:290-349).

tests/golden/svm_probe_ref.npz (make_ckpt_fixtures.py) holds what the reference's own loop produced for seeded labelled
clouds: the feature matrix it fitted its LinearSVC on, the one it scored, and the accuracy.  Here the product's
validate() -- FPS-resample on the gfx950 kernel, `return_feat` of the published model on the HIP path, the same host RNG
draws (corruption, mask) -- must give those features and that accuracy."""
import os
import random
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def test_validate_reproduces_the_reference_features_and_accuracy():
    from weights import fill_state
    from point_dae_amd import builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.svm_probe import evaluate_svm, validate
    from point_dae_amd.synthetic import labelled_clouds
    f = np.load(os.path.join(ROOT, 'tests', 'golden', 'svm_probe_ref.npz'))
    s_tr, s_te, n_tr, p_tr, n_te, p_te, seed, wseed = (int(v) for v in f['meta'])
    config = cfg_from_yaml_file(os.path.join(
        ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    config.model.NAME = 'PointCAE_transformer_fc_global_folding_local'
    config.model.transformer_config.depth = 2
    config.model.transformer_config.decoder_depth = 1
    net = fill_state(builder.model_builder(config.model), wseed).cuda()
    tr_x, tr_y = labelled_clouds(n_tr, p_tr, seed=s_tr)
    te_x, te_y = labelled_clouds(n_te, p_te, seed=s_te)
    assert np.array_equal(tr_y, f['train_labels']) and np.array_equal(te_y, f['test_labels'])

    def loader(x, y, bs=16):
        return [('ModelNet', i, (torch.from_numpy(x[i:i + bs]), torch.from_numpy(y[i:i + bs])))
                for i in range(0, len(x), bs)]
    feats = []
    hook = net.register_forward_hook(lambda m, i, o: feats.append(o.detach().clone()))

    class _Cfg:                                            # config.dataset.extra_train.others.npoints
        class dataset:
            class extra_train:
                class others:
                    npoints = 1024
    random.seed(seed), np.random.seed(seed), torch.manual_seed(seed)      # the reference ran under seed_all(seed)
    net.train()
    metric = validate(net, loader(tr_x, tr_y), loader(te_x, te_y), 0, _Cfg, log=lambda *a: None)
    hook.remove()
    assert net.training                                     # validate() restores the mode
    got = torch.cat(feats).cpu().numpy()
    want = np.concatenate([f['train_features'], f['test_features']])
    assert got.shape == want.shape == (n_tr + n_te, 384)
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err <= 1e-4, err
    assert metric.acc == pytest.approx(float(f['acc']), abs=1e-12)
    # and the reference's accuracy is what its stored features give under the same sklearn
    assert evaluate_svm(f['train_features'], f['train_labels'], f['test_features'], f['test_labels']) == float(f['acc'])
