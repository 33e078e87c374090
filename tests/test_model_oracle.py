"""CPU tests: the plain-PyTorch oracle model (oracle/model.py) against the golden
fixtures that were generated from the LIVE reference classes
(tests/golden/make_fixtures.py), and -- when the reference tree is present, i.e.
in the dev container only -- against the live reference itself."""
import os
import sys

import numpy as np
import pytest
import torch

from golden_util import check_grads, fill_state, load_fixture, model_cfg

FIXTURES = ['transformer_cfg3_b2.npz', 'transformer_allpatch_cdl1_b3.npz', 'transformer_folding_b2.npz',
            'transformer_nomask_b2.npz']


def _osteps(steps):
    B = steps.shape[1]
    return [('mul', s[:, 1:4]) if s[0, 0] == 0 else ('mat', s[:, 1:].reshape(B, 3, 3))
            for s in torch.from_numpy(steps)]


@pytest.mark.parametrize('name', FIXTURES)
def test_oracle_model_reproduces_reference_fixture(name):
    from oracle import model as OM
    fx = load_fixture(name)
    cfg = model_cfg(fx)
    torch.manual_seed(0)
    model = fill_state(getattr(OM, str(fx['cls']))(cfg), int(fx['seed'])).train()
    pts = torch.from_numpy(fx['pts'])
    cap = {}
    loss, loss2 = model(pts, pts, mask=torch.from_numpy(fx['mask']), steps=_osteps(fx['steps']), capture=cap)
    (loss + 0.005 * loss2.sum()).backward()
    # same machine class, same torch build -> the restatement is bit-identical
    assert abs(loss.item() - float(fx['loss'])) <= 1e-6 * abs(float(fx['loss']))
    np.testing.assert_array_equal(cap['center'].numpy(), fx['center'])
    np.testing.assert_allclose(cap['x_vis'].detach().numpy(), fx['x_vis'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(cap['x_rec'].detach().numpy(), fx['x_rec'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(loss2.detach().numpy().reshape(-1), fx['loss2'], rtol=1e-6)
    check_grads(model, fx, 1e-4, name)
    for bname, b in model.named_buffers():
        if b.dtype.is_floating_point:
            np.testing.assert_allclose(b.numpy(), fx['buf/' + bname], rtol=1e-5, atol=1e-6)


def test_corruption_and_mask_draws_follow_reference_rng():
    """Product-side host draws == oracle-side draws (which were bit-identical to
    the live reference when the fixtures were made) under the same seeds."""
    import random
    from oracle import model as OM
    from point_dae_amd import corrupt_util_tensor as C
    from point_dae_amd.point_cae_transformer import draw_mask

    def seed(s):
        random.seed(s), np.random.seed(s), torch.manual_seed(s)
    kinds = set()
    for s in range(40):
        seed(s)
        a = OM.draw_corruption(['affine_r3', 'Drop-Patch'], 4)
        ma = OM.draw_mask(4, 64, 0.6, 'True')
        seed(s)
        b = C.draw_corruption(['affine_r3', 'Drop-Patch'], 4)
        mb, ratio = draw_mask(4, 64, 0.6, 'True')
        assert len(a) == b.shape[0] and 1 <= len(a) <= 3
        for (kind, p), row in zip(a, b):
            kinds.add(kind)
            if kind == 'mul':
                assert (row[:, 0] == 0).all() and torch.equal(row[:, 1:4], p)
            else:
                assert (row[:, 0] == 1).all() and torch.equal(row[:, 1:].reshape(-1, 3, 3), p)
        assert torch.equal(ma, mb) and 0.5 <= ratio <= 0.8
        assert (mb.sum(1) == int(ratio * 64)).all()
    assert kinds == {'mul', 'mat'}
    assert C.draw_corruption(['clean', 'Drop-Patch'], 3).shape == (0, 3, 10)
    with pytest.raises(NotImplementedError):
        C.draw_corruption(['jitter'], 3)


@pytest.mark.skipif(not os.path.isdir('/root/reference/models'), reason='reference tree only exists in the dev container')
def test_oracle_model_matches_live_reference():
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import ref_import as R
    R.setup()
    R.cpu_cuda_noop()
    import models.PointCAE_transformer as M
    from oracle import model as OM
    from point_dae_amd.synthetic import shapenet_like_clouds
    fx = load_fixture(FIXTURES[1])
    cfg = model_cfg(fx)
    cfg.transformer_config.drop_path_rate = 0.1          # stochastic depth ON: same RNG stream both sides
    from easydict import EasyDict
    ref = fill_state(M.PointCAE_transformer(EasyDict(cfg)), 5).train()
    mine = fill_state(OM.PointCAE_transformer(cfg), 5).train()
    x = torch.from_numpy(shapenet_like_clouds(2, 1024, seed=9))
    R.seed_all(77)
    l_ref, _ = ref(x, x)
    l_ref.backward()
    R.seed_all(77)
    l_my, _ = mine(x, x)
    l_my.backward()
    assert l_ref.item() == l_my.item()
    gr = dict(ref.named_parameters())
    for n, p in mine.named_parameters():
        assert torch.equal(p.grad, gr[n].grad), n


def test_oracle_pointnetv2_reproduces_reference_fixture():
    """BASELINE config 1 (pretrain_PointCAE_clean.yaml, B=2, N=1024, Point_CAE_PointNetv2)."""
    from oracle import model as OM
    from point_dae_amd.config import cfg_from_yaml_file
    fx = load_fixture('pointnetv2_cfg1_b2.npz')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = cfg_from_yaml_file(os.path.join(root, 'cfgs', 'pretrain_PointCAE_clean.yaml')).model
    model = fill_state(OM.Point_CAE_PointNetv2(cfg), int(fx['seed'])).train()
    cap = {}
    lc, lf = model(torch.from_numpy(fx['corrupted']), torch.from_numpy(fx['clean']), capture=cap)
    (lc + 0.5 * lf).backward()
    assert abs(lc.item() - float(fx['loss_coarse'])) <= 1e-6 * abs(float(fx['loss_coarse']))
    assert abs(lf.item() - float(fx['loss_fine'])) <= 1e-6 * abs(float(fx['loss_fine']))
    np.testing.assert_allclose(cap['feature'].detach().numpy(), fx['feature'], rtol=1e-4, atol=1e-5)
    check_grads(model, fx, 2e-4, 'pointnetv2')


def test_oracle_pointnetv2_dropout_global_fixture():
    """pretrain_PointCAE_dropout_global.yaml's in-forward corruption (dropout_global_random: a random half of every
    cloud) -- fixture from the live reference, host RNG re-seeded with seed + 7 right before the forward."""
    from oracle import model as OM
    from point_dae_amd.config import cfg_from_yaml_file
    fx = load_fixture('pointnetv2_dropout_global_b2.npz')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = cfg_from_yaml_file(os.path.join(root, 'cfgs', 'pretrain_PointCAE_clean.yaml')).model
    cfg.corrupt_type = ['dropout_global']
    model = fill_state(OM.Point_CAE_PointNetv2(cfg), int(fx['seed'])).train()
    torch.manual_seed(int(fx['seed']) + 7)
    lc, lf = model(torch.from_numpy(fx['corrupted']), torch.from_numpy(fx['clean']))
    (lc + 0.5 * lf).backward()
    assert abs(lc.item() - float(fx['loss_coarse'])) <= 1e-6 * abs(float(fx['loss_coarse']))
    assert abs(lf.item() - float(fx['loss_fine'])) <= 1e-6 * abs(float(fx['loss_fine']))
    check_grads(model, fx, 2e-4, 'pointnetv2 dropout_global')


def test_oracle_dgcnn_reproduces_reference_fixture():
    """Point_CAE_DGCNN_FCOnly (the published non-Transformer model; fixture from the live reference)."""
    from oracle import model as OM
    from point_dae_amd.config import cfg_from_yaml_file
    fx = load_fixture('dgcnn_fconly_b2.npz')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = cfg_from_yaml_file(os.path.join(root, 'cfgs', 'pretrain_PointCAE_clean.yaml')).model
    model = fill_state(OM.Point_CAE_DGCNN_FCOnly(cfg), int(fx['seed'])).train()
    loss, zero = model(torch.from_numpy(fx['corrupted']), torch.from_numpy(fx['clean']))
    loss.backward()
    assert abs(loss.item() - float(fx['loss'])) <= 1e-6 * abs(float(fx['loss'])) and zero.item() == 0
    check_grads(model, fx, 2e-4, 'dgcnn')
    feat = model.eval()(None, torch.from_numpy(fx['clean']), return_feat=True)
    assert feat.shape == (int(fx['B']), 1024)


@pytest.mark.parametrize('name,items', [('dgcnn_dropout_patch_b2.npz', ['dropout_patch_pointmae']),
                                        ('dgcnn_dropout_global_p3_b2.npz', ['dropout_global_p3']),
                                        ('dgcnn_random_dropout_b2.npz', ['random_dropout'])])
def test_oracle_dgcnn_in_forward_dropouts(name, items):
    """models/PointCAE_DGCNN.py:198-221: the dropouts applied inside forward (datasets/corrupt_util.py:572-588 random
    share of every cloud, :900-924 Point-MAE style patch drop over 64 FPS centres x 32 neighbours) -- fixtures from the
    live reference with python's and torch's host generators seeded seed + 7 right before the forward."""
    import random
    from oracle import model as OM
    from point_dae_amd.config import cfg_from_yaml_file
    fx = load_fixture(name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = cfg_from_yaml_file(os.path.join(root, 'cfgs', 'pretrain_PointCAE_clean.yaml')).model
    cfg.corrupt_type = list(items)
    model = fill_state(OM.Point_CAE_DGCNN_FCOnly(cfg), int(fx['seed'])).train()
    random.seed(int(fx['seed']) + 7), torch.manual_seed(int(fx['seed']) + 7)
    loss, zero = model(torch.from_numpy(fx['corrupted']), torch.from_numpy(fx['clean']))
    loss.backward()
    assert abs(loss.item() - float(fx['loss'])) <= 1e-6 * abs(float(fx['loss']))
    check_grads(model, fx, 2e-4, name)


def test_oracle_pointnetv2_dropout_patch_fixture():
    """models/PointCAE_pointnetv2.py:143-145: the patch drop inside Point_CAE_PointNetv2.forward (live fixture)."""
    import random
    from oracle import model as OM
    from point_dae_amd.config import cfg_from_yaml_file
    fx = load_fixture('pointnetv2_dropout_patch_b2.npz')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = cfg_from_yaml_file(os.path.join(root, 'cfgs', 'pretrain_PointCAE_clean.yaml')).model
    cfg.corrupt_type = ['dropout_patch_pointmae']
    model = fill_state(OM.Point_CAE_PointNetv2(cfg), int(fx['seed'])).train()
    random.seed(int(fx['seed']) + 7), torch.manual_seed(int(fx['seed']) + 7)
    lc, lf = model(torch.from_numpy(fx['corrupted']), torch.from_numpy(fx['clean']))
    (lc + 0.5 * lf).backward()
    assert abs(lc.item() - float(fx['loss_coarse'])) <= 1e-6 * abs(float(fx['loss_coarse']))
    assert abs(lf.item() - float(fx['loss_fine'])) <= 1e-6 * abs(float(fx['loss_fine']))
    check_grads(model, fx, 2e-4, 'pointnetv2 dropout_patch')
