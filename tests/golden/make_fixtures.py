"""Generate the golden fixtures of tests/golden/ from the LIVE reference.

Runs only in the dev container (needs /root/reference, imported read-only via
ref_import.py with the native ops replaced by the CPU oracle).  What is
committed is data: inputs, the captured random draws, and the reference's
outputs -- never reference source.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_import as R          # noqa: E402
from weights import fill_state  # noqa: E402


def _sample(t, n=256):
    flat = t.detach().reshape(-1)
    idx = np.linspace(0, flat.numel() - 1, min(n, flat.numel())).astype(np.int64)
    return flat[idx].numpy(), idx


def transformer_fixture(name, B, seed, overrides, cls='PointCAE_transformer'):
    from easydict import EasyDict
    import yaml
    import models.PointCAE_transformer as M
    from oracle import model as OM
    from point_dae_amd import corrupt_util_tensor as C
    from point_dae_amd.point_cae_transformer import draw_mask
    from point_dae_amd.synthetic import shapenet_like_clouds

    cfg_path = 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'
    cfg = EasyDict(yaml.safe_load(open(os.path.join(R.REF, cfg_path)))['model'])
    for k, v in overrides.items():
        node = cfg
        parts = k.split('.')
        for p in parts[:-1]:
            node = node[p]
        node[parts[-1]] = v
    R.seed_all(seed)
    ref = fill_state(getattr(M, cls)(cfg), seed)
    ref.train()
    pts = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=seed))

    cap = {}
    ref.group_divider.register_forward_hook(lambda m, i, o: cap.update(center=o[1]))
    ref.MAE_encoder.encoder.register_forward_hook(lambda m, i, o: cap.update(tokens=o, t_nb=i[0]))
    ref.MAE_encoder.register_forward_hook(lambda m, i, o: cap.update(x_vis=o[0], mask=o[1]))
    ref.MAE_decoder.register_forward_hook(lambda m, i, o: cap.update(x_rec=o))

    # the random draws the reference is about to make (same RNG calls, same order)
    R.seed_all(seed + 1)
    steps = C.draw_corruption(cfg.corrupt_type, B)
    mask, _ = draw_mask(B, cfg.num_group, cfg.transformer_config.mask_ratio, cfg.transformer_config.rand_ratio)
    masked = 'Drop-Patch' in cfg.corrupt_type
    if not masked:                                   # NormalTransformer: no mask is drawn, its output is the tokens
        R.seed_all(seed + 1)
        steps = C.draw_corruption(cfg.corrupt_type, B)
        mask = torch.zeros(B, cfg.num_group, dtype=torch.bool)
        ref.MAE_encoder.register_forward_hook(lambda m, i, o: cap.update(x_vis=o, mask=mask))
    R.seed_all(seed + 1)
    loss, loss2 = ref(pts, pts)
    (loss + 0.005 * loss2.sum()).backward()          # runner :165-166, normal_weight 0.005
    assert torch.equal(cap['mask'], mask), 'mask draw does not reproduce the reference'

    # cross-check: the oracle model with the captured draws injected reproduces the reference bit for bit
    orc = getattr(OM, cls)(cfg)
    orc.load_state_dict(ref.state_dict())
    osteps = [('mul', s[:, 1:4]) if s[0, 0] == 0 else ('mat', s[:, 1:].reshape(B, 3, 3)) for s in steps]
    oloss, oloss2 = orc(pts, pts, mask=mask, steps=osteps)
    assert oloss.item() == loss.item() and oloss2.sum().item() == loss2.sum().item(), (oloss.item(), loss.item())

    out = dict(seed=np.int64(seed), B=np.int64(B), pts=pts.numpy(), steps=steps.numpy(), mask=mask.numpy(),
               loss=np.float32(loss.item()), loss2=loss2.detach().numpy().reshape(-1), cls=np.array(cls),
               center=cap['center'].numpy(), tokens=cap['tokens'].detach().numpy(),
               t_nb=cap['t_nb'].detach().numpy()[:, ::8], x_vis=cap['x_vis'].detach().numpy(),
               x_rec=cap['x_rec'].detach().numpy(),
               overrides=np.array(repr(sorted(overrides.items()))))
    for pname, p in ref.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)      # (an unused parameter: mask_token without masking)
        key = 'grad/' + pname
        out[key + '/norm'] = np.float64(g.double().norm().item())
        if g.numel() <= 1536:
            out[key + '/full'] = g.numpy()
        else:
            out[key + '/sample'], _ = _sample(g)
    for bname, b in ref.named_buffers():         # BatchNorm running statistics after one step
        if b.dtype.is_floating_point:
            out['buf/' + bname] = b.numpy()
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(name, 'loss', loss.item(), 'steps', steps.shape, 'masked', int(mask[0].sum()),
          'size %.0f KB' % (os.path.getsize(path) / 1024))


def pointnetv2_fixture(name, B, seed, corrupt_type=None):
    """BASELINE config 1: pretrain_PointCAE_clean.yaml, Point_CAE_PointNetv2, B=2, N=1024.  corrupt_type:
    an in-forward corruption list (pretrain_PointCAE_dropout_global.yaml: ['dropout_global']); the host RNG
    is re-seeded with seed + 7 right before each forward so that the drawn subset can be reproduced."""
    from easydict import EasyDict
    import yaml
    import pointnet2_utils as vendored
    vendored.GroupAll.ret_grouped_xyz = False     # attribute read at pointnet2_utils.py:421, never set (:386-389)
    import models.PointCAE_pointnetv2 as M
    from oracle import model as OM
    from point_dae_amd.synthetic import shapenet_like_clouds
    cfg = EasyDict(yaml.safe_load(open(os.path.join(R.REF, 'cfgs/pretrain_PointCAE_clean.yaml')))['model'])
    if corrupt_type is not None:
        cfg.corrupt_type = list(corrupt_type)
    R.seed_all(seed)
    ref = M.Point_CAE_PointNetv2(cfg)
    ref.device = torch.device('cpu')
    fill_state(ref, seed)
    ref.train()
    clean = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=seed))
    corrupted = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=seed + 100))
    cap = {}
    ref.pointnetv2_encoder.register_forward_hook(lambda m, i, o: cap.update(feature=o))
    ref.folding1.register_forward_hook(lambda m, i, o: cap.update(coarse=o))
    # the winners of the three set-abstraction max-pools (F.max_pool2d over nsample, pointnet2_modules.py:60-66): the
    # gradient is routed through them, and a near-tie resolved the other way by 1 ulp of GEMM rounding moves whole
    # gradient tensors by ~1e-2 -- the parity test injects these so that it compares like with like
    enc = ref.pointnetv2_encoder
    for lvl, sa in enumerate((enc.sa1, enc.sa2, enc.sa3)):
        sa.mlps[0].register_forward_hook(
            lambda m, i, o, lvl=lvl: cap.update({'arg%d' % lvl: o.detach().argmax(dim=3)}))      # (B, C, npoint)
    import random
    random.seed(seed + 7), torch.manual_seed(seed + 7)
    l_coarse, l_fine = ref(corrupted, clean)
    (l_coarse + 0.5 * l_fine).backward()
    orc = fill_state(OM.Point_CAE_PointNetv2(cfg), seed).train()
    random.seed(seed + 7), torch.manual_seed(seed + 7)
    o1, o2 = orc(corrupted, clean)
    assert o1.item() == l_coarse.item() and o2.item() == l_fine.item(), (o1.item(), l_coarse.item())
    out = dict(seed=np.int64(seed), B=np.int64(B), clean=clean.numpy(), corrupted=corrupted.numpy(),
               loss_coarse=np.float32(l_coarse.item()), loss_fine=np.float32(l_fine.item()),
               feature=cap['feature'].detach().numpy(), coarse=cap['coarse'].detach().numpy().reshape(B, 1024, 3))
    for lvl in range(3):                                    # product layout: (groups = B * npoint, C) uint8
        a = cap['arg%d' % lvl]
        out['sa_argmax%d' % lvl] = a.permute(0, 2, 1).reshape(-1, a.shape[1]).numpy().astype(np.uint8)
    for pname, p in ref.named_parameters():
        g = p.grad
        key = 'grad/' + pname
        out[key + '/norm'] = np.float64(g.double().norm().item())
        if g.numel() <= 1536:
            out[key + '/full'] = g.numpy()
        else:
            out[key + '/sample'], _ = _sample(g)
    for bname, b in ref.named_buffers():
        if b.dtype.is_floating_point:
            out['buf/' + bname] = b.numpy()
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(name, 'loss', l_coarse.item(), l_fine.item(), 'size %.0f KB' % (os.path.getsize(path) / 1024))


def dgcnn_fixture(name, B, seed, corrupt_type=None):
    """Point_CAE_DGCNN_FCOnly (the published non-Transformer model), B=2, N=1024.  corrupt_type: an in-forward
    corruption list (models/PointCAE_DGCNN.py:198-221); python's and torch's host generators are re-seeded with
    seed + 7 right before each forward so that the draws (patch-drop level and mask, global-drop order) reproduce."""
    from easydict import EasyDict
    import yaml
    import models.PointCAE_DGCNN as M
    from oracle import model as OM
    from point_dae_amd.synthetic import shapenet_like_clouds
    cfg = EasyDict(yaml.safe_load(open(os.path.join(R.REF, 'cfgs/pretrain_PointCAE_clean.yaml')))['model'])
    cfg.NAME = 'Point_CAE_DGCNN_FCOnly'
    if corrupt_type is not None:
        cfg.corrupt_type = list(corrupt_type)
    R.seed_all(seed)
    ref = M.Point_CAE_DGCNN_FCOnly(cfg)
    ref.device = torch.device('cpu')
    fill_state(ref, seed)
    ref.train()
    clean = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=seed))
    corrupted = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=seed + 100))
    cap = {}
    ref.dgcnn_encoder.register_forward_hook(lambda m, i, o: cap.update(feature=o))
    ref.recfc.register_forward_hook(lambda m, i, o: cap.update(coarse=o))
    import random
    random.seed(seed + 7), torch.manual_seed(seed + 7)
    loss, loss2 = ref(corrupted, clean)
    (loss + 0.5 * loss2.sum()).backward()
    orc = fill_state(OM.Point_CAE_DGCNN_FCOnly(cfg), seed).train()
    random.seed(seed + 7), torch.manual_seed(seed + 7)
    o1, o2 = orc(corrupted, clean)
    assert o1.item() == loss.item(), (o1.item(), loss.item())
    out = dict(seed=np.int64(seed), B=np.int64(B), clean=clean.numpy(), corrupted=corrupted.numpy(),
               loss=np.float32(loss.item()), feature=cap['feature'].detach().numpy(),
               coarse=cap['coarse'].detach().numpy().reshape(B, 1024, 3))
    for pname, p in ref.named_parameters():
        g = p.grad
        key = 'grad/' + pname
        out[key + '/norm'] = np.float64(g.double().norm().item())
        if g.numel() <= 1536:
            out[key + '/full'] = g.numpy()
        else:
            out[key + '/sample'], _ = _sample(g)
    for bname, b in ref.named_buffers():
        if b.dtype.is_floating_point:
            out['buf/' + bname] = b.numpy()
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(name, 'loss', loss.item(), 'size %.0f KB' % (os.path.getsize(path) / 1024))


if __name__ == '__main__':
    R.setup()
    R.cpu_cuda_noop()
    if len(sys.argv) > 1 and sys.argv[1] == 'dgcnn':
        dgcnn_fixture('dgcnn_fconly_b2.npz', 2, 31)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'dgcnn_dropout':
        dgcnn_fixture('dgcnn_dropout_patch_b2.npz', 2, 33, ['dropout_patch_pointmae'])
        dgcnn_fixture('dgcnn_dropout_global_p3_b2.npz', 2, 35, ['dropout_global_p3'])
        dgcnn_fixture('dgcnn_random_dropout_b2.npz', 2, 37, ['random_dropout'])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'pointnetv2_dropout_patch':
        pointnetv2_fixture('pointnetv2_dropout_patch_b2.npz', 2, 25, ['dropout_patch_pointmae'])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'dropout_global':
        pointnetv2_fixture('pointnetv2_dropout_global_b2.npz', 2, 23, ['dropout_global'])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'pointnetv2':
        pointnetv2_fixture('pointnetv2_cfg1_b2.npz', 2, 21)
        pointnetv2_fixture('pointnetv2_dropout_global_b2.npz', 2, 23, ['dropout_global'])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'nomask':
        # the branch without patch masking (corrupt_type without 'Drop-Patch': NormalTransformer)
        transformer_fixture('transformer_nomask_b2.npz', 2, 14, {'transformer_config.drop_path_rate': 0.0,
                            'corrupt_type': ['affine_r3'], 'transformer_config.depth': 2,
                            'transformer_config.decoder_depth': 1})
        sys.exit(0)
    pointnetv2_fixture('pointnetv2_cfg1_b2.npz', 2, 21)
    pointnetv2_fixture('pointnetv2_dropout_global_b2.npz', 2, 23, ['dropout_global'])
    pointnetv2_fixture('pointnetv2_dropout_patch_b2.npz', 2, 25, ['dropout_patch_pointmae'])
    dgcnn_fixture('dgcnn_fconly_b2.npz', 2, 31)
    dgcnn_fixture('dgcnn_dropout_patch_b2.npz', 2, 33, ['dropout_patch_pointmae'])
    dgcnn_fixture('dgcnn_dropout_global_p3_b2.npz', 2, 35, ['dropout_global_p3'])
    dgcnn_fixture('dgcnn_random_dropout_b2.npz', 2, 37, ['random_dropout'])
    transformer_fixture('transformer_folding_b2.npz', 2, 13, {'transformer_config.drop_path_rate': 0.0,
                        'transformer_config.depth': 4, 'transformer_config.decoder_depth': 2},
                        cls='PointCAE_transformer_fc_global_folding_local')
    # cfg3 architecture at full size (384-d, 12+4 blocks), B=2, stochastic depth off
    transformer_fixture('transformer_cfg3_b2.npz', 2, 11, {'transformer_config.drop_path_rate': 0.0})
    # all_patch variant, cdl1 loss, smaller stack
    transformer_fixture('transformer_allpatch_cdl1_b3.npz', 3, 12,
                        {'transformer_config.drop_path_rate': 0.0, 'all_patch': 'True', 'loss': 'cdl1',
                         'transformer_config.depth': 2, 'transformer_config.decoder_depth': 1})
