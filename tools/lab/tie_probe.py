import os, sys, random
import numpy as np, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
from oracle import model as OM
from point_dae_amd.point_cae_transformer import PointCAE_transformer
from point_dae_amd.synthetic import shapenet_like_clouds
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd import _lib as L
from golden_util import fill_state
cfg = cfg_from_yaml_file('cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml').model
cfg.group_size, cfg.num_group = 64, 32
cfg.transformer_config.depth, cfg.transformer_config.decoder_depth = 2, 1
cfg.transformer_config.drop_path_rate = 0.0
def seed(s):
    random.seed(s), np.random.seed(s), torch.manual_seed(s)
for cs in (9, 10, 11, 12, 13, 14):
    x = shapenet_like_clouds(3, 1024, seed=cs)
    ref = fill_state(OM.PointCAE_transformer(cfg), 5).train()
    seed(3); l_ref, _ = ref(torch.from_numpy(x), torch.from_numpy(x)); l_ref.backward()
    gmax = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)
    row = 'clouds %d:' % cs
    for arith in (0, 1):
        L.set_gemm_arith(arith)
        mine = fill_state(PointCAE_transformer(cfg), 5).cuda().train()
        seed(3); l_my, _ = mine(torch.from_numpy(x).cuda(), torch.from_numpy(x).cuda()); l_my.backward()
        worst = (0, '')
        for (n, p), (_, q) in zip(ref.named_parameters(), mine.named_parameters()):
            if p.grad is None: continue
            scale = max(p.grad.abs().max().item(), 1e-3 * gmax)
            e = (q.grad.cpu() - p.grad).abs().max().item() / scale
            if e > worst[0]: worst = (e, n)
        row += '  %s loss rel %.1e worst grad %.1e (%s)' % ('f32' if arith == 0 else 'bf16x3', abs(l_my.item() - l_ref.item()) / abs(l_ref.item()), worst[0], worst[1][-30:])
    print(row, flush=True)
