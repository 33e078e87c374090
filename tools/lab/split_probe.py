"""LAB: which parameter gradients differ between the two-phase (split) and the single-phase graphed step?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib, builder  # noqa: E402
from point_dae_amd.config import cfg_from_yaml_file  # noqa: E402
from point_dae_amd.data_parallel import FlatDataParallel  # noqa: E402
from point_dae_amd.graph_step import GraphedTrainStep, use_created_stream  # noqa: E402
from point_dae_amd.misc import set_random_seed  # noqa: E402
from point_dae_amd.synthetic import shapenet_like_clouds  # noqa: E402

use_created_stream()
_lib.set_deterministic(True)
config = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
config.model.transformer_config.depth = 2
config.model.transformer_config.decoder_depth = 1
B = 8
x = torch.from_numpy(shapenet_like_clouds(B * 2, 1024, seed=10)).cuda().split(B)
grads = {}
for split in (True, False):
    set_random_seed(7)
    net = builder.model_builder(config.model).cuda().train()
    model = FlatDataParallel(net, broadcast=False, process_group=None)
    model.world_size = 1
    opt, _ = builder.build_opti_sche(model, config)
    step = GraphedTrainStep(model, opt, config, B, 1024, warmup_eager=1, split=split)
    rec = []
    for i in range(5):
        set_random_seed(100 + i)
        out = step(x[i % 2])
        torch.cuda.synchronize()
        rec.append((out[0].item(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}))
    grads[split] = rec
for i in range(5):
    a, b = grads[True][i], grads[False][i]
    diff = [n for n in a[1] if not torch.equal(a[1][n], b[1][n])]
    print('step', i, 'loss', a[0], b[0], 'differing gradient tensors:', len(diff), diff[:12])
