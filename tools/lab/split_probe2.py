"""LAB: split vs single-phase graphed step with TWO gloo ranks sharing the GPU: which gradients differ, at which step?
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/lab/split_probe2.py"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib, builder  # noqa: E402
from point_dae_amd.config import cfg_from_yaml_file  # noqa: E402
from point_dae_amd.data_parallel import FlatDataParallel  # noqa: E402
from point_dae_amd.graph_step import GraphedTrainStep, use_created_stream  # noqa: E402
from point_dae_amd.misc import set_random_seed  # noqa: E402
from point_dae_amd.synthetic import shapenet_like_clouds  # noqa: E402

os.environ.setdefault('GLOO_SOCKET_IFNAME', 'lo')
dist.init_process_group('gloo')
rank = dist.get_rank()
torch.cuda.set_device(0)
use_created_stream()
_lib.set_deterministic(True)
config = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
config.model.transformer_config.depth = 2
config.model.transformer_config.decoder_depth = 1
B = 8
x = torch.from_numpy(shapenet_like_clouds(B * 2, 1024, seed=10 + rank)).cuda().split(B)
hook_log = []
grads = {}
for split in (True, False, True):
    set_random_seed(7 + rank)
    net = builder.model_builder(config.model).cuda().train()
    model = FlatDataParallel(net)
    opt, _ = builder.build_opti_sche(model, config)
    step = GraphedTrainStep(model, opt, config, B, 1024, warmup_eager=1, split=split)
    rec = []
    for i in range(6):
        out = step(x[i % 2])
        torch.cuda.synchronize()
        rec.append((out[0].item(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, sorted(step.graphs)))
    grads.setdefault(split, []).append(rec)
    dist.barrier()
for tag, a, b in (('split vs single', grads[True][0], grads[False][0]), ('split vs split again', grads[True][0], grads[True][1])):
    for i in range(6):
        diff = [n for n in a[i][1] if not torch.equal(a[i][1][n], b[i][1][n])]
        print('rank', rank, tag, 'step', i, 'loss', a[i][0], b[i][0], 'graphs', a[i][2], 'differing:', len(diff), diff[:6], flush=True)
dist.destroy_process_group()
