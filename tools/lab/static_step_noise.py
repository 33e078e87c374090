"""Eager vs graphed cfg2 steps (tests/test_gpu_model.py::test_graphed_static_step_equals_eager_step): spread of the
per-step loss differences over repetitions, in atomic and deterministic mode."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from point_dae_amd import builder, _lib
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.data_parallel import FlatDataParallel
from point_dae_amd.graph_step import GraphedStaticStep, use_created_stream
from point_dae_amd.synthetic import shapenet_like_clouds
use_created_stream()
config = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs', 'pretrain_PointCAE_affine_r3_dropout_local_4xlonger.yaml'))
config.optimizer.kwargs.lr = 1e-4
B = 8
clean = torch.from_numpy(shapenet_like_clouds(B * 2, 1024, seed=5)).cuda().split(B)
corrupted = torch.from_numpy(shapenet_like_clouds(B * 2, 1024, seed=6)).cuda().split(B)
weights = [0.0, 0.25, 0.5, 0.75, 1.0, 1.0]
for det in (False, True):
    _lib.set_deterministic(det, 256) if det else None
    worst = [0.0] * len(weights)
    wp = 0.0
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        torch.manual_seed(0)
        net_a = builder.model_builder(config.model).cuda().train()
        net_b = copy.deepcopy(net_a)
        model_a = FlatDataParallel(net_a)
        opt_a, _ = builder.build_opti_sche(model_a, config)
        model_a.zero_grad()
        eager = []
        for i, w in enumerate(weights):
            lc, lf = model_a(corrupted[i % 2], clean[i % 2])
            (lc + w * lf).backward()
            opt_a.step(); model_a.zero_grad()
            eager.append((lc.item(), lf.item()))
        model_b = FlatDataParallel(net_b)
        opt_b, _ = builder.build_opti_sche(model_b, config)
        gw = torch.zeros((), device='cuda')
        step = GraphedStaticStep(model_b, opt_b, lambda a, b: a + b * gw, B, 1024, warmup_eager=1)
        graphed = []
        for i, w in enumerate(weights):
            gw.fill_(w)
            lc, lf = step(corrupted[i % 2], clean[i % 2])
            graphed.append((lc.item(), lf.item()))
        for i, ((a0, a1), (b0, b1)) in enumerate(zip(eager, graphed)):
            worst[i] = max(worst[i], abs(a0 - b0) / abs(a0), abs(a1 - b1) / abs(a1))
        wp = max(wp, (model_a.flat_param - model_b.flat_param).abs().max().item())
    print('deterministic' if det else 'atomic', ' '.join('%.1e' % w for w in worst), 'param %.1e' % wp)
