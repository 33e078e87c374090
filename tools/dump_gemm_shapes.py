"""Every distinct product the row-GEMM family computes in one optimisation step of cfg3 (PointCAE_transformer, B = 128,
all visible-token counts the mask ratio can draw), the published variant and cfg2 (Point_CAE_PointNetv2, B = 128):
recorded at the C boundary (point_dae_amd/_lib.CALL_HOOK) while the steps run eagerly -> tests/golden/gemm_shapes.json,
the shape list of tests/test_gpu_rows3.py.    gpurun -- python tools/dump_gemm_shapes.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from point_dae_amd import _lib  # noqa: E402

gemm, wgrad = set(), set()


def hook(name, a):
    if name == 'pdae_rows_gemm':
        gemm.add((a[0], a[1], a[2], int(a[5]), int(a[7])))          # M, N, K, w_kn, epi
    elif name == 'pdae_rows_wgrad_listed':
        wgrad.add((a[0], a[1], a[2], int(a[4] is not None), int(a[6] is not None), int(a[7] is not None)))
    elif name == 'pdae_rows_wgrad_multi':
        for m, n, k in zip(list(a[1]), list(a[6]), list(a[7])):
            wgrad.add((m, n, k, 0, 0, 0))
    elif name == 'pdae_rows_wgrad':
        for n, k in zip(list(a[6]), list(a[7])):
            wgrad.add((a[0], n, k, 0, 0, 0))


def main():
    from point_dae_amd import builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedStaticStep, GraphedTrainStep, use_created_stream
    from point_dae_amd.synthetic import shapenet_like_clouds
    device = torch.device('cuda', 0)
    use_created_stream(device)
    out = {}
    B, N = 128, 1024
    x = torch.from_numpy(shapenet_like_clouds(2 * B, N, seed=7)).to(device)
    for wl in ('cfg3', 'published', 'cfg2'):
        gemm.clear(), wgrad.clear()
        config = cfg_from_yaml_file(os.path.join(ROOT, bench.CFG2 if wl == 'cfg2' else bench.CFG3))
        if wl == 'published':
            config.model.NAME = 'PointCAE_transformer_fc_global_folding_local'
        model = FlatDataParallel(builder.model_builder(config.model).to(device), broadcast=False, process_group=None)
        model.world_size = 1
        optimizer, _ = builder.build_opti_sche(model, config)
        model.train()
        model.zero_grad()
        _lib.CALL_HOOK = hook
        if wl == 'cfg2':
            step = GraphedStaticStep(model, optimizer, lambda a, b: a + 0.5 * b, B, N)
            step(x[:B], x[B:])                    # the first calls of a graphed step run eagerly
        else:
            step = GraphedTrainStep(model, optimizer, config, B, N, split=False)
            seen = set()
            for i in range(400):
                step.pts.copy_(x[:B])
                tvis = step._draw()
                if tvis in seen:
                    continue
                seen.add(tvis)
                step._fwd_bwd(tvis)
                model.zero_grad()
                if len(seen) == 20:
                    break
        _lib.CALL_HOOK = None
        torch.cuda.synchronize()
        out[wl] = {'gemm': sorted(gemm), 'wgrad': sorted(wgrad)}
        print(wl, len(gemm), 'gemm shapes,', len(wgrad), 'wgrad shapes', flush=True)
        del model, optimizer, step
        torch.cuda.empty_cache()
    with open(os.path.join(ROOT, 'gpurun_out', 'gemm_shapes.json'), 'w') as f:
        json.dump(out, f)


if __name__ == '__main__':
    main()
