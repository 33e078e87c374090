"""LAB: two processes on one GPU, each repeating the SAME eager forward + backward of the depth-2+1 model (fixed inputs,
fixed host draws, deterministic mode): which gradients ever differ from the first iteration's, by how much?"""
import os
import random
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(tag):
    from point_dae_amd import _lib, builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.graph_step import use_created_stream
    from point_dae_amd.synthetic import shapenet_like_clouds
    use_created_stream()
    _lib.set_deterministic(True)
    config = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    config.model.transformer_config.depth = 2
    config.model.transformer_config.decoder_depth = 1
    torch.manual_seed(1)
    net = builder.model_builder(config.model).cuda().train()
    x = torch.from_numpy(shapenet_like_clouds(8, 1024, seed=10)).cuda()
    from point_dae_amd import patch_embed as PE
    PE.DEBUG_KEEP = {}
    ref, bad, refk = None, {}, None
    for it in range(int(os.environ.get('ITERS', '40'))):
        random.seed(5), np.random.seed(5), torch.manual_seed(5)
        for p in net.parameters():
            p.grad = None
        lx, ln = net(x, x)
        (lx + 0.005 * ln.sum()).backward()
        torch.cuda.synchronize()
        cur = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        cur['loss'] = lx.detach().clone()
        keep = dict(PE.DEBUG_KEEP)
        if ref is None:
            ref, refk = cur, keep
        else:
            for n in cur:
                if not torch.equal(cur[n], ref[n]):
                    d = (cur[n] - ref[n]).abs()
                    bad.setdefault(n, []).append((int((d > 0).sum()), float(d.max() / (ref[n].abs().max() + 1e-30))))
                    if len(bad[n]) <= 2:
                        for k in keep:
                            if not torch.equal(keep[k], refk[k]):
                                dd = (keep[k] != refk[k]).nonzero()
                                print(tag, 'iteration', it, 'intermediate', k, tuple(keep[k].shape), 'differs in', dd.shape[0], 'elements; first', dd[0].tolist(), 'last', dd[-1].tolist(),
                                      'values', keep[k][tuple(dd[0].tolist())].item(), refk[k][tuple(dd[0].tolist())].item(), flush=True)
    print(tag, 'arith', _lib.gemm_arith(), {k: (len(v), v[0]) for k, v in bad.items()}, flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        n = int(os.environ.get('PROCS', '2'))
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'proc%d' % i]) for i in range(n)]
        for p in ps:
            p.wait()
