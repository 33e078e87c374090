"""LAB: two processes on one GPU, each repeating the patch embedder's forward + backward on fixed inputs; which of its
gradients ever differ from the first iteration's?   (python tools/lab/embed_nondet.py spawns itself twice)"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(tag):
    from point_dae_amd import _lib, nn_ops
    from point_dae_amd.graph_step import use_created_stream
    import torch.nn as nn
    use_created_stream()
    _lib.set_deterministic(True)
    torch.manual_seed(0)
    first = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True), nn.Conv1d(128, 256, 1)).cuda()
    second = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, 384, 1)).cuda()
    BG = 8 * 64
    pts = torch.randn(BG, 32, 3, device='cuda')
    vis = torch.arange(0, BG, 3, dtype=torch.int32, device='cuda')
    msk = torch.tensor([i for i in range(BG) if i % 3], dtype=torch.int32, device='cuda')
    w = torch.randn(vis.numel(), 384, device='cuda')
    ref = None
    bad = {}
    for it in range(int(os.environ.get('ITERS', '60'))):
        for p in list(first.parameters()) + list(second.parameters()):
            p.grad = None
        tok = nn_ops.patch_embed(pts, first, second, True, vis, msk)
        (tok * w).sum().backward()
        torch.cuda.synchronize()
        cur = {n: p.grad.clone() for n, p in list(first.named_parameters(prefix='first')) + list(second.named_parameters(prefix='second'))}
        cur['tok'] = tok.detach().clone()
        if ref is None:
            ref = cur
        else:
            for n in cur:
                if not torch.equal(cur[n], ref[n]):
                    bad[n] = bad.get(n, 0) + 1
    print(tag, 'arith', _lib.gemm_arith(), 'differing over the iterations:', bad, flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), 'proc%d' % i]) for i in range(2)]
        for p in ps:
            p.wait()
