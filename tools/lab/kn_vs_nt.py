"""LAB: a data gradient dX = dY . W with W as stored ([K, N]: the patch-transposing staging of gemm3_kernel) against the
same product on a transposed copy W^T ([N, K]: the k-contiguous staging), both on the planned tile shapes, back to back.
    gpurun -- python tools/lab/kn_vs_nt.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import nn_ops  # noqa: E402


def timed(fn, reps=60):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    torch.manual_seed(0)
    for M in (2944, 8192):
        for (N, K, split) in ((384, 1152, True), (384, 1536, True), (1536, 384, False), (384, 384, False)):
            dY = torch.randn(M, K, device='cuda')
            W = torch.randn(K, N, device='cuda') * K ** -0.5
            Wt = W.t().contiguous()
            kn = lambda: nn_ops.rows_gemm(dY, W, True, may_split=split)
            nt = lambda: nn_ops.rows_gemm(dY, Wt, False, may_split=split)
            a, b = kn(), nt()
            same = torch.equal(a, b) if a.shape == b.shape else 'plans differ'
            print(f"M={M} N={N} K={K} slabs={a.shape[0] if a.dim() == 3 else 1}/{b.shape[0] if b.dim() == 3 else 1}: "
                  f"[K,N] {timed(kn):6.1f} us   [N,K] copy {timed(nt):6.1f} us   equal={same}", flush=True)


if __name__ == '__main__':
    main()
