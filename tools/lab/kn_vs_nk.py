"""LAB: the data-gradient GEMMs of a Transformer block with the weight read as [K,N] (what ships: the (out,in) tensor in
place) against [N,K] (a transposed copy the optimiser would have to maintain).  50 launches each, HIP events."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from point_dae_amd import nn_ops  # noqa: E402


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (2944, 3584, 8192):
    for (N, K, epi, split) in [(384, 1152, 0, True), (384, 384, 0, False), (384, 1536, 0, True), (1536, 384, 0, False)]:   # (the last one ships with the GELU' epilogue, [K,N] only)
        dy = torch.randn(M, K, device='cuda')
        w = torch.randn(K, N, device='cuda') * 0.05                     # (out, in) = [K][N] of the data-gradient product
        wt = w.t().contiguous()
        z = torch.randn(M, N, device='cuda') if epi == 3 else None
        a = t(lambda: nn_ops.rows_gemm(dy, w, True, None, epi, z, may_split=split))
        b = t(lambda: nn_ops.rows_gemm(dy, wt, False, None, epi, z, may_split=split))
        print(f"M {M:5d}  N {N:5d}  K {K:5d}  epi {epi}:  [K,N] {a:6.1f} us   [N,K] {b:6.1f} us   ({(a / b - 1) * 100:+.1f} %)", flush=True)
