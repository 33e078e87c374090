"""LAB: would two half-batch chains on two streams hide the ~3.2 us gap between dependent kernels (and the prologue /
epilogue of the step's sub-wave GEMMs)?  12 encoder blocks forward + backward on M rows as ONE chain, against the same
work as TWO chains of M / 2 rows on two streams forked and joined inside one captured graph (separate weights per chain:
only the concurrency is measured).
    gpurun -- python tools/lab/two_chains.py"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import nn_ops  # noqa: E402
from point_dae_amd.graph_step import use_created_stream  # noqa: E402
from point_dae_amd.point_cae_transformer import Block  # noqa: E402


def chain(blocks, x, pos, B, T):
    for blk in blocks:
        x = blk(x, pos, B, T, (None, None))
    return x


def timed(g, reps=20):
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    use_created_stream(torch.device('cuda'))
    torch.manual_seed(0)
    C, depth = 384, 12
    for (B, T) in ((128, 23), (128, 64)):
        blocks = torch.nn.ModuleList([Block(C, 6, 0.0) for _ in range(depth)]).cuda()
        twin = copy.deepcopy(blocks)
        M = B * T
        x = torch.randn(M, C, device='cuda')
        pos = torch.randn(M, C, device='cuda')
        gy = torch.randn(M, C, device='cuda')
        h = B // 2 * T

        def one():
            xi = x.clone().requires_grad_(True)
            chain(blocks, xi, pos, B, T).backward(gy)

        def two(s1, s2):
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                xa = x[:h].clone().requires_grad_(True)
                chain(blocks, xa, pos[:h], B // 2, T).backward(gy[:h])
            with torch.cuda.stream(s2):
                xb = x[h:].clone().requires_grad_(True)
                chain(twin, xb, pos[h:], B // 2, T).backward(gy[h:])
            cur.wait_stream(s1)
            cur.wait_stream(s2)

        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for _ in range(2):
            one()
            two(s1, s2)
        torch.cuda.synchronize()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, capture_error_mode='thread_local'):
            one()
        with torch.cuda.graph(g2, capture_error_mode='thread_local'):
            two(s1, s2)
        print(f"B={B} T={T} ({M} rows, {depth} blocks fwd + bwd): one chain {timed(g1):6.3f} ms   two half-batch chains on two "
              f"streams {timed(g2):6.3f} ms", flush=True)


if __name__ == '__main__':
    main()
