"""LAB: the grouped weight gradients in both arithmetics (fp32-input MFMA vs exact-split bf16) on the step's groups:
the encoder stack (12 blocks x 4 layers, M = 3584), the decoder stack (4 x 4, M = 8192), the embedder's large layers.
Time per launch (grouped kernel + reductions, back to back) and the error of one layer against fp64.
    gpurun -- python tools/lab/wgrad3_lab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def jobs_of(M, layers, blocks):
    out = []
    for _ in range(blocks):
        for n, k in layers:
            out.append((torch.randn(M, n, device='cuda'), torch.randn(M, k, device='cuda'), torch.empty(n, k, device='cuda'),
                        torch.empty(n, device='cuda')))
    return out


def main():
    torch.manual_seed(0)
    block = [(1152, 384), (384, 384), (1536, 384), (384, 1536)]
    groups = {'encoder stack 12 x 4, M=3584': jobs_of(3584, block, 12), 'decoder stack 4 x 4, M=8192': jobs_of(8192, block, 4),
              'one block, M=3584': jobs_of(3584, block, 1), '3 blocks, M=3584': jobs_of(3584, block, 3),
              'embedder 262144 x 512 x 256': jobs_of(262144, [(512, 256)], 1), 'embedder 262144 x 256 x 128': jobs_of(262144, [(256, 128)], 1),
              'embedder 114688 x 384 x 512': jobs_of(114688, [(384, 512)], 1)}
    for name, jobs in groups.items():
        flops = 2.0 * sum(j[0].shape[0] * j[0].shape[1] * j[1].shape[1] for j in jobs)
        row = f"{name:>34} {flops / 1e9:7.1f} GFLOP |"
        for arith, tag in ((0, 'fp32'), (1, 'bf16x3')):
            _lib.set_gemm_arith(arith)
            t = timed(lambda: _lib.rows_wgrad_multi(jobs))
            dy, x, dw, db = jobs[-1]
            dw.fill_(float('nan'))
            _lib.rows_wgrad_multi(jobs)
            ref = dy.double().t() @ x.double()
            err = (dw.double() - ref).abs().max().item() / ref.abs().max().item()
            eb = (db.double() - dy.double().sum(0)).abs().max().item() / dy.double().sum(0).abs().max().item()
            row += f" {tag} {t:8.1f} us {flops / t / 1e6:6.1f} TF/s err {err:.1e} db {eb:.1e} |"
        print(row, flush=True)


if __name__ == '__main__':
    main()
