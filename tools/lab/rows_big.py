"""LAB: the tile shapes of pdae_rows_gemm on the LARGE products of cfg2's FoldingNet (the plan's cost model was
calibrated on M = 1664 .. 8192).  python tools/lab/rows_big.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from point_dae_amd import _lib  # noqa: E402

NAMES = ['128x128', '64x128', '128x64', '64x64', '64x192', '96x128', '128x96', '128x192']
for (M, N, K, w_kn, epi) in [(524288, 512, 512, False, 1), (524288, 512, 512, False, 4), (524288, 512, 512, True, 0),
                             (262144, 512, 256, True, 0), (65536, 256, 128, False, 1), (32768, 1024, 512, False, 1)]:
    x = torch.randn(M, K, device='cuda')
    w = torch.randn((K, N) if w_kn else (N, K), device='cuda') * 0.05
    b = torch.randn(N, device='cuda') if epi == 1 else None
    z = torch.randn(M, N, device='cuda') if epi == 4 else None
    y = torch.empty(M, N, device='cuda')
    planned = _lib.rows_gemm_plan(M, N, K, w_kn, False)[0]
    row = f"({M}, {N}, {K}) {'KN' if w_kn else 'NK'} epi {epi}: plan {NAMES[planned]} |"
    for cfg in range(8):
        f = lambda: _lib.call('pdae_rows_gemm', x, M, N, K, _lib.ptr(x), _lib.ptr(w), int(w_kn), _lib.ptr(b), epi,
                              _lib.ptr(z), _lib.ptr(y), cfg, 1, 0)
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 200
        row += f" {NAMES[cfg]} {2.0 * M * N * K / us / 1e6:5.1f}"
    print(row + "  TF/s", flush=True)
