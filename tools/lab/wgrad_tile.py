"""LAB: one weight-gradient problem timed on the grouped kernel (PDAE_WGRAD_TN caps the tile width: run twice).
    PDAE_WGRAD_TN=128 python tools/lab/wgrad_tile.py; python tools/lab/wgrad_tile.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from point_dae_amd import _lib  # noqa: E402

for (M, N, K) in [(2097152, 512, 512), (262144, 512, 256), (262144, 384, 512), (131072, 256, 1024), (8192, 1024, 512)]:
    dy = torch.randn(M, N, device='cuda')
    x = torch.randn(M, K, device='cuda')
    dw, db = torch.empty(N, K, device='cuda'), torch.empty(N, device='cuda')
    ws = torch.empty(max(_lib.rows_wgrad_workspace(M, [N], [K]), 1), device='cuda')
    f = lambda: _lib.rows_wgrad(dy, M, [dy], [x], [dw], [db], ws)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"TN cap {os.environ.get('PDAE_WGRAD_TN', '384')}: ({M}, {N}, {K}) {us:9.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TF/s", flush=True)
    del dy, x
