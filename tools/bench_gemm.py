"""GEMM micro-benchmark on the Transformer-block shapes of the step: hand-written fp32 MFMA kernels
(C ABI) vs torch.mm (hipBLASLt).  M = 128 * T_vis rows in the encoder, 8192 in the decoder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import _lib
from point_dae_amd.graph_step import use_created_stream

use_created_stream()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3      # us


Ms = [int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else [1664, 2944, 4096, 8192]
layers = [('qkv', 1152, 384), ('proj', 384, 384), ('fc1', 1536, 384), ('fc2', 384, 1536)]
tot = {}
for M in Ms:
    for name, N, K in layers:
        x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05; b = torch.randn(N, device='cuda')
        y = torch.empty(M, N, device='cuda')
        dy = torch.randn(M, N, device='cuda'); wt = w.t().contiguous(); dx = torch.empty(M, K, device='cuda')
        dw = torch.empty(N, K, device='cuda'); db = torch.empty(N, device='cuda')
        fl = 2.0 * M * N * K / 1e6     # MFLOP -> /us = TFLOP/s
        f1 = lambda: _lib.call('pdae_linear_forward', x, M, N, K, x.data_ptr(), w.data_ptr(), b.data_ptr(), 0, y.data_ptr())
        f2 = lambda: torch.mm(x, w.t())
        g1 = lambda: _lib.call('pdae_linear_backward_data', x, M, N, K, dy.data_ptr(), wt.data_ptr(), dx.data_ptr())
        g2 = lambda: torch.mm(dy, w)
        h1 = lambda: _lib.call('pdae_linear_backward_weight', x, M, N, K, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr())
        h2 = lambda: torch.mm(dy.t(), x)
        t = [timeit(f) for f in (f1, f2, g1, g2, h1, h2)]
        for k, v in zip(('fwd_mine', 'fwd_lib', 'dx_mine', 'dx_lib', 'dw_mine', 'dw_lib'), t):
            tot[(M, k)] = tot.get((M, k), 0.) + v
        print(f"M{M:5d} {name:4s} N{N:4d} K{K:4d}: fwd {t[0]:6.1f}us {fl/t[0]:6.1f}TF | lib {t[1]:6.1f}us {fl/t[1]:6.1f}TF || "
              f"dX {t[2]:6.1f}us {fl/t[2]:6.1f} | lib {t[3]:6.1f}us {fl/t[3]:6.1f} || dW {t[4]:6.1f}us {fl/t[4]:6.1f} | lib {t[5]:6.1f}us {fl/t[5]:6.1f}",
              flush=True)
    print(f"M{M:5d} block totals (us): " + "  ".join(f"{k} {tot[(M, k)]:.1f}" for k in ('fwd_mine', 'fwd_lib', 'dx_mine', 'dx_lib', 'dw_mine', 'dw_lib')), flush=True)
