"""GEMM micro-benchmark: hand-written fp32 MFMA kernels vs torch (hipBLASLt) on the step's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import _lib

def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

shapes = [(262144, 384, 512), (262144, 512, 256), (262144, 256, 128), (8192, 1536, 384), (8192, 384, 1536),
          (2944, 1152, 384), (2944, 384, 384), (8192, 1152, 384), (5248, 96, 384)]
for M, N, K in shapes:
    x = torch.randn(M, K, device='cuda'); w = torch.randn(N, K, device='cuda') * 0.05; b = torch.randn(N, device='cuda')
    y = torch.empty(M, N, device='cuda')
    f_mine = lambda: _lib.call('pdae_linear_forward', x, M, N, K, x.data_ptr(), w.data_ptr(), b.data_ptr(), 0, y.data_ptr())
    f_ref = lambda: torch.nn.functional.linear(x, w, b)
    f_mine(); ref = f_ref()
    err = ((y - ref).abs().max() / ref.abs().max()).item()
    t1, t2 = timeit(f_mine), timeit(f_ref)
    fl = 2.0 * M * N * K / 1e9
    # backward
    dy = torch.randn(M, N, device='cuda'); wt = w.t().contiguous(); dx = torch.empty(M, K, device='cuda')
    g_mine = lambda: _lib.call('pdae_linear_backward_data', x, M, N, K, dy.data_ptr(), wt.data_ptr(), dx.data_ptr())
    g_ref = lambda: dy @ w
    g_mine(); r2 = g_ref(); err2 = ((dx - r2).abs().max() / r2.abs().max()).item()
    t3, t4 = timeit(g_mine), timeit(g_ref)
    dw = torch.empty(N, K, device='cuda'); db = torch.empty(N, device='cuda')
    h_mine = lambda: _lib.call('pdae_linear_backward_weight', x, M, N, K, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr())
    h_ref = lambda: (dy.t() @ x, dy.sum(0))
    h_mine(); r3, r4 = h_ref(); err3 = ((dw - r3).abs().max() / r3.abs().max()).item(); err4 = ((db - r4).abs().max() / r4.abs().max()).item()
    t5, t6 = timeit(h_mine), timeit(h_ref)
    print(f"M{M} N{N} K{K}: fwd mine {fl/t1:7.1f} TF ({t1*1e3:7.1f}us) torch {fl/t2:7.1f} TF err {err:.1e} | dgrad {fl/t3:6.1f} vs {fl/t4:6.1f} err {err2:.1e} | wgrad {fl/t5:6.1f} vs {fl/t6:6.1f} err {err3:.1e} {err4:.1e}")
