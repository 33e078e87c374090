"""Micro-benchmark of the transformer-block kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import _lib, nn_ops

from point_dae_amd.graph_step import use_created_stream
use_created_stream()


def timeit(fn, iters=20, warm=2, reps=3):
    """us per call inside a replayed hipGraph of back-to-back calls"""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / iters * 1e3)
    return best

for M in (1664, 2944, 4096, 8192):
    C = 384
    x = torch.randn(M, C, device='cuda'); dy = torch.randn(M, C, device='cuda'); dres = torch.randn(M, C, device='cuda')
    g = torch.ones(C, device='cuda'); b = torch.zeros(C, device='cuda')
    y = torch.empty_like(x); mean = torch.empty(M, device='cuda'); rstd = torch.empty(M, device='cuda'); dx = torch.empty_like(x)
    gb = torch.zeros(2 * C, device='cuda')
    f = lambda: _lib.call('pdae_add_layernorm_forward', x, M, C, x.data_ptr(), None, g.data_ptr(), b.data_ptr(), 1e-5, None, y.data_ptr(), mean.data_ptr(), rstd.data_ptr())
    f()
    bw = lambda: _lib.call('pdae_layernorm_backward', x, M, C, dy.data_ptr(), 1, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dres.data_ptr(), dx.data_ptr(), gb.data_ptr(), gb[C:].data_ptr(), 1, None, 0)
    print(f"M={M}: ln fwd {timeit(f):.1f} us  ln bwd {timeit(bw):.1f} us  (bytes bwd {4*M*C*4/1e6:.1f} MB)")
    B, T, H = (128, M // 128, 6)
    qkv = torch.randn(M, 1152, device='cuda'); o = torch.empty(M, 384, device='cuda'); lse = torch.empty(B, H, T, device='cuda'); do = torch.randn(M, 384, device='cuda'); dqkv = torch.empty_like(qkv)
    af = lambda: _lib.call('pdae_attention_forward', qkv, B, T, H, 64, 0.125, qkv.data_ptr(), o.data_ptr(), lse.data_ptr())
    ab = lambda: _lib.call('pdae_attention_backward', qkv, B, T, H, 64, 0.125, qkv.data_ptr(), o.data_ptr(), lse.data_ptr(), do.data_ptr(), dqkv.data_ptr())
    af()
    print(f"   attention T={T}: fwd {timeit(af):.1f} us  bwd {timeit(ab):.1f} us")
