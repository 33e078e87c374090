"""Per-node cost of torch-captured hipGraphs (GPU box): chains of tiny ATen / pdae kernels, replayed."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib
from point_dae_amd.graph_step import use_created_stream
use_created_stream()
x = torch.zeros(768, device='cuda')
big = torch.zeros(16 << 20, device='cuda')
g_, b_ = torch.ones(384, device='cuda'), torch.zeros(384, device='cuda')
rows = torch.randn(3584, 384, device='cuda')
def chain_tiny(n):
    for _ in range(n): x.add_(1.0)
def chain_mixed(n):          # tiny kernel behind a 64 MB streaming write
    for _ in range(n // 2):
        big.add_(1.0); x.add_(1.0)
def chain_ln(n):
    y = torch.empty_like(rows); mean = torch.empty(3584, device='cuda'); rstd = torch.empty(3584, device='cuda')
    for _ in range(n):
        _lib.call('pdae_add_layernorm_forward', rows, 3584, 384, _lib.ptr(rows), None, _lib.ptr(g_), _lib.ptr(b_), 1e-5, None, _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd))
for name, fn, n in (('tiny aten add_', chain_tiny, 400), ('64MB add_ + tiny add_', chain_mixed, 100), ('pdae add_layernorm_fwd 3584x384', chain_ln, 200)):
    fn(8); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        fn(n)
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); g.replay(); e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    # eager for comparison
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(n); e.record(); torch.cuda.synchronize()
    print('%-34s graph replay %.2f us/node   eager %.2f us/launch' % (name, best, s.elapsed_time(e) * 1e3 / n))
