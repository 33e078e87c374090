"""LAB aggressor: loop one row-GEMM launch shape for SECS seconds (env: ARITH f32|bf16x3, CFG, SHAPE MxNxK, KN)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import _lib  # noqa: E402
from point_dae_amd.graph_step import use_created_stream  # noqa: E402
use_created_stream()
_lib.set_gemm_arith(0 if os.environ.get('ARITH', 'bf16x3') == 'f32' else 1)
M, N, K = (int(v) for v in os.environ.get('SHAPE', '2944x1152x384').split('x'))
kn = int(os.environ.get('KN', '0'))
cfg = int(os.environ.get('CFG', '-1'))
A = torch.randn(M, K, device='cuda')
W = torch.randn((K, N) if kn else (N, K), device='cuda')
C = torch.empty(M, N, device='cuda')
secs = float(os.environ.get('SECS', '12'))
t0, n = time.time(), 0
while time.time() - t0 < secs:
    for _ in range(50):
        _lib.call('pdae_rows_gemm', A, M, N, K, _lib.ptr(A), _lib.ptr(W), kn, None, 0, None, _lib.ptr(C), cfg, 1, 0)
    torch.cuda.synchronize()
    n += 50
print('agg: %d launches, arith %s cfg %d shape %s kn %d' % (n, os.environ.get('ARITH', 'bf16x3'), cfg, (M, N, K), kn), flush=True)
