"""LAB: shader clock right behind a run of SHIPPED kernels (tools/lab/clock_probe.hip): 200 back-to-back launches of one
kernel, then a one-wave probe on the same stream."""
import ctypes
import os
import subprocess
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(here)))
from point_dae_amd import _lib, nn_ops  # noqa: E402

so = os.path.join(here, 'libclock_probe.so')
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-shared', '-fPIC', os.path.join(here, 'clock_probe.hip'), '-o', so])
P = ctypes.CDLL(so)
P.clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
out = torch.zeros(2, dtype=torch.int64, device='cuda')


def probe(tag, fn, n=200):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(n):
        fn()
    P.clock_probe(out.data_ptr(), s)
    torch.cuda.synchronize()
    c, r = out.tolist()
    print(f"{tag:60s} clock right behind it {c / (r * 10.0):.2f} GHz", flush=True)


probe('idle (nothing before the probe)', lambda: None, 0)
for (M, N, K) in [(3584, 1152, 384), (8192, 1536, 384), (65536, 512, 512)]:
    x, w = torch.randn(M, K, device='cuda'), torch.randn(N, K, device='cuda') * 0.05
    probe(f'rows_gemm {M} x {N} x {K} (shipped, 64x64 tiles)', lambda: nn_ops.rows_gemm(x, w))
M = 2944
dims = [(1152, 384), (384, 384), (1536, 384), (384, 1536)] * 12
jobs = [(torch.randn(M, n, device='cuda'), torch.randn(M, k, device='cuda'), torch.empty(n, k, device='cuda'), None) for n, k in dims]
probe('rows_wgrad_multi, encoder stack (128x384 tiles)', lambda: _lib.rows_wgrad_multi(jobs), 20)
