"""Accuracy probe: row GEMMs / wgrad vs torch fp32 and fp64 on the set-abstraction shapes, with
cancellation-heavy operands (zero-mean columns, as behind a training-mode BatchNorm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import nn_ops
from point_dae_amd.graph_step import use_created_stream
use_created_stream()
torch.manual_seed(0)


def err(a, ref):
    return ((a.double() - ref).abs().max() / ref.abs().max()).item()


for M, N, K in ((32768, 64, 4), (16384, 128, 132), (16384, 128, 128), (256, 256, 260), (32768, 64, 64)):
    x = torch.randn(M, K, device='cuda') + 1.0
    if K in (4, 132, 260):
        x[:, 3] = 0
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    dy = torch.randn(M, N, device='cuda')
    dy = dy - dy.mean(0, keepdim=True)                       # zero-sum columns
    xd, wd, dyd = x.double(), w.double(), dy.double()
    y = nn_ops.rows_gemm(x, w)
    dx = nn_ops.rows_gemm(dy, w, True)
    (dw,), _ = nn_ops.rows_wgrad([dy], [x], [False])
    print('M%6d N%4d K%4d | fwd mine %.1e torch %.1e | dX mine %.1e torch %.1e | dW mine %.1e torch %.1e' % (
        M, N, K, err(y, xd @ wd.t()), err(x @ w.t(), xd @ wd.t()), err(dx, dyd @ wd), err(dy @ w, dyd @ wd),
        err(dw, dyd.t() @ xd), err(dy.t() @ x, dyd.t() @ xd)))
