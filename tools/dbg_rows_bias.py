import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import nn_ops
from point_dae_amd.graph_step import use_created_stream
use_created_stream()
torch.manual_seed(0)
for M, N, K in ((32768, 64, 4), (16384, 128, 132), (32768, 64, 64)):
    x = torch.randn(M, K, device='cuda') * 0.1
    w = torch.randn(N, K, device='cuda')
    y = nn_ops.rows_gemm(x, w)
    yl = x @ w.t()
    ref = x.double() @ w.double().t()
    em, el = (y.double() - ref), (yl.double() - ref)
    print('M%d N%d K%d: mine mean err %.2e rms %.2e | lib mean err %.2e rms %.2e | mine==lib fraction %.3f' % (
        M, N, K, em.mean().item(), em.pow(2).mean().sqrt().item(), el.mean().item(), el.pow(2).mean().sqrt().item(),
        (y == yl).float().mean().item()))
    # identical rows must give identical outputs wherever they sit
    x2 = x[:1].expand(M, K).contiguous()
    y2 = nn_ops.rows_gemm(x2, w)
    print('   identical rows -> identical outputs:', bool((y2 == y2[:1]).all()), ' first row equals row of full run:', bool((y2[0] == y[0]).all()))
    yl2 = x2 @ w.t()
    print('   library: identical rows -> identical outputs:', bool((yl2 == yl2[:1]).all()))
