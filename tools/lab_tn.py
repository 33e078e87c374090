"""Lab: the embedder's three weight-gradient GEMMs (TN kernel) in isolation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import _lib

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

R = 262144
for name, M, N, K, bn, gl in (('dW4 visible', 94208, 384, 512, True, True), ('dW3 local', R, 512, 256, False, False),
                              ('dW2', R, 256, 128, True, False)):
    dy = torch.randn(M, N, device='cuda')
    x = torch.randn(R, K, device='cuda')
    sc = torch.rand(K, device='cuda') + 0.5; sh = torch.randn(K, device='cuda')
    dw = torch.empty(N, K, device='cuda')
    groups = torch.randperm(R // 32, device='cuda')[:M // 32].sort()[0].int() if gl else None
    if bn:
        f = lambda: _lib.call('pdae_bnrelu_linear_backward_weight', x, M, N, K, dy.data_ptr(), x.data_ptr(), sc.data_ptr(), sh.data_ptr(), dw.data_ptr(), None, _lib.ptr(groups))
    else:
        f = lambda: _lib.call('pdae_linear_backward_weight', x, M, N, K, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None)
    t = timeit(f)
    xa = (x if groups is None else x.view(-1, 32, K)[groups.long()].reshape(-1, K))
    a = torch.relu(xa * sc + sh) if bn else xa
    ref = dy.t() @ a
    f()
    err = ((dw - ref).abs().max() / ref.abs().max()).item()
    t2 = timeit(lambda: torch.mm(dy.t(), a))
    print(f"{name}: {t:7.1f} us = {2.0*M*N*K/t/1e6:6.1f} TF   (torch.mm on materialised A: {t2:7.1f} us)  err {err:.1e}")
