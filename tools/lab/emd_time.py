"""LAB: EMD kernel times at the two sizes VERDICT r3 names (8 clouds of 1024 x 1024; 5248 patches of 32 x 32)."""
import sys
import torch
sys.path.insert(0, '.')
from point_dae_amd import emd
from point_dae_amd.synthetic import shapenet_like_clouds


def t(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


x = torch.from_numpy(shapenet_like_clouds(16, 1024, seed=1)).cuda()
a, b = x[:8].contiguous(), x[8:].contiguous()
print("8x1024x1024 approxmatch us", t(lambda: emd.approxmatch_forward(a, b)))
m = emd.approxmatch_forward(a, b)
print("  matchcost us", t(lambda: emd.matchcost_forward(a, b, m)))
gc = torch.ones(8, device="cuda")
print("  matchcost_grad us", t(lambda: emd.matchcost_backward(gc, a, b, m)))
p = torch.randn(5248, 32, 3, device="cuda")
q = torch.randn(5248, 32, 3, device="cuda")
print("5248x32x32 approxmatch us", t(lambda: emd.approxmatch_forward(p, q)))
mm = emd.approxmatch_forward(p, q)
print("  matchcost us", t(lambda: emd.matchcost_forward(p, q, mm)))
gc = torch.ones(5248, device="cuda")
print("  matchcost_grad us", t(lambda: emd.matchcost_backward(gc, p, q, mm)))
