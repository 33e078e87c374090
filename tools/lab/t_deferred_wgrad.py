import sys; sys.path.insert(0, '/root/repo')
import torch
from point_dae_amd import _lib, nn_ops
torch.manual_seed(0)
def mk(M, dims):
    return [torch.randn(M, n, device='cuda') for n, _ in dims], [torch.randn(M, k, device='cuda') for _, k in dims]
cases = [(3584, [(1152, 384), (384, 384), (1536, 384), (384, 1536)]), (8192, [(384, 128), (128, 4)]), (2048, [(96, 384)]),
         (5376, [(384, 384), (1536, 384), (384, 1536)]), (8192, [(1152, 384)])] * 3
data = [mk(M, d) for M, d in cases]
ref = [nn_ops.rows_wgrad(dy, x, [True] * len(dy)) for dy, x in data]
_lib.deferred_begin()
out = [nn_ops.rows_wgrad(dy, x, [True] * len(dy)) for dy, x in data]
_lib.deferred_flush(data[0][0][0])
torch.cuda.synchronize()
bad = 0
for (rw, rb), (ow, ob) in zip(ref, out):
    for a, b in zip(rw + rb, ow + ob):
        if not torch.equal(a, b):
            bad += 1
            print('mismatch', tuple(a.shape), (a - b).abs().max().item())
print('jobs', len(cases), 'bad', bad)
