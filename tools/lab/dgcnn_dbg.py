"""Layer-by-layer comparison of the EdgeConv kernels with the dense formulation (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from point_dae_amd import _lib, nn_ops
from point_dae_amd.point_cae_dgcnn import feature_knn, _parts
torch.manual_seed(0)
B, N, C, co, k = 2, 256, 64, 64, 20
R = B * N
x = torch.randn(R, C, device='cuda')
w = torch.randn(co, 2 * C, device='cuda') / (2 * C) ** 0.5
gamma = torch.linspace(-1, 1.5, co, device='cuda'); gamma[::7] = 0
beta = torch.linspace(-.3, .3, co, device='cuda')
idx = feature_knn(x, B, N, k)
ws = torch.cat([w[:, :C], w[:, C:] - w[:, :C]], 0).contiguous()
pq = nn_ops.rows_gemm(x, ws)
print('pq err', (pq - x @ ws.t()).abs().max().item())
esel, psum = torch.empty(R, co, device='cuda'), torch.empty(R, co, device='cuda')
sel = torch.empty(R, co, device='cuda', dtype=torch.int16)
part, sums = _parts(x, co)
_lib.call('pdae_edge_gather_stats', x, B, N, k, co, pq.data_ptr(), idx.data_ptr(), gamma.data_ptr(), esel.data_ptr(), sel.data_ptr(), psum.data_ptr(), part.data_ptr(), sums.data_ptr())
flat = (idx.long() + torch.arange(B, device='cuda').view(-1, 1, 1) * N).reshape(-1)
p, q = pq[:, :co], pq[:, co:]
e = p.index_select(0, flat).view(R, k, co) + q.unsqueeze(1)
print('sum e', (sums[:co] - e.double().sum((0, 1))).abs().max().item(), 'sum e2', (sums[co:] - e.double().square().sum((0, 1))).abs().max().item() / e.double().square().sum((0,1)).max().item())
print('psum', (psum - p.index_select(0, flat).view(R, k, co).sum(1)).abs().max().item())
emax, emin = e.max(1)[0], e.min(1)[0]
want = torch.where(gamma > 0, emax, torch.where(gamma < 0, emin, e[:, 0]))
print('esel', (esel - want).abs().max().item())
for name, m in (('pos', gamma > 0), ('neg', gamma < 0), ('zero', gamma == 0)):
    print(name, (esel - want)[:, m].abs().max().item(), 'vs emax', (esel - emax)[:, m].abs().max().item(), 'vs emin', (esel - emin)[:, m].abs().max().item(), 'vs e0', (esel - e[:, 0])[:, m].abs().max().item(), 'vs elast', (esel - e[:, -1])[:, m].abs().max().item())
