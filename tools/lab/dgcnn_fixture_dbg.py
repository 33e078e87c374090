"""Per-tensor gradient deviations of the GPU DGCNN model against the CPU oracle on a fixture's inputs (debug aid)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import numpy as np, torch
from weights import fill_state
from oracle import model as OM
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.point_cae_dgcnn import Point_CAE_DGCNN_FCOnly
name, items = sys.argv[1], sys.argv[2].split(',')
fx = np.load(os.path.join(ROOT, 'tests', 'golden', name))
cfg = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs', 'pretrain_PointCAE_clean.yaml')).model
cfg.NAME, cfg.corrupt_type = 'Point_CAE_DGCNN_FCOnly', items
seed = int(fx['seed'])
gm = fill_state(Point_CAE_DGCNN_FCOnly(cfg), seed).cuda().train()
om = fill_state(OM.Point_CAE_DGCNN_FCOnly(cfg), seed).train()
for m, dev in ((gm, 'cuda'), (om, 'cpu')):
    random.seed(seed + 7); torch.manual_seed(seed + 7)
    loss, _ = m(torch.from_numpy(fx['corrupted']).to(dev), torch.from_numpy(fx['clean']).to(dev))
    loss.backward()
    print(dev, loss.item())
rows = []
for (n, p), (_, q) in zip(gm.named_parameters(), om.named_parameters()):
    a, b = p.grad.cpu().double(), q.grad.double()
    rows.append(((a - b).abs().max().item() / max(b.abs().max().item(), 1e-30), (a - b).norm().item() / max(b.norm().item(), 1e-30), n))
for r in sorted(rows, reverse=True)[:12]:
    print('max-rel %.2e  l2-rel %.2e  %s' % r)
# layer-1 graph of the GPU path against fp64 top-k on the same (dropped) cloud
from point_dae_amd.corrupt_util_tensor import corrupt_in_forward
from point_dae_amd.point_cae_dgcnn import feature_knn
random.seed(seed + 7); torch.manual_seed(seed + 7)
pts = corrupt_in_forward(torch.from_numpy(fx['corrupted']).cuda()[:, :, :3].contiguous(), items)
B, N, _ = pts.shape
x4 = torch.nn.functional.pad(pts.reshape(B * N, 3), (0, 1)).contiguous()
idx = feature_knn(x4, B, N, 20).long()
xd = pts.double()
pd = -(xd.unsqueeze(2) - xd.unsqueeze(1)).square().sum(-1)
tv, ti = pd.topk(21, dim=-1)
same = (idx.sort(-1)[0] == ti[:, :, :20].sort(-1)[0]).all(-1)
gap = (tv[:, :, 19] - tv[:, :, 20])
print('N', N, 'rows whose neighbour SET differs from fp64 top-20:', (~same).sum().item(), 'of', B * N,
      '; their 20th-21st gaps:', gap[~same].abs().topk(min(5, int((~same).sum().item())), largest=True)[0].tolist() if (~same).any() else [])
print('rows with an exact 20th/21st tie in fp64:', (gap == 0).sum().item())
# every layer's graph against fp64 top-k of the SAME layer input, and how many rows sit on a near-tie
from point_dae_amd import point_cae_dgcnn as D
orig = D.feature_knn
def probe(x, B, N, k=20):
    idx = orig(x, B, N, k)
    xd = x.double().view(B, N, -1)
    pd = -(xd.unsqueeze(2) - xd.unsqueeze(1)).square().sum(-1)
    tv, ti = pd.topk(k + 1, dim=-1)
    same = (idx.long().sort(-1)[0] == ti[:, :, :k].sort(-1)[0]).all(-1)
    gap = (tv[:, :, k - 1] - tv[:, :, k]) / tv[:, :, k].abs().clamp_min(1e-30)
    print('  layer input C=%d: rows differing from fp64 top-k %d of %d; rows with relative 20th/21st gap < 1e-5: %d, < 1e-6: %d' % (
        x.shape[1], (~same).sum().item(), B * N, (gap < 1e-5).sum().item(), (gap < 1e-6).sum().item()))
    return idx
D.feature_knn = probe
random.seed(seed + 7); torch.manual_seed(seed + 7)
gm(torch.from_numpy(fx['corrupted']).cuda(), torch.from_numpy(fx['clean']).cuda())
# the oracle's graphs (fp32 on the CPU) against the product's, layer by layer
D.feature_knn = orig
pg, og = [], []
D.feature_knn = lambda *a: pg.append(orig(*a)) or pg[-1]
oknn = OM.dgcnn_knn
OM.dgcnn_knn = lambda x, k: og.append(oknn(x, k)) or og[-1]
for m, dev in ((gm, 'cuda'), (om, 'cpu')):
    random.seed(seed + 7); torch.manual_seed(seed + 7)
    m(torch.from_numpy(fx['corrupted']).to(dev), torch.from_numpy(fx['clean']).to(dev))
for li, (a, b) in enumerate(zip(pg, og)):
    same = (a.long().cpu().sort(-1)[0] == b.sort(-1)[0]).all(-1)
    print('  layer %d: rows whose neighbour set differs between product and oracle: %d of %d' % (li + 1, (~same).sum().item(), same.numel()))
