import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import copy, torch, torch.nn as nn
from point_dae_amd.patch_embed import patch_embed
from test_gpu_gemm import _embed_reference
torch.manual_seed(0)
first = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True), nn.Conv1d(128, 256, 1)).cuda()
second = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, 384, 1)).cuda()
for m in list(first) + list(second):
    if isinstance(m, nn.BatchNorm1d):
        m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
for BG in (256, 8192):
    first_r, second_r = copy.deepcopy(first), copy.deepcopy(second)
    first_d, second_d = copy.deepcopy(first).double(), copy.deepcopy(second).double()
    f2, s2 = copy.deepcopy(first), copy.deepcopy(second)
    pts = torch.randn(BG, 32, 3, device='cuda') * 0.2
    go = torch.randn(BG, 384, device='cuda')
    out = patch_embed(pts, f2, s2, True); out.backward(go)
    ref = _embed_reference(pts, first_r, second_r); ref.backward(go)
    refd = _embed_reference(pts.double(), first_d, second_d); refd.backward(go.double())
    print('BG', BG, 'out err mine', ((out-refd).abs().max()/refd.abs().max()).item(), 'torch32', ((ref-refd).abs().max()/refd.abs().max()).item())
    names=[n for n,_ in list(first.named_parameters())+list(second.named_parameters())]
    for n,a,b,c in zip(names, list(f2.parameters())+list(s2.parameters()), list(first_r.parameters())+list(second_r.parameters()), list(first_d.parameters())+list(second_d.parameters())):
        sc=c.grad.abs().max().item()
        print('  %-10s scale %.3e  mine-vs-f64 %.2e   torch32-vs-f64 %.2e' % (n, sc, (a.grad.double()-c.grad).abs().max().item()/sc, (b.grad.double()-c.grad).abs().max().item()/sc))
