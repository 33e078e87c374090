import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import copy, torch, torch.nn as nn
from point_dae_amd.patch_embed import patch_embed
torch.manual_seed(0)
first = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True), nn.Conv1d(128, 256, 1)).cuda()
second = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, 384, 1)).cuda()
for BG in (256, 8192):
    pts = torch.randn(BG, 32, 3, device='cuda') * 0.2
    sel = torch.arange(0, BG, 3, device='cuda', dtype=torch.int32)
    gsel = torch.randn(sel.numel(), 384, device='cuda')
    res=[]
    for mode in ('full','compact','full2'):
        f, s = copy.deepcopy(first), copy.deepcopy(second)
        if mode.startswith('full'):
            out = patch_embed(pts, f, s, True)[sel.long()]
        else:
            out = patch_embed(pts, f, s, True, sel)
        out.backward(gsel)
        res.append([p.grad.clone() for p in list(f.parameters())+list(s.parameters())])
    names=[n for n,_ in list(first.named_parameters())+list(second.named_parameters())]
    for i,n in enumerate(names):
        sc=res[0][i].abs().max().item()+1e-12
        print(BG, n, 'scale %.2e full-vs-compact %.2e  full-vs-full2 %.2e'%(sc,(res[0][i]-res[1][i]).abs().max().item()/sc,(res[0][i]-res[2][i]).abs().max().item()/sc))
