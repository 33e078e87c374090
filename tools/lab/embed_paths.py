"""Diagnostic (GPU box): the embedder's gradients along its three product paths against the CPU oracle Encoder."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import model as OM
from point_dae_amd.graph_step import use_created_stream
from point_dae_amd.patch_embed import patch_embed_layerwise
from point_dae_amd.point_cae_transformer import Encoder
use_created_stream()
torch.manual_seed(3)
enc = Encoder(384).train()
for bn in (enc.first_conv[1], enc.second_conv[1]):
    torch.nn.init.uniform_(bn.weight, 0.5, 1.5), torch.nn.init.uniform_(bn.bias, -0.2, 0.2)
g = torch.Generator().manual_seed(11)
for BG, scale in ((64, 0.2), (256, 0.2), (64, 1.0)):
    pts = torch.randn(BG, 32, 3, generator=g) * scale
    vis = torch.arange(0, BG, 2, dtype=torch.int32)
    msk = torch.arange(1, BG, 2, dtype=torch.int32)
    W = torch.randn(BG // 2, 384, generator=g)
    o = OM.Encoder(384).train(); o.load_state_dict(enc.state_dict())
    tok = o(pts.reshape(1, BG, 32, 3))[0][vis.long()]
    (tok * W).sum().backward()
    og = {n: p.grad.clone() for n, p in o.named_parameters()}
    gmax = max(v.abs().max().item() for v in og.values())

    def run(kind):
        e = Encoder(384).cuda().train(); e.load_state_dict(enc.state_dict())
        p = pts.cuda().reshape(1, BG, 32, 3)
        if kind == 'fused_dense':
            t = e(p, groups=vis.cuda())
        elif kind == 'fused_algebra':
            t = e(p, groups=vis.cuda(), masked=msk.cuda())
        else:
            t = patch_embed_layerwise(p.reshape(BG, 32, 3), e.first_conv, e.second_conv, vis.cuda())
        (t * W.cuda()).sum().backward()
        worst = max(((q.grad.cpu() - og[n]).abs().max().item() / max(og[n].abs().max().item(), 1e-3 * gmax), n)
                    for n, q in e.named_parameters())
        return (t.detach().cpu() - tok).abs().max().item() / tok.abs().max().item(), worst
    print('BG', BG, 'scale', scale, {k: run(k) for k in ('fused_dense', 'fused_algebra', 'layerwise')})
