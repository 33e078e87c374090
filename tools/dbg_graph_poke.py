"""Which gradients change when one captured step graph is replayed before / after a null-stream D2H copy?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from point_dae_amd import builder
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.data_parallel import FlatDataParallel
from point_dae_amd.graph_step import GraphedTrainStep
from point_dae_amd.synthetic import shapenet_like_clouds
from point_dae_amd.misc import set_random_seed

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
cfg.npoints = 1024
dev = torch.device('cuda')
if os.environ.get('ONE_STREAM'):
    torch.cuda.set_stream(torch.cuda.Stream())
set_random_seed(0)
model = FlatDataParallel(builder.model_builder(cfg.model).to(dev))
opt, _ = builder.build_opti_sche(model, cfg)
model.train(); model.zero_grad()
x = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=7)).to(dev)
step = GraphedTrainStep(model, opt, cfg, B, 1024)
for _ in range(6):
    step(x)
torch.cuda.synchronize()
tv = step.last_tvis
g = step.graphs[tv]
side = torch.cuda.Stream()
names = model.names


def replay():
    if os.environ.get('ONE_STREAM'):
        g.replay()
        torch.cuda.current_stream().synchronize()
    else:
        with torch.cuda.stream(side):
            g.replay()
        side.synchronize()
    return step.outputs[tv][0].item(), model.flat_grad.clone()


l0, g0 = replay()
l1, g1 = replay()
for _ in range(3):
    print('extra replay loss', replay()[0])
print('repeat before poke: loss', l0, l1, 'max grad diff', (g0 - g1).abs().max().item())
mode = os.environ.get('POKE_MODE', 'null')
if mode == 'null':
    z = model.flat_param[:1 << 20].cpu()          # the poke: 4 MB device-to-host on the NULL stream
elif mode == 'side':
    with torch.cuda.stream(side):
        z = model.flat_param[:1 << 20].cpu()
elif mode == 'side2':
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        z = model.flat_param[:1 << 20].cpu()
elif mode == 'pinned_null':
    zz = torch.empty(1 << 20, pin_memory=True); zz.copy_(model.flat_param[:1 << 20]); torch.cuda.synchronize()
elif mode == 'h2d_null':
    zz = torch.randn(1 << 20); zd = zz.to(dev); torch.cuda.synchronize()
elif mode == 'd2d_null':
    zd = model.flat_param[:1 << 20].clone(); torch.cuda.synchronize()
elif mode == 'kernel_null':
    zd = torch.randn(1 << 20, device=dev) * 2; torch.cuda.synchronize()
if mode == 'sync_only':
    torch.cuda.synchronize()
print('poke mode', mode)
l2, g2 = replay()
print('after poke: loss', l2, 'max grad diff', (g0 - g2).abs().max().item(), 'nan', torch.isnan(g2).sum().item())
bad = []
for (off, n), name in zip(model.offsets, names):
    a, b = g0[off:off + n], g2[off:off + n]
    d = (a - b).abs().max().item()
    s = a.abs().max().item() + 1e-20
    if not (d / s < 1e-3):
        bad.append((name, d / s))
print(len(bad), 'of', len(names), 'parameter gradients differ; first:', bad[:8], '... last:', bad[-4:])
