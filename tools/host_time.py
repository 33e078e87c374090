"""Host-side cost of one graphed step (no device sync inside): where does the CPU time go?"""
import os, sys, time
os.environ.setdefault('PDAE_RING', '64')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from point_dae_amd import builder
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.data_parallel import FlatDataParallel
from point_dae_amd.graph_step import GraphedTrainStep
from point_dae_amd.synthetic import shapenet_like_clouds
from point_dae_amd.misc import set_random_seed

cfg = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
cfg.npoints = 1024
dev = torch.device('cuda')
set_random_seed(0)
model = FlatDataParallel(builder.model_builder(cfg.model).to(dev))
opt, _ = builder.build_opti_sche(model, cfg)
model.train()
B = 128
x = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=1)).to(dev)
step = GraphedTrainStep(model, opt, cfg, B, 1024)
for _ in range(30):
    step(x)
torch.cuda.synchronize()
T = dict(copy=0.0, draw=0.0, replay=0.0, opt=0.0)
n = 40
t_all = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter(); step.pts.copy_(x, non_blocking=True)
    t1 = time.perf_counter(); tv = step._draw()
    t2 = time.perf_counter()
    g = step.graphs.get(tv) or step._capture(tv)
    g.replay()
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter()
    T['copy'] += t1 - t0; T['draw'] += t2 - t1; T['replay'] += t3 - t2; T['opt'] += t4 - t3
host = time.perf_counter() - t_all
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
print('host ms/step', host / n * 1e3, 'wall ms/step', tot / n * 1e3, {k: round(v / n * 1e3, 3) for k, v in T.items()})
