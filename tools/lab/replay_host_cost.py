"""LAB: host cost of ONE hipGraphLaunch of the step's graph with the GPU idle (no queue back-pressure), against the
steady-state figure of tools/host_time.py (6.5-6.9 ms per replay with the host running ahead)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from point_dae_amd import builder
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.data_parallel import FlatDataParallel
from point_dae_amd.graph_step import GraphedTrainStep, use_created_stream
from point_dae_amd.synthetic import shapenet_like_clouds
from point_dae_amd.misc import set_random_seed

use_created_stream()
cfg = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
set_random_seed(0)
model = FlatDataParallel(builder.model_builder(cfg.model).cuda())
opt, _ = builder.build_opti_sche(model, cfg)
model.train()
B = 128
x = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=1)).cuda()
step = GraphedTrainStep(model, opt, cfg, B, 1024)
for _ in range(30):
    step(x)
torch.cuda.synchronize()
idle, busy = [], []
for _ in range(20):
    step.pts.copy_(x, non_blocking=True)
    tv = step._draw()
    g = step.graphs.get(tv) or step._capture(tv)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()      # GPU idle when the launch starts
    idle.append(t1 - t0)
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()      # one replay already queued
    busy.append(t1 - t0)
    t0 = time.perf_counter(); g.replay(); t1 = time.perf_counter()      # two queued
    busy.append(t1 - t0)
    torch.cuda.synchronize()
idle.sort(), busy.sort()
print('hipGraphLaunch of the step graph, host side: GPU idle median %.2f ms (min %.2f); behind queued replays median %.2f ms (max %.2f)' % (
    idle[len(idle) // 2] * 1e3, idle[0] * 1e3, busy[len(busy) // 2] * 1e3, busy[-1] * 1e3))

# the whole host side of a step (H2D staging copy, draws, hipGraphLaunch, optimiser bookkeeping), iteration by iteration
# from an idle GPU: the first iterations show the CPU work, later ones the wait for queue space
os.environ['PDAE_RING'] = '64'
torch.cuda.synchronize()
its = []
for i in range(48):
    t0 = time.perf_counter()
    step(x)
    its.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print('host ms per step(x) call, iterations 1..48 from an idle GPU:')
print(' '.join('%.2f' % v for v in its))
