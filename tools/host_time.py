"""Host-side cost of one graphed step (no device sync inside): where does the CPU time go?"""
import os, sys, time
os.environ.setdefault('PDAE_RING', '64')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from point_dae_amd import builder
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.data_parallel import FlatDataParallel
from point_dae_amd.graph_step import GraphedTrainStep, use_created_stream
from point_dae_amd.synthetic import shapenet_like_clouds
from point_dae_amd.misc import set_random_seed

cfg = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
cfg.npoints = 1024
dev = torch.device('cuda')
use_created_stream()      # as the runner and bench.py do: graphs launched beside legacy NULL-stream work are ~50x slower to launch
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
# CPU work of a step = what the host does while it is NOT waiting: measured on the first iterations from an idle GPU.
# (Averaging many iterations measures something else: once ~18 replays are queued the hardware queue is full and
# hipGraphLaunch blocks until the GPU has drained one -- round 2's "6.7 ms in hipGraphLaunch" was that wait.)
import statistics
rows = []
n = 40
torch.cuda.synchronize()
t_all = time.perf_counter()
for _ in range(n):
    t0 = time.perf_counter(); step.pts.copy_(x, non_blocking=True)
    t1 = time.perf_counter(); tv = step._draw()
    t2 = time.perf_counter()
    g = step.graphs.get(tv) or step._capture(tv)
    g.replay()
    t3 = time.perf_counter(); opt.step()
    t4 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
host = time.perf_counter() - t_all
torch.cuda.synchronize()
tot = time.perf_counter() - t_all
first = rows[:8]
med = lambda k, rs: round(statistics.median(r[k] for r in rs) * 1e3, 3)
names = ('copy', 'draw', 'replay', 'opt')
print('CPU work per step (first 8 iterations, GPU idle at the start):', {nm: med(k, first) for k, nm in enumerate(names)},
      'total %.3f ms' % sum(med(k, first) for k in range(4)))
print('all %d iterations: host %.2f ms/step of which waiting for queue space; wall %.2f ms/step; replay calls that blocked (> 2 ms): %d' % (
    n, host / n * 1e3, tot / n * 1e3, sum(1 for r in rows if r[2] > 2e-3)))
