"""Run INSIDE a checkout of a commit that still shows the NULL-stream failure (<= 03b7f3c; `git worktree add .old 03b7f3c`,
build its csrc, copy this file to .old/tools/): after a device-to-host copy on the NULL stream, are the per-step
host-to-device copies of the host draws (steps / vis / msk) still delivered, are parameters / inputs intact, and which
replays go wrong?  Result on MI355X, ROCm 7.2: every buffer intact after each step, but replays are SPORADICALLY wrong
from the first device-to-host copy on (steps 49, 50, 55, 58 of 46..59: losses 1.8e3, 55, 5.2e2, 7e25; the others
normal) -- a race around the replay, not a corrupted pool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, warnings
warnings.simplefilter('ignore')
from point_dae_amd import builder
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.data_parallel import FlatDataParallel
from point_dae_amd.graph_step import GraphedTrainStep
from point_dae_amd.synthetic import shapenet_like_clouds
from point_dae_amd.misc import set_random_seed
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
cfg.npoints = 1024
dev = torch.device('cuda'); B = 128
set_random_seed(0)
model = FlatDataParallel(builder.model_builder(cfg.model).to(dev))
opt, _ = builder.build_opti_sche(model, cfg)
model.train(); model.zero_grad()
pool = torch.from_numpy(shapenet_like_clouds(B * 16, 1024, seed=7)).to(dev).split(B)
step = GraphedTrainStep(model, opt, cfg, B, 1024)                  # on the legacy NULL stream
for i in range(70):
    lx, _ = step(pool[i % len(pool)])
    if 46 <= i < 60:                                               # (the checks below ARE device-to-host copies)
        torch.cuda.synchronize()
        slot = step.ring[(step.slot - 1) % step.RING]
        ok = [torch.equal(getattr(step, k).cpu(), slot[k]) for k in ('steps', 'vis', 'msk')]
        print(i, 'loss %.5f' % lx.item(), 'H2D delivered', ok, 'grads finite', torch.isfinite(model.flat_grad).all().item(),
              'pts ok', torch.equal(step.pts, pool[i % len(pool)]), flush=True)
