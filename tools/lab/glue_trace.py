"""Which Python lines launch the non-pdae (PyTorch glue) kernels of one step?  (GPU box)
    python tools/lab/glue_trace.py > gpurun_out/glue_trace.txt
torch.profiler with stacks over ONE eager forward+backward of the graphed step's body (B=128); prints every ATen op
that launched device work together with the innermost frames inside point_dae_amd/."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import builder  # noqa: E402
from point_dae_amd.config import cfg_from_yaml_file  # noqa: E402
from point_dae_amd.data_parallel import FlatDataParallel  # noqa: E402
from point_dae_amd.graph_step import GraphedTrainStep, use_created_stream  # noqa: E402
from point_dae_amd.synthetic import shapenet_like_clouds  # noqa: E402

use_created_stream()
config = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
if len(sys.argv) > 1:
    config.model.NAME = sys.argv[1]
model = FlatDataParallel(builder.model_builder(config.model).cuda())
opt, _ = builder.build_opti_sche(model, config)
model.train()
B = 128
x = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=1)).cuda()
step = GraphedTrainStep(model, opt, config, B, 1024, split=False)
step.pts.copy_(x)
for _ in range(3):
    tvis = step._draw()
    step._phase1(tvis)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tvis = step._draw()
    step._phase1(tvis)
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if ev.device_type.name != 'CPU' or not ev.kernels:
        continue
    kn = [k.name for k in ev.kernels]
    if all(k.startswith('pdae::') or 'pdae' in k for k in kn):
        continue
    frames = [f for f in (ev.stack or []) if 'point_dae_amd' in f or 'bench' in f][:3]
    rows.append((ev.time_range.start, ev.name + ' ' + str(getattr(ev, 'input_shapes', ''))[:90], [k[:60] for k in kn], sum(k.duration for k in ev.kernels), frames))
rows.sort()
tot = 0.0
for t, name, kn, dur, frames in rows:
    tot += dur
    print('%-120s %6.1f us  %s' % (name[:120], dur, ' | '.join(f.replace(ROOT + '/', '') for f in frames)))
    for k in kn:
        print('      -> ' + k)
print('glue launches', sum(len(r[2]) for r in rows), 'device us', tot, 'tvis', tvis)
