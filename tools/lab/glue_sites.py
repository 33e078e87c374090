"""Which Python lines issue the framework (ATen) launches of one step?  (GPU box)
    python tools/lab/glue_sites.py > gpurun_out/glue_sites.txt
A TorchDispatchMode around ONE eager pass of the graphed step's body (B=128, split as bench.py runs it) logs every ATen op that
is not a view / metadata op, with its shapes and the innermost frames inside point_dae_amd/ -- tools/lab/glue_trace.py gives the
device times of the same ops, this gives the call sites (torch.profiler's with_stack comes back empty on this build)."""
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from point_dae_amd import builder  # noqa: E402
from point_dae_amd.config import cfg_from_yaml_file  # noqa: E402
from point_dae_amd.data_parallel import FlatDataParallel  # noqa: E402
from point_dae_amd.graph_step import GraphedTrainStep, use_created_stream  # noqa: E402
from point_dae_amd.synthetic import shapenet_like_clouds  # noqa: E402

VIEWS = ('view', 'reshape', 'expand', 'unsqueeze', 'squeeze', 'transpose', 't.default', 'slice', 'select', 'detach', 'alias',
         'as_strided', 'permute', 'empty', 'unbind', 'split', 'narrow', '_unsafe_view', 'size', 'stride', 'numel', 'is_',
         'sym_', 'dim', 'item', '_local_scalar', 'lift_fresh', 'unflatten', 'flatten')


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        short = name.split('aten.')[-1]
        if not any(short.startswith(v) for v in VIEWS):
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            if args and isinstance(args[0], (list, tuple)):
                shapes = ['%d tensors' % len(args[0])]
            frames = [f for f in traceback.extract_stack() if 'point_dae_amd' in f.filename][-3:]
            self.rows.append((short, shapes, ['%s:%d %s' % (os.path.basename(f.filename), f.lineno, f.name) for f in frames]))
        return func(*args, **(kwargs or {}))


use_created_stream()
what = sys.argv[1] if len(sys.argv) > 1 else 'cfg3'
B = 128
x = torch.from_numpy(shapenet_like_clouds(2 * B, 1024, seed=1)).cuda()
if what == 'cfg2':
    from point_dae_amd.graph_step import GraphedStaticStep  # noqa: E402
    config = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs', 'pretrain_PointCAE_affine_r3_dropout_local_4xlonger.yaml'))
    model = FlatDataParallel(builder.model_builder(config.model).cuda())
    opt, _ = builder.build_opti_sche(model, config)
    model.train()
    step = GraphedStaticStep(model, opt, lambda a, b: a + float(config.normal_weight) * b * 0.5, B, 1024)
    step.corrupted.copy_(x[:B]), step.clean.copy_(x[B:])
    body = step._fwd_bwd
else:
    config = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    if what != 'cfg3':
        config.model.NAME = what
    model = FlatDataParallel(builder.model_builder(config.model).cuda())
    opt, _ = builder.build_opti_sche(model, config)
    model.train()
    step = GraphedTrainStep(model, opt, config, B, 1024)
    step.pts.copy_(x[:B])
    body = lambda: step._fwd_bwd(step._draw())  # noqa: E731
for _ in range(2):
    body()
torch.cuda.synchronize()
with Sites() as s:
    body()
torch.cuda.synchronize()
for short, shapes, frames in s.rows:
    print('%-34s %-70s %s' % (short[:34], str(shapes)[:70], ' < '.join(reversed(frames))))
print('ops', len(s.rows), what)
