"""NULL-stream hazard, narrowed: at HEAD the captured step holds no library GEMM and survives every
NULL-stream action (tools/soak.py matrix).  This puts ONE kind of library GEMM back (the plain-store row GEMMs become
torch.matmul = rocBLAS / hipBLASLt) and repeats the failing sequence of round 1: steps on the NULL stream, a
device-to-host copy of the parameters at step 50.
    python tools/lab/null_stream_lib_gemm.py [lib|own] [cpu|sync|none]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from point_dae_amd import builder, nn_ops
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.data_parallel import FlatDataParallel
from point_dae_amd.graph_step import GraphedTrainStep
from point_dae_amd.synthetic import shapenet_like_clouds
from point_dae_amd.misc import set_random_seed
which, poke = (sys.argv + ['lib', 'cpu'])[1:3]
if which == 'lib':
    own = nn_ops.rows_gemm

    def rows_gemm(x, w, w_kn=False, bias=None, epi=0, z=None, may_split=False, big_cfg=None):
        if epi == 0 and bias is None:
            return x @ (w if w_kn else w.t())                # the library GEMM (no split-K slabs)
        return own(x, w, w_kn, bias, epi, z, may_split, big_cfg)
    nn_ops.rows_gemm = rows_gemm
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs/pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
cfg.npoints = 1024
dev = torch.device('cuda')
set_random_seed(0)
B = 128
model = FlatDataParallel(builder.model_builder(cfg.model).to(dev))
opt, _ = builder.build_opti_sche(model, cfg)
model.train(); model.zero_grad()
pool = torch.from_numpy(shapenet_like_clouds(B * 16, 1024, seed=7)).to(dev).split(B)
import warnings; warnings.simplefilter('ignore')
step = GraphedTrainStep(model, opt, cfg, B, 1024)            # on the legacy NULL stream, deliberately
acc = torch.zeros((), device=dev)
out = []
for i in range(100):
    if i == 50:
        if poke == 'cpu': z = model.flat_param.cpu()
        if poke == 'sync': torch.cuda.synchronize()
    lx, _ = step(pool[i % len(pool)])
    acc += lx.reshape(())
    if (i + 1) % 25 == 0:
        out.append('%.3f' % (acc.item() / 25 * 1000)); acc.zero_()
print(which, 'GEMMs, poke', poke, ': loss*1000 per 25 steps', out)
