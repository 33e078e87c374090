"""Step time of Point_CAE_DGCNN_FCOnly, B=32, N=1024: round 3's host code (framework top-k / gather / BatchNorm /
LeakyReLU / max over a (B N 20, C) tensor; tools/lab/archive/point_cae_dgcnn_r03.py) against csrc/dgcnn.hip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'lab', 'archive'))
import torch
from point_dae_amd import builder
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.data_parallel import FlatDataParallel
from point_dae_amd.graph_step import GraphedStaticStep, use_created_stream
from point_dae_amd.synthetic import shapenet_like_clouds
import point_cae_dgcnn_r03 as old
from point_dae_amd.point_cae_dgcnn import Point_CAE_DGCNN_FCOnly as New
dev = torch.device('cuda', 0)
use_created_stream(dev)
cfg = cfg_from_yaml_file(os.path.join(ROOT, 'cfgs', 'pretrain_PointCAE_affine_r3_dropout_local_4xlonger.yaml'))
cfg.model.NAME = 'Point_CAE_DGCNN_FCOnly'
B, N = 32, 1024
x = torch.from_numpy(shapenet_like_clouds(2 * B, N, seed=700)).to(dev)
for name, cls in (('round-3 host code', old.Point_CAE_DGCNN_FCOnly), ('dgcnn.hip', New)):
    torch.manual_seed(0)
    model = FlatDataParallel(cls(cfg.model).to(dev), broadcast=False, process_group=None)
    model.world_size = 1
    opt, _ = builder.build_opti_sche(model, cfg)
    model.train(); model.zero_grad()
    step = GraphedStaticStep(model, opt, lambda a, b: a + b, B, N)
    for i in range(4):
        out = step(x[:B], x[B:])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        out = step(x[:B], x[B:])
    torch.cuda.synchronize()
    print('%-18s %.3f ms/step  loss %.6f' % (name, (time.perf_counter() - t0) * 100, float(out[0])))
