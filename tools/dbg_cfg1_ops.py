"""Check every row-GEMM / wgrad call of a cfg1 step against fp64 on the same operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from golden_util import load_fixture, fill_state
from point_dae_amd import nn_ops
from point_dae_amd.config import cfg_from_yaml_file
from point_dae_amd.point_cae_pointnetv2 import Point_CAE_PointNetv2
from point_dae_amd.graph_step import use_created_stream
use_created_stream()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fx = load_fixture('pointnetv2_cfg1_b2.npz')
cfg = cfg_from_yaml_file(os.path.join(root, 'cfgs', 'pretrain_PointCAE_clean.yaml')).model
og, ow = nn_ops.rows_gemm, nn_ops.rows_wgrad


def rel(a, ref):
    return ((a.double() - ref).abs().max() / (ref.abs().max() + 1e-30)).item()


def gemm(x, w, w_kn=False, bias=None, epi=0, z=None, may_split=False):
    y = og(x, w, w_kn, bias, epi, z, may_split)
    ref = x.double() @ (w.double() if w_kn else w.double().t())
    if bias is not None:
        ref = ref + bias.double()
    if epi == 1:
        ref = ref.relu()
    e = rel(y if y.dim() == 2 else y.sum(0), ref)
    print('gemm  M%6d K%5d N%5d kn%d epi%d contiguous x%d w%d  err %.1e %s' % (
        x.shape[0], x.shape[1], ref.shape[1], w_kn, epi, x.is_contiguous(), w.is_contiguous(), e, '<<<<' if e > 1e-5 else ''))
    return y


def wgrad(dys, xs, wb, *a, **k):
    dws, dbs = ow(dys, xs, wb, *a, **k)
    for dy, x, dw in zip(dys, xs, dws):
        e = rel(dw, dy.double().t() @ x.double())
        print('wgrad M%6d N%5d K%5d contiguous dy%d x%d  err %.1e %s' % (
            dy.shape[0], dy.shape[1], x.shape[1], dy.is_contiguous(), x.is_contiguous(), e, '<<<<' if e > 1e-4 else ''))
    return dws, dbs


nn_ops.rows_gemm, nn_ops.rows_wgrad = gemm, wgrad
model = fill_state(Point_CAE_PointNetv2(cfg), int(fx['seed'])).cuda().train()
lc, lf = model(torch.from_numpy(fx['corrupted']).cuda(), torch.from_numpy(fx['clean']).cuda())
(lc + 0.5 * lf).backward()
