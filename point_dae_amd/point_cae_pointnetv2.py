"""PointNet++ denoising auto-encoder (BASELINE configs 1-2), MI355X host side.

Reference: models/PointCAE_pointnetv2.py:61-173 (`Point_CAE_PointNetv2`) with the
encoder of models/pointnetv2_util.py:319-346; the set-abstraction module is the
third-party pointnet2_ops `PointnetSAModule`, whose vendored twin
(extensions/pointnet2/pointnet2_modules.py:31-72,124-158, pytorch_utils.py)
gives the parameter names kept here (`...mlps.0.layer{i}.conv.weight`,
`...layer{i}.bn.bn.weight`).

    model(corrupted_pts, pts) -> (loss_coarse, loss_fine)

Data path on MI355X: FPS+centre gather, ball query and the Chamfer losses are the
gfx950 kernels; activations are rows (points) x channels, so grouping is a row
gather and max-pool a reduction over consecutive rows.  folding2's first layer
is not run on the materialised (B,16384,1029) tensor (67 MB per cloud in the
reference, :157-163): its weight is split into the grid (2), coarse-point (3) and
global-feature (1024) column blocks, which are multiplied once per grid cell,
once per coarse point and once per cloud and added by broadcasting -- the same
sum, 25.9 -> 8.7 GFLOP per cloud.
"""
import itertools
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, nn_ops, sa_mlp
from .chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
from .corrupt_util_tensor import corrupt_in_forward
from .pointnet2_utils import ball_query, furthest_point_sample_with_centres
from .registry import MODELS


class _ConvBN(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module('conv', nn.Conv2d(cin, cout, kernel_size=(1, 1), bias=False))
        bn = nn.Sequential()
        bn.add_module('bn', nn.BatchNorm2d(cout))
        self.add_module('bn', bn)
        self.add_module('activation', nn.ReLU(inplace=True))
        nn.init.kaiming_normal_(self.conv.weight)

    def rows(self, x, pad_at=None):
        """(rows, cin) -> relu(bn(conv)) (rows, cout); BatchNorm statistics over the rows.
        pad_at: x carries a zero column at that index (the 3 xyz columns padded to 4 so that the
        row GEMM reduces over a multiple of 4); the weight gets the matching zero column."""
        bn = self.bn.bn
        w = self.conv.weight.reshape(self.conv.weight.shape[0], -1)
        if pad_at is not None:
            w = torch.cat([w[:, :pad_at], w.new_zeros(w.shape[0], 1), w[:, pad_at:]], dim=1)
        y = nn_ops.linear_any(x, w)
        if isinstance(bn, nn.SyncBatchNorm):
            return F.relu(bn(y))                       # --sync_bn: the module owns the cross-replica statistics
        if self.training:
            bn.num_batches_tracked += 1
        y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, self.training,
                         bn.momentum, bn.eps)
        return F.relu(y)


class SharedMLP(nn.Sequential):
    def __init__(self, spec):
        super().__init__()
        for i in range(len(spec) - 1):
            self.add_module('layer{}'.format(i), _ConvBN(spec[i], spec[i + 1]))


class _GroupRows(torch.autograd.Function):
    """QueryAndGroup in row layout (pointnet2_utils.py:345-361) as one kernel each way (csrc/ball_group.hip
    sa_group_rows*): rows [xyz - centre | 0 | features]; gradient to the features only (coordinates are inputs)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, idx, features):
        B, N, _ = xyz.shape
        _, npoint, ns = idx.shape
        C = 0 if features is None else features.shape[1]
        out = torch.empty((B * npoint * ns, 4 + C), device=xyz.device, dtype=torch.float32)
        xyz, new_xyz, idx = xyz.contiguous(), new_xyz.contiguous(), idx.contiguous()
        feats = features.contiguous() if features is not None else None
        _lib.call('pdae_sa_group_rows', xyz, B, N, npoint, ns, C, _lib.ptr(xyz), _lib.ptr(new_xyz), _lib.ptr(idx),
                  _lib.ptr(feats), _lib.ptr(out))
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, npoint, ns, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        B, N, npoint, ns, C = ctx.dims
        if C == 0 or not ctx.needs_input_grad[3]:
            return None, None, None, None
        dout = dout.contiguous()
        dfeat = torch.empty((B * N, C), device=dout.device, dtype=torch.float32)
        _lib.call('pdae_sa_group_rows_grad', dout, B, N, npoint, ns, C, _lib.ptr(idx), _lib.ptr(dout), _lib.ptr(dfeat))
        return None, None, None, dfeat


GROUP_FUSED = os.environ.get('PDAE_SA_GROUP', 'fused') != 'torch'      # (A/B: the index_select / cat form)


class PointnetSAModule(nn.Module):
    """FPS -> ball query -> group (centre-subtracted xyz || features) -> shared MLP -> max."""

    def __init__(self, mlp, npoint=None, radius=None, nsample=None, use_xyz=True):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        spec = list(mlp)
        if use_xyz:
            spec[0] += 3
        self.mlps = nn.ModuleList([SharedMLP(spec)])

    def forward(self, xyz, features=None):
        """xyz (B,N,3); features rows (B*N, C) or None -> new_xyz (B,np,3), rows (B*np, C')."""
        B, N, _ = xyz.shape
        if self.npoint is not None:
            with torch.no_grad():
                _, new_xyz = furthest_point_sample_with_centres(xyz, self.npoint)
                idx = ball_query(self.radius, self.nsample, xyz, new_xyz)          # (B,np,ns) i32
            C = 0 if features is None else features.shape[1]
            if (GROUP_FUSED and C % 4 == 0 and C <= 1024 and N <= 4096 and
                    4 * (2 * N + 1 + self.npoint * self.nsample) <= 150 * 1024):     # (a cloud's sort lives in LDS)
                g = _GroupRows.apply(xyz, new_xyz, idx, features)      # xyz - centre | 0 | features: K a multiple of 4
            else:
                flat = (idx.long() + torch.arange(B, device=xyz.device).view(B, 1, 1) * N).reshape(-1)
                g = xyz.reshape(B * N, 3).index_select(0, flat).reshape(B, self.npoint, self.nsample, 3)
                g = (g - new_xyz.unsqueeze(2)).reshape(-1, 3)
                zero = g.new_zeros(g.shape[0], 1)
                g = torch.cat([g, zero] + ([features.index_select(0, flat)] if features is not None else []), dim=1)
            groups, per = B * self.npoint, self.nsample
        else:
            new_xyz = None
            g = xyz.reshape(B * N, 3)
            g = torch.cat([g, g.new_zeros(g.shape[0], 1)] + ([features] if features is not None else []), dim=1)
            groups, per = B, N
        layers = list(self.mlps[0])
        if self.training and per <= 256 and g.shape[0] % 32 == 0 and not any(
                isinstance(layer.bn.bn, nn.SyncBatchNorm) for layer in layers):
            return new_xyz, sa_mlp.shared_mlp_max(g, layers, per, pad_at=3)      # fused (sa_mlp.py)
        for i, layer in enumerate(layers):
            g = layer.rows(g, pad_at=3 if i == 0 else None)
        return new_xyz, g.reshape(groups, per, -1).max(dim=1)[0]


class PointNetv2_encoder(nn.Module):
    def __init__(self, num_channel=3):
        super().__init__()
        self.sa1 = PointnetSAModule(npoint=512, radius=0.2, nsample=32, mlp=[0, 64, 64, 128])
        self.sa2 = PointnetSAModule(npoint=128, radius=0.4, nsample=64, mlp=[128, 128, 128, 256])
        self.sa3 = PointnetSAModule(mlp=[256, 256, 512, 1024])

    def forward(self, xyz):
        l1_xyz, l1 = self.sa1(xyz, None)
        l2_xyz, l2 = self.sa2(l1_xyz, l1)
        _, l3 = self.sa3(l2_xyz, l2)
        return l3                                                  # (B, 1024)


@MODELS.register_module()
class Point_CAE_PointNetv2(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.corrupt_type = config.corrupt_type
        self.grid_size, self.grid_scale, self.num_coarse = 4, 0.05, 1024
        self.num_fine = self.grid_size ** 2 * self.num_coarse
        self.pointnetv2_encoder = PointNetv2_encoder()
        self.folding1 = nn.Sequential(nn.Linear(1024, 1024), nn.ReLU(), nn.Linear(1024, 1024), nn.ReLU(),
                                      nn.Linear(1024, self.num_coarse * 3))
        self.folding2 = nn.Sequential(nn.Conv1d(1024 + 2 + 3, 512, 1), nn.ReLU(), nn.Conv1d(512, 512, 1),
                                      nn.ReLU(), nn.Conv1d(512, 3, 1))
        x = np.linspace(-self.grid_scale, self.grid_scale, self.grid_size)
        grid = torch.tensor(np.array(list(itertools.product(x, x)))).float()   # (16, 2), build_grid :94-99
        self.register_buffer('grid', grid, persistent=False)
        self.loss = config.loss
        self.build_loss_func(self.loss)

    @property
    def draws_in_forward(self):
        """True when forward() draws random numbers on the host (the in-forward corruptions): such a step
        must not be captured into a hipGraph -- the draw and its H2D copy would be replayed frozen."""
        return any(item in ('dropout_global', 'dropout_patch_pointmae') for item in self.corrupt_type)

    def build_loss_func(self, loss_type):
        if loss_type == 'cdl1':
            self.loss_func = ChamferDistanceL1()
        elif loss_type == 'cdl2':
            self.loss_func = ChamferDistanceL2()
        else:
            raise NotImplementedError(loss_type)

    def forward(self, corrupted_pts, pts, vis=False, capture=None, **kwargs):
        nn_ops.begin_step(pts.device)
        corrupted_pts = corrupted_pts[:, :, :3].contiguous()
        pts = pts[:, :, :3].contiguous()
        # the CUDA-side dropouts of the reference's forward (:143-149); everything else came from the loader
        corrupted_pts = corrupt_in_forward(corrupted_pts, self.corrupt_type, ('dropout_patch_pointmae', 'dropout_global'))
        B = pts.shape[0]
        feature = self.pointnetv2_encoder(corrupted_pts)                       # (B, 1024)
        f1 = self.folding1
        coarse = nn_ops.mlp_chain(feature, [f1[0], f1[2], f1[4]])
        coarse = coarse.view(B, self.num_coarse, 3)
        # folding2[0] on [grid(2) | coarse point(3) | global feature(1024)], split by column block
        w = self.folding2[0].weight.squeeze(-1)                                # (512, 1029)
        g2 = self.grid_size ** 2
        wg, wc, wf = nn_ops.split_weight_cols(w, [(0, 2), (2, 5), (5, w.shape[1])])          # (narrow blocks padded to 4 columns)
        a = nn_ops.linear_any(feature, wf, self.folding2[0].bias)                           # (B, 512)  once per cloud
        p = nn_ops.linear_any(nn_ops.pad2d(coarse.reshape(-1, 3), 0, 1), wc).reshape(B, self.num_coarse, 1, -1)   # once per coarse point
        gd = nn_ops.linear_any(nn_ops.pad2d(self.grid, 0, 2), wg)               # (16, 512)     once per grid cell
        off = nn_ops.fold_mlp(a, p.reshape(B * self.num_coarse, -1), gd, self.folding2[2], self.folding2[4],
                              B, self.num_coarse, g2)
        fine = off.reshape(B, self.num_coarse, g2, 3) + coarse.unsqueeze(2)
        fine = fine.reshape(B, self.num_fine, 3)
        if capture is not None:
            capture.update(feature=feature, coarse=coarse, fine=fine)
        return self.loss_func(coarse, pts), self.loss_func(fine, pts)
