"""Operator API of the reference's pointnet2 extension, on the gfx950 kernels.

Same names, argument order, dtypes and gradients as
extensions/pointnet2/pointnet2_utils.py:49-289 (and the third-party
pointnet2_ops.pointnet2_utils twin the models actually import,
utils/misc.py:10), so reference model code runs against it unchanged.
"""
import torch
from torch.autograd import Function

from . import _lib


class FurthestPointSampling(Function):
    """pointnet2_utils.py:49-78: xyz (B,N,3) f32 -> (B,npoint) i32, no grad."""

    @staticmethod
    def forward(ctx, xyz, npoint):
        _lib.require(xyz, "xyz", torch.float32, 3)
        B, N, C = xyz.shape
        if C != 3:
            raise RuntimeError("xyz must be (B, N, 3)")
        idx = torch.empty((B, int(npoint)), dtype=torch.int32, device=xyz.device)
        _lib.call("pdae_furthest_point_sampling", xyz, B, N, int(npoint), _lib.ptr(xyz),
                  _lib.ptr(idx), None)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad=None):
        return None, None


furthest_point_sample = FurthestPointSampling.apply


def furthest_point_sample_with_centres(xyz, npoint):
    """FPS + the gather of utils/misc.py:18-19 in one launch:
    -> idx (B,npoint) i32, centres (B,npoint,3) f32 (no grad to xyz, like the
    reference where the pretraining inputs carry none)."""
    _lib.require(xyz, "xyz", torch.float32, 3)
    B, N, _ = xyz.shape
    idx = torch.empty((B, int(npoint)), dtype=torch.int32, device=xyz.device)
    ctr = torch.empty((B, int(npoint), 3), dtype=torch.float32, device=xyz.device)
    _lib.call("pdae_furthest_point_sampling", xyz, B, N, int(npoint), _lib.ptr(xyz),
              _lib.ptr(idx), _lib.ptr(ctr))
    return idx, ctr


class GatherOperation(Function):
    """pointnet2_utils.py:81-115: features (B,C,N), idx (B,m) i32 -> (B,C,m)."""

    @staticmethod
    def forward(ctx, features, idx):
        _lib.require(features, "features", torch.float32, 3)
        _lib.require(idx, "idx", torch.int32, 2)
        B, C, N = features.shape
        m = idx.shape[1]
        out = torch.empty((B, C, m), dtype=torch.float32, device=features.device)
        _lib.call("pdae_gather_points", features, B, C, N, m, _lib.ptr(features), _lib.ptr(idx),
                  _lib.ptr(out))
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, m = grad_out.shape
        grad = torch.empty((B, C, ctx.N), dtype=torch.float32, device=grad_out.device)
        _lib.call("pdae_gather_points_grad", grad_out, B, C, ctx.N, m, _lib.ptr(grad_out),
                  _lib.ptr(idx), _lib.ptr(grad))
        return grad, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    """pointnet2_utils.py:118-147: unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) = sqrt of the
    squared distances, idx (B,n,3) i32), no grad."""

    @staticmethod
    def forward(ctx, unknown, known):
        _lib.require(unknown, "unknown", torch.float32, 3)
        _lib.require(known, "known", torch.float32, 3)
        B, n, _ = unknown.shape
        m = known.shape[1]
        dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknown.device)
        idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknown.device)
        _lib.call("pdae_three_nn", unknown, B, n, m, _lib.ptr(unknown), _lib.ptr(known), _lib.ptr(dist2),
                  _lib.ptr(idx))
        ctx.mark_non_differentiable(idx)
        return torch.sqrt(dist2), idx

    @staticmethod
    def backward(ctx, a=None, b=None):
        return None, None


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    """pointnet2_utils.py:150-204: features (B,c,m), idx (B,n,3) i32, weight (B,n,3) -> (B,c,n);
    grad -> features."""

    @staticmethod
    def forward(ctx, features, idx, weight):
        _lib.require(features, "features", torch.float32, 3)
        _lib.require(idx, "idx", torch.int32, 3)
        _lib.require(weight, "weight", torch.float32, 3)
        B, c, m = features.shape
        n = idx.shape[1]
        out = torch.empty((B, c, n), dtype=torch.float32, device=features.device)
        _lib.call("pdae_three_interpolate", features, B, c, m, n, _lib.ptr(features), _lib.ptr(idx),
                  _lib.ptr(weight), _lib.ptr(out))
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, c, n = grad_out.shape
        grad = torch.empty((B, c, ctx.m), dtype=torch.float32, device=grad_out.device)
        _lib.call("pdae_three_interpolate_grad", grad_out, B, c, n, ctx.m, _lib.ptr(grad_out), _lib.ptr(idx),
                  _lib.ptr(weight), _lib.ptr(grad))
        return grad, None, None


three_interpolate = ThreeInterpolate.apply


class BallQuery(Function):
    """pointnet2_utils.py:258-289: (radius, nsample, xyz (B,N,3), new_xyz (B,m,3))
    -> idx (B,m,nsample) i32, no grad."""

    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        _lib.require(xyz, "xyz", torch.float32, 3)
        _lib.require(new_xyz, "new_xyz", torch.float32, 3)
        B, N, _ = xyz.shape
        m = new_xyz.shape[1]
        idx = torch.empty((B, m, int(nsample)), dtype=torch.int32, device=xyz.device)
        _lib.call("pdae_ball_query", xyz, B, N, m, float(radius), int(nsample),
                  _lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(idx))
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad=None):
        return None, None, None, None


ball_query = BallQuery.apply


class GroupingOperation(Function):
    """pointnet2_utils.py:207-255: features (B,C,N), idx (B,np,ns) i32 ->
    (B,C,np,ns); grad flows to features."""

    @staticmethod
    def forward(ctx, features, idx):
        _lib.require(features, "features", torch.float32, 3)
        _lib.require(idx, "idx", torch.int32, 3)
        B, C, N = features.shape
        _, npnt, ns = idx.shape
        out = torch.empty((B, C, npnt, ns), dtype=torch.float32, device=features.device)
        _lib.call("pdae_group_points", features, B, C, N, npnt, ns, _lib.ptr(features),
                  _lib.ptr(idx), _lib.ptr(out))
        ctx.save_for_backward(idx)
        ctx.N = N
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, npnt, ns = grad_out.shape
        grad = torch.empty((B, C, ctx.N), dtype=torch.float32, device=grad_out.device)
        _lib.call("pdae_group_points_grad", grad_out, B, C, ctx.N, npnt, ns, _lib.ptr(grad_out),
                  _lib.ptr(idx), _lib.ptr(grad))
        return grad, None


grouping_operation = GroupingOperation.apply


class QueryAndGroup(torch.nn.Module):
    """pointnet2_utils.py:292-375 (the options the pretraining path uses):
    ball query + group xyz (centre-subtracted) [+ features]."""

    def __init__(self, radius, nsample, use_xyz=True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz, new_xyz, features=None):
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            if not self.use_xyz:
                raise RuntimeError("Cannot have not features and not use xyz as a feature!")
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


class GroupAll(torch.nn.Module):
    """pointnet2_utils.py:378-424: one group holding every point."""

    def __init__(self, use_xyz=True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
