"""Chamfer distance operator + loss modules on the gfx950 kernels.

Mirrors extensions/chamfer_dist/__init__.py: ChamferFunction :14-26,
ChamferDistanceL2 :29-44, ChamferDistanceL2_split :380-395,
ChamferDistanceL1 :397-417.  `forward` / `backward` below are the pybind
entries chamfer.forward / chamfer.backward (chamfer_cuda.cpp:36-39).
"""
import torch

from . import _lib


def forward(xyz1, xyz2):
    """chamfer.forward: (B,n,3),(B,m,3) -> [dist1 (B,n), dist2 (B,m), idx1, idx2 (i32)]."""
    _lib.require(xyz1, "xyz1", torch.float32, 3)
    _lib.require(xyz2, "xyz2", torch.float32, 3)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    if xyz2.shape[0] != B or xyz1.shape[2] != 3 or xyz2.shape[2] != 3:
        raise RuntimeError("chamfer expects xyz1 (B,n,3) and xyz2 (B,m,3)")
    dev = xyz1.device
    dist1 = torch.empty((B, n), dtype=torch.float32, device=dev)
    dist2 = torch.empty((B, m), dtype=torch.float32, device=dev)
    idx1 = torch.empty((B, n), dtype=torch.int32, device=dev)
    idx2 = torch.empty((B, m), dtype=torch.int32, device=dev)
    _lib.call("pdae_chamfer_forward", xyz1, B, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2),
              _lib.ptr(dist1), _lib.ptr(dist2), _lib.ptr(idx1), _lib.ptr(idx2))
    return [dist1, dist2, idx1, idx2]


def backward(xyz1, xyz2, idx1, idx2, grad_dist1, grad_dist2):
    """chamfer.backward -> [grad_xyz1, grad_xyz2]."""
    for t, nm in ((xyz1, "xyz1"), (xyz2, "xyz2"), (grad_dist1, "grad_dist1"),
                  (grad_dist2, "grad_dist2")):
        _lib.require(t, nm, torch.float32)
    _lib.require(idx1, "idx1", torch.int32, 2)
    _lib.require(idx2, "idx2", torch.int32, 2)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    _lib.call("pdae_chamfer_backward", xyz1, B, n, _lib.ptr(xyz1), m, _lib.ptr(xyz2),
              _lib.ptr(idx1), _lib.ptr(idx2), _lib.ptr(grad_dist1), _lib.ptr(grad_dist2),
              _lib.ptr(g1), _lib.ptr(g2))
    return [g1, g2]


class ChamferFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        dist1, dist2, idx1, idx2 = forward(xyz1, xyz2)
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        ctx.mark_non_differentiable(idx1, idx2)
        ctx.set_materialize_grads(False)      # no zero-filled gradients for the two index outputs (2 fills per step)
        return dist1, dist2, idx1, idx2

    @staticmethod
    def backward(ctx, grad_dist1, grad_dist2, grad_idx1=None, grad_idx2=None):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        if grad_dist1 is None and grad_dist2 is None:
            return None, None
        if grad_dist1 is None:
            grad_dist1 = torch.zeros(idx1.shape, dtype=xyz1.dtype, device=xyz1.device)
        if grad_dist2 is None:
            grad_dist2 = torch.zeros(idx2.shape, dtype=xyz2.dtype, device=xyz2.device)
        g1, g2 = backward(xyz1, xyz2, idx1, idx2, grad_dist1.contiguous(),
                          grad_dist2.contiguous())
        return g1, g2


def _drop_zero_points(xyz1, xyz2):
    # __init__.py:38-42: only for batch size 1
    nz1 = torch.sum(xyz1, dim=2).ne(0)
    nz2 = torch.sum(xyz2, dim=2).ne(0)
    return xyz1[nz1].unsqueeze(dim=0), xyz2[nz2].unsqueeze(dim=0)


class _ChamferL2Loss(torch.autograd.Function):
    """mean(dist1) + mean(dist2) as ONE node: the two Chamfer launches + one reduction launch forward, the two
    gradient launches backward (the gradient of a mean is a constant: no expand / divide / fill launches)."""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1, xyz2 = xyz1.contiguous(), xyz2.contiguous()
        dist1, dist2, idx1, idx2 = forward(xyz1, xyz2)
        ws = torch.empty(257, dtype=torch.float32, device=xyz1.device)        # 256 partials + the result
        out = ws[256].reshape(())
        _lib.call("pdae_mean_sum2", xyz1, dist1.numel(), _lib.ptr(dist1), dist2.numel(), _lib.ptr(dist2), _lib.ptr(ws),
                  _lib.ptr(out))
        ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        return out

    @staticmethod
    def backward(ctx, g):
        xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        B, n, _ = xyz1.shape
        g1, g2 = torch.empty_like(xyz1), torch.empty_like(xyz2)
        _lib.call("pdae_chamfer_backward_mean", xyz1, B, n, _lib.ptr(xyz1), xyz2.shape[1], _lib.ptr(xyz2), _lib.ptr(idx1),
                  _lib.ptr(idx2), _lib.ptr(g.contiguous()), _lib.ptr(g1), _lib.ptr(g2))
        return g1, g2


class ChamferDistanceL2(torch.nn.Module):
    """mean(dist1) + mean(dist2)"""

    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        if xyz1.size(0) == 1 and self.ignore_zeros:
            xyz1, xyz2 = _drop_zero_points(xyz1, xyz2)
        _lib.require(xyz1, "xyz1", torch.float32) if xyz1.is_contiguous() else None
        if not xyz1.is_cuda:
            raise RuntimeError("xyz1 must be a tensor on the GPU (CPU not supported)")
        if xyz1.numel() == 0 or xyz2.numel() == 0:
            dist1, dist2, _, _ = ChamferFunction.apply(xyz1, xyz2)
            return torch.mean(dist1) + torch.mean(dist2)
        return _ChamferL2Loss.apply(xyz1, xyz2)


class ChamferDistanceL2_split(torch.nn.Module):
    """(mean(dist1), mean(dist2))"""

    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        if xyz1.size(0) == 1 and self.ignore_zeros:
            xyz1, xyz2 = _drop_zero_points(xyz1, xyz2)
        dist1, dist2, _, _ = ChamferFunction.apply(xyz1, xyz2)
        return torch.mean(dist1), torch.mean(dist2)


class ChamferDistanceL1(torch.nn.Module):
    """(mean(sqrt(dist1)) + mean(sqrt(dist2))) / 2"""

    def __init__(self, ignore_zeros=False):
        super().__init__()
        self.ignore_zeros = ignore_zeros

    def forward(self, xyz1, xyz2):
        if xyz1.size(0) == 1 and self.ignore_zeros:
            xyz1, xyz2 = _drop_zero_points(xyz1, xyz2)
        dist1, dist2, _, _ = ChamferFunction.apply(xyz1, xyz2)
        return (torch.mean(torch.sqrt(dist1)) + torch.mean(torch.sqrt(dist2))) / 2
