"""Approximate earth mover's distance on the gfx950 kernels.

Mirrors extensions/emd/emd.py:5-49 (EarthMoverDistanceFunction,
earth_mover_distance) and the pybind entries emd_cuda.approxmatch_forward /
matchcost_forward / matchcost_backward (extensions/emd/cuda/emd.cpp:23-27).
"""
import torch

from . import _lib


def approxmatch_forward(xyz1, xyz2):
    """(B,n,3),(B,m,3) -> match (B,m,n)."""
    _lib.require(xyz1, "xyz1", torch.float32, 3)
    _lib.require(xyz2, "xyz2", torch.float32, 3)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    match = torch.empty((B, m, n), dtype=torch.float32, device=xyz1.device)
    # the reference's scratch (emd.cpp:12: temp (b, (n + m) * 2)): the remain / ratio vectors of the per-phase launches
    # that clouds of more than 64 points take; the one-wave-per-pair kernel of small clouds keeps them on chip
    temp = torch.empty((B, 2 * (n + m)), dtype=torch.float32, device=xyz1.device) if max(n, m) > 64 else None
    _lib.call("pdae_emd_approxmatch", xyz1, B, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2),
              _lib.ptr(match), _lib.ptr(temp))
    return match


def matchcost_forward(xyz1, xyz2, match):
    _lib.require(xyz1, "xyz1", torch.float32, 3)
    _lib.require(xyz2, "xyz2", torch.float32, 3)
    _lib.require(match, "match", torch.float32, 3)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    cost = torch.empty((B,), dtype=torch.float32, device=xyz1.device)
    _lib.call("pdae_emd_matchcost", xyz1, B, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2),
              _lib.ptr(match), _lib.ptr(cost))
    return cost


def matchcost_backward(grad_cost, xyz1, xyz2, match):
    _lib.require(grad_cost, "grad_cost", torch.float32, 1)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    g1 = torch.empty_like(xyz1)
    g2 = torch.empty_like(xyz2)
    _lib.call("pdae_emd_matchcost_grad", xyz1, B, n, m, _lib.ptr(grad_cost), _lib.ptr(xyz1),
              _lib.ptr(xyz2), _lib.ptr(match), _lib.ptr(g1), _lib.ptr(g2))
    return g1, g2


class EarthMoverDistanceFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        match = approxmatch_forward(xyz1, xyz2)
        cost = matchcost_forward(xyz1, xyz2, match)
        ctx.save_for_backward(xyz1, xyz2, match)
        return cost

    @staticmethod
    def backward(ctx, grad_cost):
        xyz1, xyz2, match = ctx.saved_tensors
        return matchcost_backward(grad_cost.contiguous(), xyz1, xyz2, match)


class earth_mover_distance(torch.nn.Module):
    """mean over the batch of cost / n1 (emd.py:29-49)."""

    def forward(self, xyz1, xyz2, transpose=False):
        cost = EarthMoverDistanceFunction.apply(xyz1, xyz2)
        return (cost / xyz1.size(1)).mean()
