"""In-forward (tensor) corruptions of the pretraining step.

Host side of datasets/corrupt_util_tensor.py: `corrupt_data(neighborhood,
center, type)` (:706-727) with 'affine_r3' = 1-3 of {translate, scale_nonorm,
rotate, reflection, shear} at level 4.  The random maps are drawn on the host
with exactly the reference's RNG calls, in its order (python `random`, the CPU
torch generator, numpy's global generator), so a run seeded like the reference
(utils/misc.py:42-66) draws the same maps; they are then applied by one HIP
kernel (csrc/corrupt.hip) instead of a chain of broadcast multiplies, batched
matmuls and small H2D copies.

Bug-compatible on purpose: 'translate' multiplies, the z
reflection lands on the x axis.
"""
import math
import random

import numpy as np
import torch

from . import _lib

affine_corruptions = ['translate', 'scale_nonorm', 'rotate', 'reflection', 'shear']
_LEVEL = 4


def _rot(axis, ang):
    B = ang.shape[0]
    R = torch.eye(3).repeat(B, 1, 1)
    c, s = torch.cos(ang), torch.sin(ang)
    i, j = {'x': (1, 2), 'y': (2, 0), 'z': (0, 1)}[axis]
    R[:, i, i], R[:, i, j], R[:, j, i], R[:, j, j] = c, -s, s, c
    return R


def _draw(name, B):
    """-> (10,)-wide rows [kind, params...] for each of the B samples."""
    row = torch.zeros(B, 10)
    if name in ('scale_nonorm', 'translate'):
        if name == 'scale_nonorm':
            s = [1.6, 1.7, 1.8, 1.9, 2.0][_LEVEL]
            v = torch.FloatTensor(B, 1, 1, 3).uniform_(1. / s, s)
        else:
            s = [0.1, 0.2, 0.3, 0.4, 0.5][_LEVEL]
            v = torch.FloatTensor(B, 1, 1, 3).uniform_(-s, s)
        row[:, 1:4] = v.reshape(B, 3)
        return row
    if name == 'rotate':
        clip = math.pi / 5 * (_LEVEL + 1)
        ang = torch.FloatTensor(B, 3).uniform_(-clip, clip)
        R = torch.matmul(_rot('z', ang[:, 2]), torch.matmul(_rot('y', ang[:, 1]), _rot('x', ang[:, 0])))
    elif name == 'reflection':
        r = torch.from_numpy(np.random.choice(np.array([1, -1]), size=(B, 3))).float()
        Rx, Ry, Rz = (torch.eye(3).repeat(B, 1, 1) for _ in range(3))
        Rx[:, 0, 0], Ry[:, 1, 1], Rz[:, 0, 0] = r[:, 0], r[:, 1], r[:, 2]
        R = torch.matmul(Rz, torch.matmul(Ry, Rx))
    elif name == 'shear':
        clip = (_LEVEL + 1) * 0.1
        sh = torch.from_numpy(np.random.uniform(-clip, clip, size=(B, 6)))
        R = torch.eye(3).repeat(B, 1, 1)
        for col, (i, j) in enumerate(((0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1))):
            R[:, i, j] = sh[:, col]
    else:
        raise KeyError(name)
    row[:, 0] = 1.0
    row[:, 1:] = R.reshape(B, 9)
    return row


def draw_corruption(corrupt_type, batch_size):
    """The random part of corrupt_data: -> steps (nsteps, B, 10) f32 on the host
    (nsteps may be 0)."""
    steps = []
    for item in corrupt_type:
        if item in ('clean', 'Drop-Patch'):
            continue
        if item == 'affine_r3':
            number = random.choice([1, 2, 3])
            for name in random.sample(affine_corruptions, number):
                steps.append(_draw(name, batch_size))
        else:
            # the reference's generic branch reads an undefined `level`
            # (corrupt_util_tensor.py:722-726) and raises NameError
            raise NotImplementedError(f'in-forward corruption {item!r}')
    if not steps:
        return torch.zeros(0, batch_size, 10)
    return torch.stack(steps, 0)


def corrupt_patches(neighborhood, center, steps):
    """neighborhood (B,G,k,3) centre-subtracted, center (B,G,3), steps from
    draw_corruption -> (gt_neighborhood, transformed_neighborhood,
    transformed_center) as PointCAE_transformer.forward :680-684 leaves them."""
    _lib.require(neighborhood, 'neighborhood', torch.float32, 4)
    _lib.require(center, 'center', torch.float32, 3)
    B, G, K, _ = neighborhood.shape
    steps = steps.to(device=neighborhood.device, dtype=torch.float32).contiguous()
    gt = torch.empty_like(neighborhood)
    t_nb = torch.empty_like(neighborhood)
    t_c = torch.empty_like(center)
    _lib.call('pdae_patch_affine', neighborhood, B, G, K, steps.shape[0], _lib.ptr(neighborhood),
              _lib.ptr(center), _lib.ptr(steps) if steps.numel() else None, _lib.ptr(gt),
              _lib.ptr(t_nb), _lib.ptr(t_c))
    return gt, t_nb, t_c


def corrupt_data(neighborhood, center, type=['clean']):
    """Reference signature (corrupt_util_tensor.py:706): absolute-coordinate
    patches in, transformed patches and centres out."""
    steps = draw_corruption(type, neighborhood.shape[0])
    rel = (neighborhood - center.unsqueeze(2)).contiguous()
    _, t_nb, t_c = corrupt_patches(rel, center.contiguous(), steps)
    return t_nb + t_c.unsqueeze(2), t_c
