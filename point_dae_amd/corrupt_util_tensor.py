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


# ---- whole-cloud dropouts that the non-Transformer models apply inside forward() -------------------------------
# (datasets/corrupt_util.py:572-588, :900-924; dispatched by models/PointCAE_pointnetv2.py:143-149 and
# models/PointCAE_DGCNN.py:198-221).  The random draws are the reference's host calls in its order (python `random`,
# the CPU torch generator), so a run seeded like the reference keeps the same points; FPS / kNN run on the gfx950
# kernels.  A model that applies one of these must be stepped eagerly (`draws_in_forward`): a captured graph would
# replay one frozen draw -- and the surviving point count changes from step to step.
IN_FORWARD = ('dropout_patch_pointmae', 'dropout_global', 'dropout_global_p1', 'dropout_global_p3', 'dropout_global_p5',
              'dropout_global_p7', 'dropout_global_p9', 'random_dropout')


def dropout_global_random(pointcloud, drop_rate=0.5):
    """A random (1 - drop_rate) share of every cloud: one CPU torch.rand per batch, argsort, take (:572-588)."""
    num_samples, num_points = pointcloud.size(0), pointcloud.size(1)
    inx = torch.rand(num_samples, num_points, 1).argsort(1).to(pointcloud.device)
    pointcloud = torch.take_along_dim(pointcloud, inx, dim=1)
    return pointcloud[:, :int(num_points * (1 - drop_rate)), :].contiguous()


def dropout_patch_random(pc_tensor, level=None):
    """Point-MAE style patch dropout (:900-924): 64 FPS centres, their 32 nearest points each, a random subset of the
    64 patches kept (each with probability 1 - prob, prob = level / 10 + 0.5, level ~ U[0, 4); at least one) ->
    (B, kept * 32, 3) absolute coordinates (points shared by kept patches appear more than once, as in the reference)."""
    from .knn_cuda import knn
    from .pointnet2_utils import furthest_point_sample_with_centres
    if level is None:
        level = random.random() * 4
    prob = level / 10.0 + 0.5
    batch_size, num_points, _ = pc_tensor.shape
    xyz = pc_tensor[:, :, :3].contiguous()
    _, center = furthest_point_sample_with_centres(xyz, 64)
    _, idx = knn(xyz, center, 32)                                            # (B, 64, 32)
    idx = (idx + torch.arange(0, batch_size, device=pc_tensor.device).view(-1, 1, 1) * num_points).view(-1)
    neighborhood = xyz.view(batch_size * num_points, -1)[idx, :].view(batch_size, 64, 32, 3)
    group_mask = torch.rand(64) > prob
    if group_mask.sum().item() == 0:
        group_mask[0] = True
    neighborhood = neighborhood[:, group_mask.to(pc_tensor.device)]
    return neighborhood.reshape(batch_size, -1, 3).contiguous()


def corrupt_in_forward(corrupted_pts, corrupt_type, items=IN_FORWARD):
    """The forward-side dispatch of the two models: every item of `corrupt_type` that is one of `items` acts on the
    (B, N, 3) cloud in list order; everything else was applied by the data loader."""
    for item in corrupt_type:
        if item not in items:
            continue
        if item == 'dropout_patch_pointmae':
            corrupted_pts = dropout_patch_random(corrupted_pts)
        elif item == 'dropout_global':
            corrupted_pts = dropout_global_random(corrupted_pts)
        elif item.startswith('dropout_global_p'):
            corrupted_pts = dropout_global_random(corrupted_pts, drop_rate=int(item[-1]) / 10.0)
        elif item == 'random_dropout':
            if random.random() > 0.5:
                corrupted_pts = dropout_patch_random(corrupted_pts)
            else:
                corrupted_pts = dropout_global_random(corrupted_pts)
    return corrupted_pts
