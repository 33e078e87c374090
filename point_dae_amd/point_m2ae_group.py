"""Point-M2AE's hierarchical grouping on the gfx950 kernels.

Reference: models/Point_M2AE_modules.py:219-248 (`Group`: FPS centres, kNN patches, centre-subtracted
neighbourhoods, and the FLAT neighbour indices `idx + b * N` as a third result), models/Point_M2AE.py:245-263
(the three-level pyramid: level 0 groups the cloud, level i > 0 groups the centres of level i - 1; shipped
configuration `num_groups [512, 256, 64]`, `group_sizes [16, 8, 8]` on 2048-point clouds), :85-97 (`rand_mask`),
:107-121 (multi-scale masking by back-propagation) and :132 (token merging).

Per level: one FPS launch (centres gathered in the same launch), one kNN launch (indices, centre-subtracted
patches written by the kernel), one index pass (csrc/hier_group.hip); the reference runs FPS, two transposes, a
gather, a Python loop over the batch inside KNN_CUDA, and six elementwise launches.  No gradients flow through
any of it (FPS / kNN have none; the neighbourhoods are inputs of the token embedders).  There is no CPU path.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .knn_cuda import knn
from .pointnet2_utils import furthest_point_sample_with_centres


class Group(nn.Module):
    """FPS + kNN (Point_M2AE_modules.py:219-248) -> neighborhood (B,G,M,3) centre-subtracted, center (B,G,3),
    idx (B*G*M,) int64 = neighbour index + b * N, flat into the (B*N) rows the level was grouped from."""

    def __init__(self, num_group, group_size):
        super().__init__()
        self.num_group, self.group_size = num_group, group_size

    @torch.no_grad()
    def forward(self, xyz):
        B, N, _ = xyz.shape
        xyz = xyz.contiguous()
        _, center = furthest_point_sample_with_centres(xyz, self.num_group)
        _, idx, neighborhood = knn(xyz, center, self.group_size, with_neighbourhood=True)
        assert idx.size(1) == self.num_group and idx.size(2) == self.group_size
        flat = torch.empty(B * self.num_group * self.group_size, dtype=torch.int64, device=xyz.device)
        _lib.call('pdae_flatten_group_index', xyz, B, N, self.num_group * self.group_size, _lib.ptr(idx), _lib.ptr(flat))
        return neighborhood, center, flat


class HierarchicalGroup(nn.Module):
    """The tokenizer pyramid of Point_M2AE.forward (:245-263): `group_dividers[i]`, level 0 on the points and level
    i > 0 on the centres of level i - 1.  -> (neighborhoods, centers, idxs), finest level first."""

    def __init__(self, num_groups, group_sizes):
        super().__init__()
        assert len(num_groups) == len(group_sizes)
        self.group_dividers = nn.ModuleList(Group(g, k) for g, k in zip(num_groups, group_sizes))

    @torch.no_grad()
    def forward(self, pts):
        pts = pts[:, :, :3].contiguous()
        neighborhoods, centers, idxs = [], [], []
        src = pts
        for divider in self.group_dividers:
            neighborhood, center, idx = divider(src)
            neighborhoods.append(neighborhood)
            centers.append(center)
            idxs.append(idx)
            src = center
        return neighborhoods, centers, idxs


def rand_mask(B, G, mask_ratio):
    """H_Encoder.rand_mask (Point_M2AE.py:85-97) with the reference's host RNG calls: int(ratio * G) masked tokens
    per sample, one np.random.shuffle per sample.  -> bool (B, G) on the host, True = masked."""
    num_mask = int(mask_ratio * G)
    overall = np.zeros([B, G])
    for i in range(B):
        mask = np.hstack([np.zeros(G - num_mask), np.ones(num_mask)])
        np.random.shuffle(mask)
        overall[i, :] = mask
    return torch.from_numpy(overall).to(torch.bool)


@torch.no_grad()
def multi_scale_mask(top_masked, idxs, centers):
    """Multi-scale masking by back-propagation (Point_M2AE.py:107-121): from the mask drawn at the coarsest level,
    a token of level i - 1 stays masked unless a visible token of level i was grouped from it.  top_masked: bool
    (B, G_last) on the device; idxs / centers as HierarchicalGroup returns them.
    -> [bool (B, G_0), ..., bool (B, G_last)] finest level first (the reference's `bool_masked_pos` after its
    reverse()).  The reference's quirk is kept: flat token 0 of a finer level turns visible whenever any coarser
    token is masked (csrc/hier_group.hip)."""
    masks = [top_masked.contiguous()]
    for i in range(len(idxs) - 1, 0, -1):
        B, G, _ = centers[i].shape
        Gc = centers[i - 1].shape[1]
        k = idxs[i].numel() // (B * G)
        parent = masks[-1].reshape(-1).to(torch.uint8)
        child = torch.empty(B * Gc, dtype=torch.uint8, device=parent.device)
        _lib.call('pdae_mask_propagate', parent, B * G, k, B * Gc, _lib.ptr(parent), _lib.ptr(idxs[i]), _lib.ptr(child))
        masks.append(child.bool().reshape(B, Gc))
    masks.reverse()
    return masks


def merge_tokens(x_vis, idx, B, G2, k2):
    """Token merging (Point_M2AE.py:132): the finer level's tokens (B, G1, C) gathered into the coarser level's
    patches by the flat group indices -> (B, G2, k2, C).  Differentiable (index_select)."""
    C = x_vis.shape[-1]
    return x_vis.reshape(-1, C).index_select(0, idx).reshape(B, G2, k2, C)
