"""ShapeNet-55 pretraining set with the loader-side pipeline ON THE DEVICE.

Reference item (datasets/ShapeNet55Dataset.py:90-119): read an 8192-point cloud (datasets/io.py),
`augment_data` ('norm' = centre + scale to the unit sphere, corrupt_util.py:7-17), clean = random
subset of `npoints`; corrupted = `corrupt_data(whole cloud, corrupt_type)` (corrupt_util.py:1046-1096:
'affine_r3' = 1-3 of translate / scale_nonorm / rotate / reflection / shear in random order,
'dropout_local' :590-612) followed by another random subset.  The reference does this per item in 8
DataLoader worker processes; at the ~9 k clouds/s of the training step on one MI355X that is 60 k
argsorts of 8192-point clouds per second on host cores.

Here the stored clouds are resident in HBM (ShapeNet-55's 41 952 training clouds x 8192 x 3 fp32 =
4.1 GB of the 288 GB), a batch is gathered, normalised, corrupted and sub-sampled by device launches
(dropout_local: csrc/pipeline.hip; affine maps and subsets: elementwise / sort launches), and the host
only draws the random parameters (a few hundred numbers per batch) with the reference's
distributions.  Random streams are this module's own (numpy Generator seeded with seed + rank): the
reference's per-worker global numpy / python generators are not reproducible across worker counts
either.

Sources: a directory of .npy clouds listed by `<DATA_PATH>/<subset>.txt` (the reference layout), else
synthetic ShapeNet-shaped clouds (there is no dataset in the image).
"""
import math
import os

import numpy as np
import torch

from . import _lib
from .registry import DATASETS
from .synthetic import labelled_clouds, shapenet_like_clouds

AFFINE = ('translate', 'scale_nonorm', 'rotate', 'reflection', 'shear')
_PASS = ('clean', 'dropout_patch_pointmae', 'Drop-Patch')
_AUGS = ('clean', 'norm', 'scale', 'translate', 'rotate_z', 'rotate')          # corrupt_util.augment_data :1155-1175
_CORRUPTIONS = ('affine_r3', 'dropout_local') + AFFINE + ('rotate_z', 'scale', 'jitter', 'add_global', 'add_local',
                                                          'nonuniform_density')


def pc_normalize(x):
    """(B,P,3) -> centred, max norm 1 per cloud (corrupt_util._pc_normalize)."""
    x = x - x.mean(dim=1, keepdim=True)
    m = x.square().sum(-1).sqrt().amax(dim=1)
    return x / m.view(-1, 1, 1)


def _rot(a):
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def draw_affine_map(rng, name):
    """One map x -> x M + tr with the numpy versions' parameters (corrupt_util.py): 'translate' ADDS
    U(-.5,.5)^3 (:130-141), 'scale_nonorm' U(.5,2)^3 (:82-93), 'rotate' Rz Ry Rx with angles U(-pi,pi)
    (:241-263), 'rotate_z' (:537-570), 'reflection' diag(+-1) (:390-409), 'shear' U(-.5,.5) off-diagonals
    (:412-428); the augmentations 'aug_scale' U(2/3,3/2)^3 (:1105-1108), 'aug_translate' U(-.2,.2)^3
    (:1110-1112)."""
    M, tr = np.eye(3), np.zeros(3)
    if name == 'translate':
        tr = rng.uniform(-0.5, 0.5, 3)
    elif name == 'aug_translate':
        tr = rng.uniform(-0.2, 0.2, 3)
    elif name == 'scale_nonorm':
        M = np.diag(rng.uniform(0.5, 2.0, 3))
    elif name == 'aug_scale':
        M = np.diag(rng.uniform(2.0 / 3.0, 1.5, 3))
    elif name == 'rotate':
        M = _rot(rng.uniform(-math.pi, math.pi, 3))
    elif name == 'rotate_z':
        a = rng.uniform(-math.pi, math.pi)
        M = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    elif name == 'reflection':
        M = np.diag(rng.choice(np.array([1.0, -1.0]), 3))
    elif name == 'shear':
        s = rng.uniform(-0.5, 0.5, 6)
        M = np.array([[1, s[0], s[1]], [s[2], 1, s[3]], [s[4], s[5], 1]])
    else:
        raise NotImplementedError(name)
    return M, tr


def draw_affine(rng, B, names_of):
    """Per cloud the composition of the maps `names_of(rng)` lists -> A (B,3,3), t (B,3), y = x @ A + t."""
    A = np.tile(np.eye(3), (B, 1, 1))
    t = np.zeros((B, 3))
    for b in range(B):
        for name in names_of(rng):
            M, tr = draw_affine_map(rng, name)
            A[b] = A[b] @ M                  # y = (x A + t) M + tr
            t[b] = t[b] @ M + tr
    return A.astype(np.float32), t.astype(np.float32)


def draw_affine_r3(rng, B):
    """'affine_r3' of corrupt_util.corrupt_data (:1062-1070) for B clouds: per cloud 1-3 distinct maps of
    AFFINE in random order."""
    def names(r):
        number = int(r.integers(1, 4))
        return [AFFINE[int(i)] for i in r.choice(len(AFFINE), size=number, replace=False)]
    return draw_affine(rng, B, names)


def apply_affine(x, A, t):
    """y = x A + t per cloud without a batched GEMM: three broadcast multiply-adds."""
    A, t = torch.from_numpy(A).to(x.device), torch.from_numpy(t).to(x.device)
    return (x[..., 0:1] * A[:, None, 0, :] + x[..., 1:2] * A[:, None, 1, :]) + (x[..., 2:3] * A[:, None, 2, :] + t[:, None, :])


def sphere_points(rng, B, n):
    """n points uniform in the unit ball per cloud (corrupt_util._sample_points_inside_unit_sphere :42-56)."""
    r = np.power(rng.uniform(0.0, 1.0, (B, n, 1)), 1.0 / 3.0)
    theta = np.arccos(rng.uniform(-1.0, 1.0, (B, n, 1)))
    phi = rng.uniform(0.0, 2.0 * np.pi, (B, n, 1))
    return np.concatenate([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)],
                          axis=2).astype(np.float32)


def draw_dropout_local(rng, B, P):
    """The draws of corrupt_dropout_local (:590-612) for B clouds of P points: ratio U(.1,.5), 1-7
    clusters, sizes = counts of `total` uniform cluster labels, and per cluster the seed as a rank among
    the survivors (a shuffle's first element).  -> nclusters (B,), seed_rank (B,8), sizes (B,8) int32."""
    ncl = np.zeros(B, np.int32)
    rank = np.zeros((B, 8), np.int32)
    sizes = np.zeros((B, 8), np.int32)
    for b in range(B):
        total = int(P * rng.uniform(0.1, 0.5))
        n = int(rng.integers(1, 8))
        counts = np.bincount(rng.integers(0, n, total), minlength=n)
        alive = P
        ncl[b] = n
        for c in range(n):
            sizes[b, c] = counts[c]
            rank[b, c] = int(rng.integers(0, alive))
            alive -= int(counts[c])
    return ncl, rank, sizes


def dropout_local(x, ncl, rank, sizes):
    """x (B,P,3) on the device + host draws -> alive (B,P) bool (csrc/pipeline.hip)."""
    B, P, _ = x.shape
    dev = x.device
    alive = torch.empty((B, P), dtype=torch.uint8, device=dev)
    n_d, r_d, s_d = (torch.from_numpy(a).to(dev) for a in (ncl, rank, sizes))
    _lib.call('pdae_dropout_local', x, B, P, _lib.ptr(x.contiguous()), _lib.ptr(n_d), _lib.ptr(r_d), _lib.ptr(s_d),
              _lib.ptr(alive))
    return alive.bool()


def random_subset(x, n, alive=None, generator=None):
    """A uniformly random subset of n points per cloud, in random order (ShapeNet.random_sample :76-88);
    `alive` restricts it to the surviving points (clouds with fewer than n survivors are refilled by
    sampling survivors with replacement, as the reference does)."""
    B, P, _ = x.shape
    keys = torch.rand((B, P), device=x.device, generator=generator)
    if alive is not None:
        keys = torch.where(alive, keys, keys + 2.0)          # dead points sort behind every survivor
    order = keys.argsort(dim=1)[:, :n]
    if alive is not None:
        cnt = alive.sum(1, keepdim=True)
        if bool((cnt < n).any()):
            refill = (torch.rand((B, n), device=x.device, generator=generator) * cnt).long().clamp_(max=P - 1)
            pos = torch.arange(n, device=x.device).view(1, n)
            order = torch.where(pos < cnt, order, keys.argsort(dim=1).gather(1, refill))
    return x.gather(1, order.unsqueeze(-1).expand(B, n, 3))


def load_npy_clouds(pc_path, data_path, subset, whole=False, limit=None):
    """The reference's file layout: `<data_path>/<subset>.txt` lists `<taxonomy>-<model>.npy` files
    under `pc_path` (ShapeNet55Dataset.py:37-62).  -> (clouds (num,P,>=3) float32, [(taxonomy, model)])."""
    lines = open(os.path.join(data_path, '%s.txt' % subset)).read().split()
    if whole:
        lines += open(os.path.join(data_path, 'test.txt')).read().split()
    if limit:
        lines = lines[:limit]
    ids, clouds = [], []
    for line in lines:
        tax = line.split('-')[0]
        ids.append((tax, line[len(tax) + 1:].split('.')[0]))
        clouds.append(np.load(os.path.join(pc_path, line)).astype(np.float32))
    return np.stack(clouds, 0), ids


@DATASETS.register_module()
class ShapeNet:
    def __init__(self, config):
        self.npoints = config.get('npoints', 1024)
        self.dense = config.get('N_POINTS', 8192)
        self.bs = config.get('bs', 128)
        # batches per epoch: given, else the reference's len(DataLoader) = ceil(len(set) / (bs * world)) once a listed
        # set is loaded (DistributedSampler shards it, builder.py:19), else 50 synthetic batches
        self.steps = config.get('steps_per_epoch', None)
        self.rank, self.world = int(config.get('rank', 0)), max(int(config.get('world', 1)), 1)
        self.seed = config.get('seed', 0)
        self.device = config.get('device', 'cuda')
        self.pool = config.get('pool', 4)
        self.aug_type = list(config.get('aug_type', ['clean']))
        self.corrupt_type = list(config.get('corrupt_type', ['clean']))
        self.subset = config.get('subset', 'train')
        self.pc_path, self.data_path = config.get('PC_PATH'), config.get('DATA_PATH')
        self.whole = bool(config.get('whole', False))
        for item in self.corrupt_type:
            if not (item in _PASS or 'dropout_global' in item or item in _CORRUPTIONS):
                raise NotImplementedError('loader-side corruption %r (implemented: %s)' % (item, ', '.join(_CORRUPTIONS)))
        for item in self.aug_type:
            if item not in _AUGS:
                raise NotImplementedError('augmentation %r (implemented: %s)' % (item, ', '.join(_AUGS)))
        self.rng = np.random.default_rng(self.seed)
        self.gen = None
        self._clouds, self.ids = None, None

    def _materialise(self):
        listed = self.pc_path and self.data_path and os.path.exists(os.path.join(str(self.data_path), '%s.txt' % self.subset))
        if listed:
            clouds, self.ids = load_npy_clouds(self.pc_path, self.data_path, self.subset, self.whole)
            clouds = clouds[:, :, :3]
        else:
            n = self.bs * self.pool
            clouds = shapenet_like_clouds(n, self.dense, seed=self.seed, dense=max(self.dense, 8192))
            self.ids = [('synthetic', str(i)) for i in range(n)]
        self._clouds = torch.from_numpy(np.ascontiguousarray(clouds)).to(self.device)
        self.gen = torch.Generator(device=self._clouds.device)
        self.gen.manual_seed(self.seed)
        self.listed = bool(listed)
        if self.steps is None:
            self.steps = -(-clouds.shape[0] // (self.bs * self.world)) if listed else 50

    def __len__(self):
        if self.steps is None:
            self._materialise()
        return self.steps

    def batch(self, index):
        """-> (corrupted (B,npoints,3), clean (B,npoints,3)) on the device."""
        x = self._clouds.index_select(0, index)
        B = x.shape[0]
        for item in self.aug_type:                                  # in the configured order (:1157-1173)
            if item == 'norm':
                x = pc_normalize(x)
            elif item in ('scale', 'translate'):
                x = apply_affine(x, *draw_affine(self.rng, B, lambda r, n='aug_' + item: [n]))
            elif item in ('rotate_z', 'rotate'):
                x = apply_affine(x, *draw_affine(self.rng, B, lambda r, n=item: [n]))
        B, P, _ = x.shape
        clean = random_subset(x, self.npoints, generator=self.gen)
        y, alive, touched = x, None, False
        for item in self.corrupt_type:
            if item in _PASS or 'dropout_global' in item:
                continue                                            # applied in the model's forward
            touched = True
            if item == 'affine_r3':
                y = apply_affine(y, *draw_affine_r3(self.rng, B))
            elif item in AFFINE or item == 'rotate_z':
                y = apply_affine(y, *draw_affine(self.rng, B, lambda r, n=item: [n]))
            elif item == 'scale':                                   # U(.5,2)^3 then re-normalised (:59-69)
                y = pc_normalize(apply_affine(y, *draw_affine(self.rng, B, lambda r: ['scale_nonorm'])))
            elif item == 'jitter':                                  # sigma = 0.01 (level + 1), level U(0,4) (:179-191)
                sigma = torch.from_numpy((0.01 * (self.rng.uniform(0.0, 4.0, (B, 1, 1)) + 1.0)).astype(np.float32))
                y = y + sigma.to(y.device) * torch.randn(y.shape, device=y.device, generator=self.gen)
            elif item == 'add_global':                              # +50 % points uniform in the unit ball (:830-841, level 4)
                extra = torch.from_numpy(sphere_points(self.rng, B, int(y.shape[1] * 0.5))).to(y.device)
                y = torch.cat([y, extra], dim=1)
                if alive is not None:
                    alive = torch.cat([alive, torch.ones(extra.shape[:2], dtype=torch.bool, device=y.device)], dim=1)
            elif item == 'add_local':                               # +50 % points in 1-7 Gaussian clusters around random
                P2 = y.shape[1]                                     # points of the cloud (:844-870, level 4)
                total = int(P2 * 0.5)
                seeds = np.zeros((B, total), np.int64)              # per added point: which cloud point it sits on
                sig = np.zeros((B, total, 1), np.float32)
                for b in range(B):
                    n = int(self.rng.integers(1, 8))
                    counts = np.bincount(self.rng.integers(0, n, total), minlength=n)
                    pts = self.rng.integers(0, P2, n)               # (the reference shuffles and takes the first n)
                    seeds[b] = np.repeat(pts, counts)
                    sig[b, :, 0] = np.repeat(self.rng.uniform(0.075, 0.125, n), counts)
                centre = y.gather(1, torch.from_numpy(seeds).to(y.device).unsqueeze(-1).expand(B, total, 3))
                extra = centre + torch.from_numpy(sig).to(y.device) * torch.randn((B, total, 3), device=y.device,
                                                                                   generator=self.gen)
                d2 = extra.square().sum(-1, keepdim=True)
                extra = torch.where(d2 > 1, extra / d2, extra)      # pulled back inside the unit sphere as the reference
                y = torch.cat([y, extra], dim=1)
                if alive is not None:
                    alive = torch.cat([alive, torch.ones((B, total), dtype=torch.bool, device=y.device)], dim=1)
            elif item == 'nonuniform_density':                      # distance-gated drop from a random viewpoint (:875-897)
                gate = torch.from_numpy((self.rng.uniform(0.0, 4.0, (B, 1)) / 4.0 + 0.1).astype(np.float32)).to(y.device)
                v = self.rng.normal(0.0, 1.0, (B, 3))
                v = torch.from_numpy((v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)).to(y.device)
                dist = (y - v[:, None, :]).square().sum(-1).sqrt() / 2.0        # (d - (|v| - 1)) / 2 with |v| = 1
                keep = dist * gate < torch.rand(dist.shape, device=y.device, generator=self.gen)
                alive = keep if alive is None else (alive & keep)
            elif item == 'dropout_local':
                keep = dropout_local(y.contiguous(), *draw_dropout_local(self.rng, B, y.shape[1]))
                alive = keep if alive is None else (alive & keep)
        corrupted = random_subset(y, self.npoints, alive, self.gen) if touched else clean
        return corrupted, clean

    def __iter__(self):
        if self._clouds is None:
            self._materialise()
        n = self._clouds.shape[0]
        if self.listed and self.world > 1:
            # DistributedSampler semantics: ONE permutation per epoch shared by the ranks (seeded without the
            # rank), padded by wrapping to a multiple of world, rank r takes every world-th index
            self._epoch = getattr(self, '_epoch', -1) + 1
            shared = np.random.default_rng([self.seed - self.rank, self._epoch]).permutation(n)
            total = -(-n // self.world) * self.world
            order = np.resize(shared, total)[self.rank::self.world]
            n = order.shape[0]
        else:
            order = self.rng.permutation(n)
        for i in range(self.steps):
            sel = order[(np.arange(self.bs) + i * self.bs) % order.shape[0]]
            corrupted, clean = self.batch(torch.from_numpy(sel).to(self._clouds.device))
            yield self.ids[int(sel[0])][0], i, corrupted, clean


@DATASETS.register_module()
class ModelNet:
    """Labelled clouds for the SVM probe (datasets/ModelNetDataset.py of the reference yields
    (taxonomy, model_id, (points, label))).  No ModelNet40 files ship with the image: labelled synthetic
    clouds stand in (synthetic.labelled_clouds); `count` clouds, batches of `bs`, resident on the device."""

    def __init__(self, config):
        self.npoints = config.get('npoints', 1024)
        self.bs = config.get('bs', 32)
        self.count = config.get('count', 256)
        self.subset = config.get('subset', 'test')
        self.device = config.get('device', 'cuda')
        seed = config.get('seed', 0) + (1000 if self.subset == 'train' else 2000)
        x, y = labelled_clouds(self.count, self.npoints, seed=seed, classes=min(3, config.get('NUM_CATEGORY', 3)))
        rank, world = int(config.get('rank', 0)), max(int(config.get('world', 1)), 1)
        if world > 1:      # DistributedSampler(shuffle=False): wrap-padded to a multiple of world, every world-th item
            sel = np.resize(np.arange(self.count), -(-self.count // world) * world)[rank::world]
            x, y = x[sel], y[sel]
            self.count = len(sel)
        self.x, self.y = torch.from_numpy(x).to(self.device), torch.from_numpy(y).to(self.device)

    def __len__(self):
        return (self.count + self.bs - 1) // self.bs

    def __iter__(self):
        for i in range(0, self.count, self.bs):
            yield 'ModelNet', i, (self.x[i:i + self.bs], self.y[i:i + self.bs])
