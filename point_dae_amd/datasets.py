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


def pc_normalize(x):
    """(B,P,3) -> centred, max norm 1 per cloud (corrupt_util._pc_normalize)."""
    x = x - x.mean(dim=1, keepdim=True)
    m = x.square().sum(-1).sqrt().amax(dim=1)
    return x / m.view(-1, 1, 1)


def draw_affine_r3(rng, B):
    """'affine_r3' of corrupt_util.corrupt_data (:1062-1070) for B clouds: per cloud 1-3 distinct maps of
    AFFINE in random order, each x -> x A (+ t) with the numpy versions' parameters (translate ADDS
    U(-.5,.5)^3 :130-141, scale U(.5,2)^3 :82-93, rotate Rz Ry Rx with angles U(-pi,pi) :241-263,
    reflection diag(+-1) :390-409, shear U(-.5,.5) off-diagonals :412-428).  -> A (B,3,3), t (B,3)
    with y = x @ A + t (the composition of the chosen maps)."""
    A = np.tile(np.eye(3), (B, 1, 1))
    t = np.zeros((B, 3))
    for b in range(B):
        number = int(rng.integers(1, 4))
        for name in rng.choice(len(AFFINE), size=number, replace=False):
            name = AFFINE[int(name)]
            M, tr = np.eye(3), np.zeros(3)
            if name == 'translate':
                tr = rng.uniform(-0.5, 0.5, 3)
            elif name == 'scale_nonorm':
                M = np.diag(rng.uniform(0.5, 2.0, 3))
            elif name == 'rotate':
                a = rng.uniform(-math.pi, math.pi, 3)
                Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
                Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
                Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
                M = Rz @ Ry @ Rx
            elif name == 'reflection':
                M = np.diag(rng.choice(np.array([1.0, -1.0]), 3))
            else:
                s = rng.uniform(-0.5, 0.5, 6)
                M = np.array([[1, s[0], s[1]], [s[2], 1, s[3]], [s[4], s[5], 1]])
            A[b] = A[b] @ M                  # y = (x A + t) M + tr
            t[b] = t[b] @ M + tr
    return A.astype(np.float32), t.astype(np.float32)


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
        self.steps = config.get('steps_per_epoch', 50)
        self.seed = config.get('seed', 0)
        self.device = config.get('device', 'cuda')
        self.pool = config.get('pool', 4)
        self.aug_type = list(config.get('aug_type', ['clean']))
        self.corrupt_type = list(config.get('corrupt_type', ['clean']))
        self.subset = config.get('subset', 'train')
        self.pc_path, self.data_path = config.get('PC_PATH'), config.get('DATA_PATH')
        self.whole = bool(config.get('whole', False))
        for item in self.corrupt_type:
            if not (item in _PASS or 'dropout_global' in item or item in ('affine_r3', 'dropout_local')):
                raise NotImplementedError('loader-side corruption %r (implemented: affine_r3, dropout_local)' % item)
        for item in self.aug_type:
            if item not in ('clean', 'norm'):
                raise NotImplementedError('augmentation %r (implemented: norm)' % item)
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

    def __len__(self):
        return self.steps

    def batch(self, index):
        """-> (corrupted (B,npoints,3), clean (B,npoints,3)) on the device."""
        x = self._clouds.index_select(0, index)
        if 'norm' in self.aug_type:
            x = pc_normalize(x)
        B, P, _ = x.shape
        clean = random_subset(x, self.npoints, generator=self.gen)
        y, alive, touched = x, None, False
        for item in self.corrupt_type:
            if item == 'affine_r3':
                A, t = draw_affine_r3(self.rng, B)
                A, t = torch.from_numpy(A).to(x.device), torch.from_numpy(t).to(x.device)
                # y = x A + t without a batched GEMM: three broadcast multiply-adds
                y = (y[..., 0:1] * A[:, None, 0, :] + y[..., 1:2] * A[:, None, 1, :]) + (y[..., 2:3] * A[:, None, 2, :] + t[:, None, :])
                touched = True
            elif item == 'dropout_local':
                alive = dropout_local(y.contiguous(), *draw_dropout_local(self.rng, B, P))
                touched = True
        corrupted = random_subset(y, self.npoints, alive, self.gen) if touched else clean
        return corrupted, clean

    def __iter__(self):
        if self._clouds is None:
            self._materialise()
        n = self._clouds.shape[0]
        order = self.rng.permutation(n)
        for i in range(self.steps):
            sel = order[(np.arange(self.bs) + i * self.bs) % n]
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
        self.x, self.y = torch.from_numpy(x).to(self.device), torch.from_numpy(y).to(self.device)

    def __len__(self):
        return (self.count + self.bs - 1) // self.bs

    def __iter__(self):
        for i in range(0, self.count, self.bs):
            yield 'ModelNet', i, (self.x[i:i + self.bs], self.y[i:i + self.bs])
