"""ShapeNet-55 pretraining set with the loader-side pipeline ON THE DEVICE.

Reference item (datasets/ShapeNet55Dataset.py:90-119): read an 8192-point cloud (datasets/io.py),
`augment_data` ('norm' = centre + scale to the unit sphere, corrupt_util.py:7-17), clean = random
subset of `npoints`; corrupted = `corrupt_data(whole cloud, corrupt_type)` (corrupt_util.py:1046-1096:
'affine_r3' = 1-3 of translate / scale_nonorm / rotate / reflection / shear in random order,
'dropout_local' :590-612) followed by another random subset.  The reference does this per item in 8
DataLoader worker processes; at the ~9 k clouds/s of the training step on one MI355X that is 60 k
argsorts of 8192-point clouds per second on host cores.

Here the stored clouds are resident in HBM (ShapeNet-55's 41 952 training clouds x 8192 x 3 fp32 =
4.1 GB of the 288 GB) and a batch is gathered, normalised, corrupted and sub-sampled by the kernels of
csrc/pipeline.hip -- norm + affine maps + jitter in one launch, add_global / add_local / nonuniform_density /
dropout_local one launch each, the random subset one launch -- each a pure function of the clouds and of
random draws, pinned on the LIVE reference's `ShapeNet.__getitem__` with its draws recorded
(tests/golden/make_loader_fixtures.py, tests/test_pipeline.py).  In production the host draws the
small parameters (maps, levels, cluster sizes: a few hundred numbers per batch) with the reference's
distributions and the device draws the bulk noise / keys.  Random streams are this module's own (numpy
Generator seeded with seed + rank, torch device generator): the reference's per-worker global numpy /
python generators are not reproducible across worker counts either.

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


def _dev(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=device, dtype=dtype) if dtype is not None else t.to(device)


def pipeline_norm_affine(x, normalise=False, maps=None, sigma=None, noise=None, stride=None):
    """x (B,P,3) -> y (B,stride,3) with rows [0,P) = jitter(affine maps(norm(x))) (csrc/pipeline.hip norm_affine).
    maps: list per cloud of up to three (M (3,3), t (3,)) pairs applied in order, y = x M + t; sigma (B,) and
    noise (B,P,3) on the device: y += sigma * noise."""
    B, P, _ = x.shape
    stride = P if stride is None else stride
    dev = x.device
    y = torch.empty((B, stride, 3), device=dev)
    nm = mp = None
    if maps is not None:
        packed = np.zeros((B, 3, 12), np.float32)
        packed[:, :, 0] = packed[:, :, 4] = packed[:, :, 8] = 1.0
        counts = np.zeros(B, np.int32)
        for b, ms in enumerate(maps):
            counts[b] = len(ms)
            for q, (M, t) in enumerate(ms):
                packed[b, q, :9], packed[b, q, 9:] = np.asarray(M, np.float64).reshape(-1), np.asarray(t, np.float64)
        nm, mp = _dev(counts, dev), _dev(packed, dev)
    _lib.call('pdae_pipeline_norm_affine', x, B, P, stride, int(bool(normalise)), _lib.ptr(x.contiguous()), _lib.ptr(nm),
              _lib.ptr(mp), _lib.ptr(sigma), _lib.ptr(noise), _lib.ptr(y))
    return y


def pipeline_add_global(y, p0, count, u):
    """append count[b] points uniform in the unit ball behind row p0 of every cloud; u (B,nmax,3) uniforms."""
    B, stride, _ = y.shape
    _lib.call('pdae_pipeline_add_global', y, B, u.shape[1], stride, p0, _lib.ptr(count), _lib.ptr(u), _lib.ptr(y))


def pipeline_add_local(y, p0, count, seed, sigma, noise):
    """append count[b] cluster points: cloud point seed[b,j] + sigma[b,j] * noise[b,j], pulled inside the unit sphere."""
    B, stride, _ = y.shape
    _lib.call('pdae_pipeline_add_local', y, B, seed.shape[1], stride, p0, _lib.ptr(count), _lib.ptr(seed), _lib.ptr(sigma),
              _lib.ptr(noise), _lib.ptr(y))


def pipeline_density(y, p, view, gate, r, alive):
    """alive[b,k] &= normalised distance to view[b] * gate[b] < r[b,k] for the first p rows."""
    B, stride, _ = y.shape
    _lib.call('pdae_pipeline_density', y, B, p, stride, _lib.ptr(y), _lib.ptr(view), _lib.ptr(gate), _lib.ptr(r),
              _lib.ptr(alive))


def pipeline_subset(y, p, n, keys, alive=None):
    """-> (B,n,3): the n alive points with the smallest keys among the first p rows, in key order."""
    B, stride, _ = y.shape
    out = torch.empty((B, n, 3), device=y.device)
    _lib.call('pdae_pipeline_subset', y, B, p, stride, n, _lib.ptr(y), _lib.ptr(alive), _lib.ptr(keys), _lib.ptr(out))
    return out


def draw_affine_maps(rng, names):
    """[(M, t)] for the named maps, in order (draw_affine_map's parameter ranges)."""
    return [draw_affine_map(rng, n) for n in names]



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
        """-> (corrupted (B,npoints,3), clean (B,npoints,3)) on the device, every stage on csrc/pipeline.hip:
        ShapeNet.__getitem__ (:90-119) = augment_data -> random_sample -> corrupt_data -> random_sample."""
        x = self._clouds.index_select(0, index)
        B, P, _ = x.shape
        dev, rng = x.device, self.rng
        # ---- augment_data (:1155-1175), in the configured order: 'norm' and the affine augmentations.  'norm' is
        # folded into the kernel when it comes first (the shipped configurations); otherwise it is its own pass
        aug = [a for a in self.aug_type if a != 'clean']
        norm_first = bool(aug) and aug[0] == 'norm'
        maps = [[] for _ in range(B)]
        for item in aug[1:] if norm_first else aug:
            if item == 'norm':
                x = pipeline_norm_affine(x, True, maps if any(maps) else None)
                maps = [[] for _ in range(B)]
                continue
            name = 'aug_' + item if item in ('scale', 'translate') else item
            for b in range(B):
                maps[b].append(draw_affine_map(rng, name))
        while max(len(m) for m in maps) > 3:                        # more than three augmentation maps: extra passes
            x = pipeline_norm_affine(x, norm_first, [m[:3] for m in maps])
            maps, norm_first = [m[3:] for m in maps], False
        data = pipeline_norm_affine(x, norm_first, maps if any(maps) else None)
        clean = pipeline_subset(data, P, self.npoints, torch.rand((B, P), device=dev, generator=self.gen))
        # ---- corrupt_data (:1046-1096) on the whole cloud
        items = [c for c in self.corrupt_type if not (c in _PASS or 'dropout_global' in c)]   # those run in the model's forward
        if not items:
            return clean, clean
        n_add = int(P * 0.5) if any(c in ('add_global', 'add_local') for c in items) else 0   # level <= 4: at most +50 %
        stride = P + n_add
        maps = [[] for _ in range(B)]
        sigma = noise = None
        tail = []
        for item in items:                                          # the affine maps and jitter of the list: one launch
            if item == 'affine_r3':
                for b in range(B):
                    number = int(rng.integers(1, 4))
                    for i in rng.choice(len(AFFINE), size=number, replace=False):
                        maps[b].append(draw_affine_map(rng, AFFINE[int(i)]))
            elif item in AFFINE or item == 'rotate_z':
                for b in range(B):
                    maps[b].append(draw_affine_map(rng, item))
            elif item == 'jitter' and not tail:                     # sigma = 0.01 (level + 1), level in 0..4 (:1090-1092)
                sigma = _dev((0.01 * (rng.integers(0, 5, B) + 1.0)).astype(np.float32), dev)
                noise = torch.randn((B, P, 3), device=dev, generator=self.gen)
            else:
                tail.append(item)
        if max(len(m) for m in maps) > 3:
            raise NotImplementedError('more than three affine maps in one corrupt_type list')
        y = pipeline_norm_affine(data[:, :P].contiguous() if stride != P else data, False, maps if any(maps) else None,
                                 sigma, noise, stride) if (any(maps) or noise is not None or stride != P) else data
        cur, alive = P, None                                        # rows in use; survivors
        for item in tail:
            if item in ('add_global', 'add_local') and (alive is not None or cur != P):
                raise NotImplementedError('%s after a drop / another add in one corrupt_type list' % item)
            if item == 'add_global':                                # int(P (level + 1) 0.1) ball points (:830-841)
                count = (P * (rng.integers(0, 5, B) + 1) * 0.1).astype(np.int32)
                pipeline_add_global(y, P, _dev(count, dev), torch.rand((B, n_add, 3), device=dev, generator=self.gen)
                                    * torch.tensor([1.0, 2.0, 2.0 * math.pi], device=dev) - torch.tensor([0.0, 1.0, 0.0], device=dev))
                cur, added = P + n_add, count
            elif item == 'add_local':                               # 1-7 Gaussian clusters on random cloud points (:844-870)
                count = (P * (rng.integers(0, 5, B) + 1) * 0.1).astype(np.int32)
                seed = np.zeros((B, n_add), np.int32)
                sig = np.zeros((B, n_add), np.float32)
                for b in range(B):
                    ncl = int(rng.integers(1, 8))
                    sizes = np.bincount(rng.integers(0, ncl, int(count[b])), minlength=ncl)
                    pts = rng.choice(P, ncl, replace=False)         # the first ncl points of a shuffle
                    seed[b, :count[b]] = np.repeat(pts, sizes)
                    sig[b, :count[b]] = np.repeat(rng.uniform(0.075, 0.125, ncl), sizes)
                pipeline_add_local(y, P, _dev(count, dev), _dev(seed, dev), _dev(sig, dev),
                                   torch.randn((B, n_add, 3), device=dev, generator=self.gen))
                cur, added = P + n_add, count
            elif item == 'nonuniform_density':                      # (:875-897) gate = level / 4 + 0.1, level in 0..4
                if alive is None:
                    alive = torch.ones((B, stride), dtype=torch.uint8, device=dev)
                v = rng.normal(0.0, 1.0, (B, 3))
                v = v / np.linalg.norm(v, axis=1, keepdims=True)
                gate = rng.integers(0, 5, B) / 4.0 + 0.1
                pipeline_density(y, cur, _dev(v.astype(np.float32), dev), _dev(gate.astype(np.float32), dev),
                                 torch.rand((B, cur), device=dev, generator=self.gen), alive)
            elif item == 'dropout_local':
                if cur != P or stride != P:
                    raise NotImplementedError('dropout_local after an add in one corrupt_type list')
                keep = dropout_local(y, *draw_dropout_local(rng, B, P)).to(torch.uint8)
                alive = keep if alive is None else (alive & keep)
            elif item == 'scale':                                   # U(.5,2)^3 then re-normalised (:59-69)
                y = pipeline_norm_affine(pipeline_norm_affine(y, False, [[draw_affine_map(rng, 'scale_nonorm')] for _ in range(B)]), True)
            else:
                raise NotImplementedError(item)
        if cur != P:                                                # rows behind P + added[b] are not part of cloud b
            if alive is None:
                alive = torch.ones((B, stride), dtype=torch.uint8, device=dev)
            alive &= (torch.arange(stride, device=dev).view(1, -1) < (P + _dev(added, dev)).view(-1, 1)).to(torch.uint8)
        corrupted = pipeline_subset(y, cur, self.npoints, torch.rand((B, stride), device=dev, generator=self.gen), alive)
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
