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
AFFINE_V2 = ('translate', 'scale_nonorm', 'rotate_level1', 'reflection', 'shear_1p')        # corrupt_util.py:1043
# 'affine_rN[_v2]' of corrupt_util.corrupt_data (:1053-1088): 1..N distinct maps of the pool in random order
_AFFINE_SETS = {'affine_r3': (AFFINE, 3), 'affine_r5': (AFFINE, 5), 'affine_r3_v2': (AFFINE_V2, 3), 'affine_r5_v2': (AFFINE_V2, 5)}
# the single maps of the reference's `corruptions` table (corrupt_util.py:984-1038) as (kind, parameter); the parameter
# ranges are pinned on the live functions' recorded draws (tests/golden/make_loader_variant_fixtures.py)
_MAP_PARAMS = {
    'translate': ('translate', 0.5), 'translate_tiny': ('translate', 0.1), 'translate_middle': ('translate', 0.3),
    'translate_too_large': ('translate', 0.8),                                              # x + U(-s, s)^3       :130-177
    'scale_nonorm': ('scale', 2.0), 'scale_nonorm_1p5': ('scale', 1.5), 'scale_nonorm_4': ('scale', 4.0),
    'scale_nonorm_10': ('scale', 10.0),                                                     # x * U(1/s, s)^3      :82-128
    'rotate': ('rotate', math.pi),                                                          # Rz Ry Rx, U(-c, c)^3 :241-263
    'rotate_level0': ('rotate', math.pi / 5), 'rotate_level1': ('rotate', 2 * math.pi / 5),
    'rotate_level2': ('rotate', 3 * math.pi / 5), 'rotate_level3': ('rotate', 4 * math.pi / 5),
    'rotate_level4': ('rotate', math.pi),                                                   # c = pi / 5 (l + 1)   :265-388
    'rotate_z': ('rotate_z_level', math.pi / 5),                                            # c = pi / 5 (level+1) :537-570
    'reflection': ('reflection', None),                                                     # diag(+-1)            :390-409
    'shear': ('shear', 0.5), 'shear_p1': ('shear', 0.1), 'shear_p3': ('shear', 0.3), 'shear_p8': ('shear', 0.8),
    'shear_1p': ('shear', 1.0), 'shear_2p': ('shear', 2.0),                                 # off-diagonals U(-c, c) :412-518
    'shear_small': ('shear_level', 0.02),                                                   # c = 0.02 (level + 1) :520-535
    # augment_data (:1105-1175): PointcloudScale U(2/3, 3/2)^3, PointcloudTranslate U(-.2, .2)^3, full-circle rotations
    'aug_scale': ('scale', 1.5), 'aug_translate': ('translate', 0.2), 'aug_rotate_z': ('rotate_z', math.pi),
    'aug_rotate': ('rotate', math.pi),
}
_JITTER = {'jitter': None, 'jitter_p01': 0.01, 'jitter_p03': 0.03, 'jitter_p05': 0.05, 'jitter_p1': 0.1}   # :179-239
# dropout_local variants (:590-828): (drop ratio or None = U(.1, .5), exclusive upper bound of randint(1, hi) clusters or
# 0 = exactly one cluster without a draw)
_DROPOUT_LOCAL = {'dropout_local': (None, 8), 'dropout_local_c5d1': (0.1, 5), 'dropout_local_c5d3': (0.3, 5),
                  'dropout_local_c5d5': (0.5, 5), 'dropout_local_c5d7': (0.7, 5), 'dropout_local_c5d9': (0.9, 5),
                  'dropout_local_c1d3': (0.3, 0), 'dropout_local_c2d3': (0.3, 2), 'dropout_local_c3d3': (0.3, 3),
                  'dropout_local_c8d3': (0.3, 8)}
_SCALE_SINGLE = (1.6, 1.7, 1.8, 1.9, 2.0)                                                   # :71-80, by level
_PASS = ('clean', 'dropout_patch_pointmae', 'Drop-Patch')
_AUGS = ('clean', 'norm', 'scale', 'translate', 'rotate_z', 'rotate')          # corrupt_util.augment_data :1155-1175
_CORRUPTIONS = (tuple(_AFFINE_SETS) + tuple(n for n in _MAP_PARAMS if not n.startswith('aug_')) + tuple(_JITTER)
                + tuple(_DROPOUT_LOCAL) + ('scale', 'scale_single', 'add_global', 'add_local', 'nonuniform_density'))
# names some reference YAMLs carry that the reference's own dispatcher does not know (a KeyError there too)
_UNKNOWN_UPSTREAM = ('affine_r3_tiny', 'affine_r3_middle', 'scan')


def _rot(a):
    Rx = np.array([[1, 0, 0], [0, np.cos(a[0]), -np.sin(a[0])], [0, np.sin(a[0]), np.cos(a[0])]])
    Ry = np.array([[np.cos(a[1]), 0, np.sin(a[1])], [0, 1, 0], [-np.sin(a[1]), 0, np.cos(a[1])]])
    Rz = np.array([[np.cos(a[2]), -np.sin(a[2]), 0], [np.sin(a[2]), np.cos(a[2]), 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def map_needs_level(name):
    return _MAP_PARAMS[name][0] in ('rotate_z_level', 'shear_level')


def affine_map_from_draw(name, draw):
    """(M, tr) of the map `name` for the values the numpy version draws (one uniform / choice call, corrupt_util.py)."""
    kind = _MAP_PARAMS[name][0]
    kind = kind[:-6] if kind.endswith('_level') else kind
    draw = np.asarray(draw, np.float64).reshape(-1)
    M, tr = np.eye(3), np.zeros(3)
    if kind == 'translate':
        tr = draw
    elif kind in ('scale', 'reflection'):
        M = np.diag(draw)
    elif kind == 'rotate':
        M = _rot(draw)
    elif kind == 'rotate_z':
        a = draw[0]
        M = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    elif kind == 'shear':
        M = np.array([[1, draw[0], draw[1]], [draw[2], 1, draw[3]], [draw[4], draw[5], 1]])
    return M, tr


def draw_affine_map(rng, name, level=None):
    """One map x -> x M + tr of the table above with the numpy versions' parameters (corrupt_util.py); `level` (0..4,
    the dispatcher's random.choice, :1090-1092) only matters to 'rotate_z' and 'shear_small' and is drawn here when
    not given."""
    if name not in _MAP_PARAMS:
        raise NotImplementedError(name)
    kind, par = _MAP_PARAMS[name]
    if kind.endswith('_level'):
        level = int(rng.integers(0, 5)) if level is None else int(level)
        kind, par = kind[:-6], par * (level + 1)
    if kind == 'translate':
        draw = rng.uniform(-par, par, 3)
    elif kind == 'scale':
        draw = rng.uniform(1.0 / par, par, 3)
    elif kind == 'rotate':
        draw = rng.uniform(-par, par, 3)
    elif kind == 'rotate_z':
        draw = rng.uniform(-par, par, 1)
    elif kind == 'reflection':
        draw = rng.choice(np.array([1.0, -1.0]), 3)
    else:
        draw = rng.uniform(-par, par, 6)
    return affine_map_from_draw(name, draw)


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


def draw_affine_set(rng, B, item='affine_r3'):
    """'affine_r3' / 'affine_r5' / '..._v2' of corrupt_util.corrupt_data (:1053-1088) for B clouds: per cloud 1-N distinct
    maps of the pool in random order."""
    pool, most = _AFFINE_SETS[item]

    def names(r):
        number = int(r.integers(1, most + 1))
        return [pool[int(i)] for i in r.choice(len(pool), size=number, replace=False)]
    return draw_affine(rng, B, names)


def draw_affine_r3(rng, B):
    return draw_affine_set(rng, B, 'affine_r3')



def sphere_points(rng, B, n):
    """n points uniform in the unit ball per cloud (corrupt_util._sample_points_inside_unit_sphere :42-56)."""
    r = np.power(rng.uniform(0.0, 1.0, (B, n, 1)), 1.0 / 3.0)
    theta = np.arccos(rng.uniform(-1.0, 1.0, (B, n, 1)))
    phi = rng.uniform(0.0, 2.0 * np.pi, (B, n, 1))
    return np.concatenate([r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)],
                          axis=2).astype(np.float32)


def draw_dropout_local(rng, B, P, item='dropout_local'):
    """The draws of corrupt_dropout_local (:590-612) and its _cXdY variants (:614-828) for B clouds of P points: ratio
    U(.1,.5) or the variant's constant, 1..hi-1 clusters, sizes = counts of `total` uniform cluster labels, and per
    cluster the seed as a rank among the survivors (a shuffle's first element).
    -> nclusters (B,), seed_rank (B,8), sizes (B,8) int32."""
    ratio, hi = _DROPOUT_LOCAL[item]
    ncl = np.zeros(B, np.int32)
    rank = np.zeros((B, 8), np.int32)
    sizes = np.zeros((B, 8), np.int32)
    for b in range(B):
        total = int(P * (rng.uniform(0.1, 0.5) if ratio is None else ratio))
        n = int(rng.integers(1, hi)) if hi else 1
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
    n_d, r_d, s_d = (_dev(a, dev) for a in (ncl, rank, sizes))
    _lib.call('pdae_dropout_local', x, B, P, _lib.ptr(x.contiguous()), _lib.ptr(n_d), _lib.ptr(r_d), _lib.ptr(s_d),
              _lib.ptr(alive))
    return alive.bool()


def _dev(a, device, dtype=None):
    """Host draws -> device WITHOUT blocking the host: a pageable source makes .to() a synchronous, stream-ordered copy,
    i.e. the host would wait for everything queued on the stream (the previous optimisation step) at every small
    parameter tensor of a batch.  Pinned staging (torch's caching host allocator keeps the block until the copy ran)."""
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    if torch.device(device).type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


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
        self.shared_seed = config.get('shared_seed', None)
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
            if item in _UNKNOWN_UPSTREAM:
                raise NotImplementedError('loader-side corruption %r: the reference\'s own dispatcher (corrupt_util.py:984-1096) '
                                          'has no such entry either (KeyError there)' % item)
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
            if listed:
                per_rank = -(-clouds.shape[0] // self.world)        # DistributedSampler pads to ceil(N / world)
                # tools/builder.py:21,28: drop_last = (subset == 'train') => floor for the training loader, ceil otherwise
                self.steps = max(per_rank // self.bs, 1) if self.subset == 'train' else -(-per_rank // self.bs)
            else:
                self.steps = 50

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
        # ---- augment_data (:1155-1175), strictly in the configured order.  The kernel normalises FIRST and applies its
        # (<= 3) maps second, so a 'norm' is folded into the launch of the maps that FOLLOW it; maps drawn before a
        # 'norm' are flushed (normalise off, three per pass) before that 'norm' gets its own launch.
        aug = [a for a in self.aug_type if a != 'clean']
        maps = [[] for _ in range(B)]
        norm_pending = False

        def flush(x, norm_pending, maps):
            """apply [norm?] + the pending maps in order, three maps per pass -> x"""
            while norm_pending or any(maps):
                x = pipeline_norm_affine(x, norm_pending, [m[:3] for m in maps] if any(maps) else None)
                maps, norm_pending = [m[3:] for m in maps], False
            return x
        for item in aug:
            if item == 'norm':
                x = flush(x, norm_pending, maps)                     # everything drawn so far comes before this norm
                maps, norm_pending = [[] for _ in range(B)], True
                continue
            for b in range(B):
                maps[b].append(draw_affine_map(rng, 'aug_' + item))
        data = flush(x, norm_pending, maps) if (norm_pending or any(maps)) else pipeline_norm_affine(x, False, None)
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
            if item in _AFFINE_SETS or item in _MAP_PARAMS:
                if noise is not None or tail:
                    raise NotImplementedError('an affine map behind jitter / an add / a drop in one corrupt_type list')
                for b in range(B):
                    if item in _AFFINE_SETS:
                        pool, most = _AFFINE_SETS[item]
                        number = int(rng.integers(1, most + 1))
                        for i in rng.choice(len(pool), size=number, replace=False):
                            maps[b].append(draw_affine_map(rng, pool[int(i)]))
                    else:
                        maps[b].append(draw_affine_map(rng, item))
            elif item in _JITTER and not tail and noise is None:
                # sigma = 0.01 (level + 1), level in 0..4 (:179-191, :1090-1092), or the variant's constant
                fixed = _JITTER[item]
                sig = 0.01 * (rng.integers(0, 5, B) + 1.0) if fixed is None else np.full(B, fixed)
                sigma = _dev(sig.astype(np.float32), dev)
                noise = torch.randn((B, P, 3), device=dev, generator=self.gen)
            elif item in _JITTER:
                raise NotImplementedError('jitter behind an add / a drop / another jitter in one corrupt_type list')
            else:
                tail.append(item)
        y = data
        if any(maps) or noise is not None or stride != P:
            passes = max(1, -(-max(len(m) for m in maps) // 3))      # three maps per launch ('affine_r5': up to two)
            for c in range(passes):
                last = c == passes - 1
                part = [m[3 * c:3 * c + 3] for m in maps]
                y = pipeline_norm_affine(y, False, part if any(part) else None, sigma if last else None,
                                         noise if last else None, stride if last else None)
        cur, alive = P, None                                        # rows in use; survivors
        for item in tail:
            if item in ('add_global', 'add_local') and (alive is not None or cur != P):
                raise NotImplementedError('%s after a drop / another add in one corrupt_type list' % item)
            if item == 'add_global':                                # int(P (level + 1) 0.1) ball points (:830-841)
                count = (P * (rng.integers(0, 5, B) + 1) * 0.1).astype(np.int32)
                if getattr(self, '_ball_affine', None) is None:       # (radius, cos theta, phi) ranges of the three uniforms
                    self._ball_affine = (_dev(np.array([1.0, 2.0, 2.0 * math.pi], np.float32), dev),
                                         _dev(np.array([0.0, 1.0, 0.0], np.float32), dev))
                pipeline_add_global(y, P, _dev(count, dev), torch.rand((B, n_add, 3), device=dev, generator=self.gen)
                                    * self._ball_affine[0] - self._ball_affine[1])
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
            elif item in _DROPOUT_LOCAL:
                if cur != P or stride != P or alive is not None:
                    raise NotImplementedError('%s after an add / another drop in one corrupt_type list' % item)
                alive = dropout_local(y, *draw_dropout_local(rng, B, P, item)).to(torch.uint8)
            elif item in ('scale', 'scale_single'):
                # 'scale' U(.5,2)^3 (:59-69); 'scale_single' ONE factor U(1/s, s), s by level (:71-80); both re-normalised
                if cur != P or stride != P or alive is not None:
                    raise NotImplementedError('%s after an add / a drop in one corrupt_type list' % item)
                if item == 'scale':
                    ms = [[draw_affine_map(rng, 'scale_nonorm')] for _ in range(B)]
                else:
                    ms = []
                    for b in range(B):
                        s_ = _SCALE_SINGLE[int(rng.integers(0, 5))]
                        ms.append([(np.eye(3) * rng.uniform(1.0 / s_, s_), np.zeros(3))])
                y = pipeline_norm_affine(pipeline_norm_affine(y, False, ms), True)
            else:
                raise NotImplementedError(item)
        if cur != P:                                                # rows behind P + added[b] are not part of cloud b
            if alive is None:
                alive = torch.ones((B, stride), dtype=torch.uint8, device=dev)
            alive &= (torch.arange(stride, device=dev).view(1, -1) < (P + _dev(added, dev)).view(-1, 1)).to(torch.uint8)
        corrupted = pipeline_subset(y, cur, self.npoints, torch.rand((B, stride), device=dev, generator=self.gen), alive)
        return corrupted, clean

    def set_epoch(self, epoch):
        """DistributedSampler.set_epoch: the next iteration draws epoch `epoch`'s shared permutation (resume-safe)."""
        self._epoch, self._epoch_set = int(epoch), True

    def __iter__(self):
        if self._clouds is None:
            self._materialise()
        n = self._clouds.shape[0]
        if self.listed and self.world > 1:
            # DistributedSampler semantics: ONE permutation per epoch shared by the ranks (seeded without the
            # rank), padded by wrapping to a multiple of world, rank r takes every world-th index
            # `shared_seed` (config; the runner passes args.seed) is the rank-independent base; without it the constructor's
            # seed is taken to be base + rank (main.py:81 seeds rank r with seed + r).  set_epoch(e) names the epoch
            # (the reference calls sampler.set_epoch(epoch), runner_pretrain.py:115); otherwise epochs count from 0.
            if getattr(self, '_epoch_set', None) is None:
                self._epoch = getattr(self, '_epoch', -1) + 1
            self._epoch_set = None
            base = self.shared_seed if self.shared_seed is not None else self.seed - self.rank
            shared = np.random.default_rng([base, self._epoch]).permutation(n)
            total = -(-n // self.world) * self.world
            order = np.resize(shared, total)[self.rank::self.world]
            n = order.shape[0]
        else:
            order = self.rng.permutation(n)
        # The pipeline's launches sit in front of the step on the same stream; what matters is that the HOST never waits:
        # every small parameter tensor goes through pinned staging (_dev), so the host runs ahead of the GPU and the
        # pipeline's few small kernels are the only cost (measured end to end through main.py, 300-step epochs: 10 030-10 060
        # clouds/s = the bare step's rate; with pageable copies the host waited for the previous step at each of them:
        # 9 500.  Producing the next batch on a side stream as well changed nothing.)
        dev = self._clouds.device
        for i in range(self.steps):
            sel = order[(np.arange(self.bs) + i * self.bs) % order.shape[0]]
            corrupted, clean = self.batch(_dev(sel, dev))
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
