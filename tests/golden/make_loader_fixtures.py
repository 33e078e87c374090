"""Golden vectors of the loader-side pipeline from the LIVE reference data set class.

Runs only where /root/reference exists.  Builds a tiny ShapeNet-55 layout on disk (8 seeded .npy clouds + train.txt),
instantiates the reference's own `datasets.ShapeNet55Dataset.ShapeNet` on it and calls its `__getitem__`
(:90-119: IO -> augment_data 'norm' -> random_sample -> corrupt_data -> random_sample) once per configuration and cloud,
while every draw it makes from `np.random` / `random` is recorded.  The recorded draws are turned into the inputs of
oracle/pipeline.py (affine maps, jitter noise, ball uniforms, cluster seeds / sigmas / noise, viewpoint and gate,
sub-sampling permutations); the oracle must then reproduce the reference's (corrupted, clean) item.  The fixture
stores the input clouds, those draws and the reference's outputs: data only.

    python tests/golden/make_loader_fixtures.py
"""
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_import as R  # noqa: E402
from oracle import ops as O  # noqa: E402
from oracle import pipeline as OP  # noqa: E402

P, NPTS, B = 2048, 1024, 8
CONFIGS = {                                              # corrupt_type lists that ship in the reference's cfgs/
    'affine_r3': ['affine_r3'],
    'jitter': ['jitter'],
    'affine_r3_jitter': ['affine_r3', 'jitter'],
    'add_global': ['add_global'],
    'add_local': ['add_local'],
    'nonuniform_density': ['nonuniform_density'],
    'affine_r3_dropout_local': ['affine_r3', 'dropout_local'],
}
AFFINE = ['translate', 'scale_nonorm', 'rotate', 'reflection', 'shear']


class Recorder:
    """Records every draw the reference makes from the global generators, in order."""
    NP = ('uniform', 'randn', 'normal', 'rand', 'randint', 'choice', 'shuffle')
    PY = ('choice', 'sample', 'random')

    def __enter__(self):
        self.log, self._np, self._py = [], {}, {}
        for name in self.NP:
            self._np[name] = getattr(np.random, name)
            setattr(np.random, name, self._wrap('np.' + name, self._np[name]))
        for name in self.PY:
            self._py[name] = getattr(random, name)
            setattr(random, name, self._wrap('py.' + name, self._py[name]))
        return self

    def _wrap(self, tag, fn):
        def inner(*a, **k):
            out = fn(*a, **k)
            rec = a[0].copy() if tag == 'np.shuffle' else (np.copy(out) if isinstance(out, np.ndarray) else out)
            self.log.append((tag, rec))
            return out
        return inner

    def __exit__(self, *exc):
        for name, fn in self._np.items():
            setattr(np.random, name, fn)
        for name, fn in self._py.items():
            setattr(random, name, fn)


def load_reference_dataset():
    R.setup()
    for name, attrs in (('h5py', {}), ('torchvision', {}), ('torchvision.transforms', {})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    import importlib
    return importlib.import_module('datasets.ShapeNet55Dataset')


class Draws:
    def __init__(self, log):
        self.log, self.i = log, 0

    def take(self, tag):
        t, v = self.log[self.i]
        assert t == tag, (self.i, t, tag)
        self.i += 1
        return v


def replay(cloud, corrupt_type, log):
    """Walk the recorded draws in the order __getitem__ makes them -> structured inputs + the oracle's item."""
    d = Draws(log)
    rec = {}
    data = OP.pc_normalize(cloud.astype(np.float32)).astype(np.float32)     # augment_data(['norm']) into data[:, :3]
    rec['perm_clean'] = d.take('np.shuffle')
    clean = OP.random_sample(data, NPTS, rec['perm_clean'])
    pc = data
    src = np.arange(P)             # row j of the reference's current array = point src[j] of the DEVICE-order array
                                   # (original order, added points appended): the reference shuffles / re-sorts rows
    maps = []                                                                # affine maps as (3x3 M, 3 t): y = x M + t
    for item in corrupt_type:
        if item == 'affine_r3':
            number = d.take('py.choice')
            names = d.take('py.sample')
            assert len(names) == number
            steps = []
            for name in names:
                d.take('py.choice')                                          # the level, unused by these maps
                if name == 'translate':
                    v = d.take('np.uniform')
                    steps.append(('translate', v)), maps.append((np.eye(3), v))
                elif name == 'scale_nonorm':
                    v = d.take('np.uniform')
                    steps.append(('scale_nonorm', v)), maps.append((np.diag(v), np.zeros(3)))
                elif name == 'rotate':
                    Rm = OP.rotation_matrix(d.take('np.uniform'))
                    steps.append(('matrix', Rm)), maps.append((Rm, np.zeros(3)))
                elif name == 'reflection':
                    Rm = OP.reflection_matrix(d.take('np.choice'))
                    steps.append(('matrix', Rm)), maps.append((Rm, np.zeros(3)))
                elif name == 'shear':
                    Rm = OP.shear_matrix(d.take('np.uniform'))
                    steps.append(('matrix', Rm)), maps.append((Rm, np.zeros(3)))
            pc = OP.affine_steps(pc, steps)
        elif item == 'jitter':
            rec['jitter_level'] = d.take('py.choice')
            rec['jitter_noise'] = d.take('np.randn')
            pc = OP.jitter(pc, rec['jitter_level'], rec['jitter_noise'])
        elif item == 'add_global':
            rec['add_level'] = d.take('py.choice')
            u = [d.take('np.uniform') for _ in range(3)]
            rec['ball_u'] = np.concatenate(u, axis=1)                        # (n, 3): radius, cos(theta), phi draws
            pc = OP.add_global(pc, rec['add_level'], *u)
            src = np.concatenate([src, np.arange(len(src), pc.shape[0])])
        elif item == 'add_local':
            rec['add_level'] = d.take('py.choice')
            ncl = d.take('np.randint')
            labels = d.take('np.randint')
            sizes = [int((labels == i).sum()) for i in range(ncl)]
            order = d.take('np.rand').argsort(axis=0)[:, 0]                  # _shuffle_pointcloud :25-26
            sigmas, noise = [], []
            for k in sizes:
                sigmas.append(d.take('np.uniform'))
                noise.append(d.take('np.randn'))
            rec['local_order'], rec['local_sizes'] = order, np.array(sizes)
            rec['local_sigmas'], rec['local_noise'] = np.array(sigmas), np.concatenate(noise, axis=0)
            pc = OP.add_local(pc, rec['add_level'], order, sizes, sigmas, rec['local_noise'])
            src = np.concatenate([src[order], np.arange(len(src), pc.shape[0])])
        elif item == 'nonuniform_density':
            rec['density_level'] = d.take('py.choice')
            rec['density_v'] = d.take('np.normal')
            rec['density_r'] = d.take('np.uniform')
            keep = OP.density_keep(pc, rec['density_level'], rec['density_v'], rec['density_r'])
            rec['density_keep'] = keep
            pc, src = pc[keep], src[keep]
        elif item == 'dropout_local':
            d.take('py.choice')
            ratio = d.take('np.uniform')[0]
            total = int(pc.shape[0] * ratio)
            ncl = d.take('np.randint')
            labels = d.take('np.randint')
            sizes = [int((labels == i).sum()) for i in range(ncl)]
            assert sum(sizes) == total
            # per cluster one _shuffle_pointcloud: the seed is the first point of the shuffled SURVIVORS
            alive = np.ones(pc.shape[0], bool)
            pcf = np.ascontiguousarray(pc.astype(np.float32))
            ranks = []
            cur = np.arange(pc.shape[0])                                     # indices of the survivors, reference order
            for k in sizes:
                sh = d.take('np.rand').argsort(axis=0)[:, 0]
                cur = cur[sh]
                seed_idx = cur[0]
                ranks.append(int(alive[:seed_idx].sum()))
                dist = np.sum((pc[cur] - pc[cur][:1, :]) ** 2, axis=1, keepdims=True)
                idx = dist.argsort(axis=0)[::-1, 0]
                cur = cur[idx][:len(cur) - k]
                alive[:] = False
                alive[cur] = True
            rec['dl_nclusters'] = np.array([ncl], np.int32)
            rec['dl_rank'] = np.zeros((1, 8), np.int32)
            rec['dl_rank'][0, :ncl] = ranks
            rec['dl_sizes'] = np.zeros((1, 8), np.int32)
            rec['dl_sizes'][0, :ncl] = sizes
            got = O.dropout_local(pcf[None], rec['dl_nclusters'], rec['dl_rank'], rec['dl_sizes'])[0].astype(bool)
            assert np.array_equal(got, alive), 'C oracle dropout_local disagrees with the replayed reference'
            rec['dl_alive'] = alive
            rec['dl_input'] = pcf                 # the fp32 cloud the drop ran on (bit-exact masks need identical inputs)
            pc, src = pc[cur], src[cur]          # the reference's array is left in its last distance-sorted order
        else:
            raise KeyError(item)
    rec['n_before_sample'] = pc.shape[0]
    refill = d.take('np.choice') if pc.shape[0] < NPTS else None
    rec['perm_corrupt'] = d.take('np.shuffle')
    corrupted = OP.random_sample(pc, NPTS, rec['perm_corrupt'], refill)
    assert d.i == len(log), (d.i, len(log))
    # which device-order points the reference picked, in output order (what the sub-sampling keys must encode)
    ext = src if refill is None else np.concatenate([src, src[refill]])
    rec['select'] = ext[rec['perm_corrupt'][:NPTS]].astype(np.int32)
    rec['select_clean'] = rec['perm_clean'][:NPTS].astype(np.int32)
    if maps:
        rec['maps'] = np.stack([np.concatenate([M.reshape(-1), t]) for M, t in maps])      # (n, 12)
    return rec, clean.astype(np.float32), corrupted.astype(np.float32)


def main():
    ds_mod = load_reference_dataset()
    from easydict import EasyDict
    O.build()
    from point_dae_amd.synthetic import shapenet_like_clouds
    clouds = (shapenet_like_clouds(B, P, seed=41) * 0.7 + 0.1).astype(np.float32)   # off-centre, not unit: 'norm' has work
    clouds[2, 11] = clouds[2, 4]                                                    # a duplicated point
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'pc'))
    names = []
    for b in range(B):
        names.append('%08d-model%02d.npy' % (2691156 + b, b))
        np.save(os.path.join(tmp, 'pc', names[-1]), clouds[b])
    open(os.path.join(tmp, 'train.txt'), 'w').write('\n'.join(names) + '\n')
    open(os.path.join(tmp, 'test.txt'), 'w').write(names[0] + '\n')
    out = {'clouds': clouds}
    for tag, corrupt_type in CONFIGS.items():
        cfg = EasyDict(DATA_PATH=tmp, PC_PATH=os.path.join(tmp, 'pc'), subset='train', N_POINTS=P, npoints=NPTS,
                       aug_type=['norm'], corrupt_type=corrupt_type)
        ds = ds_mod.ShapeNet(cfg)
        assert len(ds) == B
        for b in range(B):
            random.seed(1000 + b), np.random.seed(2000 + b)
            with Recorder() as r:
                tax, mid, corrupted, clean = ds[b]
            rec, o_clean, o_corrupted = replay(clouds[b], corrupt_type, r.log)
            # the oracle restatement reproduces the live item (fp64 chains rounded to fp32 at the end: exact or 1 ulp)
            assert np.array_equal(o_clean, clean.numpy()), (tag, b)
            err = np.abs(o_corrupted - corrupted.numpy()).max()
            assert err <= 1e-6, (tag, b, err)
            out['%s/%d/clean' % (tag, b)] = clean.numpy()
            out['%s/%d/corrupted' % (tag, b)] = corrupted.numpy()
            for k, v in rec.items():
                v = np.asarray(v)
                if v.dtype == np.float64 and k in ('jitter_noise', 'local_noise', 'density_r', 'ball_u'):
                    v = v.astype(np.float32)                 # the device pipeline consumes fp32 draws; so does the test's oracle
                if k.startswith('perm'):
                    continue                                 # `select` / `select_clean` carry what matters of them
                out['%s/%d/%s' % (tag, b, k)] = v
        print('%-24s oracle == live ShapeNet.__getitem__ on %d clouds' % (tag, B))
    path = os.path.join(HERE, 'loader_pipeline_ref.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


if __name__ == '__main__':
    main()
