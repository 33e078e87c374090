"""Pins oracle_dropout_local against the LIVE reference (runs only where /root/reference exists).

The reference's corrupt_dropout_local (datasets/corrupt_util.py:590-612) draws inside the function;
here its draws are captured while it runs (cluster sizes, and the seed = first point of each internal
shuffle), translated into the oracle's inputs (seed rank among the survivors in index order) and the
oracle's survivor set is required to equal the reference's output cloud as a SET of points.  The
fixture stores inputs, draws and the expected survivor mask: data only.

    python tests/golden/make_pipeline_fixtures.py
"""
import importlib.util
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ops as O  # noqa: E402

REF = '/root/reference/datasets/corrupt_util.py'


def load_reference():
    import types
    # the module imports two CUDA-only third-party packages at its line 897-898 (used by other functions)
    for name, attrs in (('knn_cuda', {'KNN': (lambda **kw: None)}), ('pointnet2_ops', {}), ('pointnet2_ops.pointnet2_utils', {})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
    sys.modules['pointnet2_ops'].pointnet2_utils = sys.modules['pointnet2_ops.pointnet2_utils']
    spec = importlib.util.spec_from_file_location('ref_corrupt_util', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    O.build()
    rng = np.random.default_rng(7)
    B, P = 8, 640
    clouds = rng.uniform(-1, 1, (B, P, 3)).astype(np.float32)
    clouds[1, 5] = clouds[1, 9]                          # duplicated point
    nclusters = np.zeros(B, np.int32)
    seed_rank = np.zeros((B, 8), np.int32)
    sizes = np.zeros((B, 8), np.int32)
    want = np.zeros((B, P), np.uint8)
    for b in range(B):
        np.random.seed(100 + b)
        seeds, ks = [], []
        orig_shuffle, orig_sizes = ref._shuffle_pointcloud, ref._gen_random_cluster_sizes

        def shuffle(pcd):
            out = orig_shuffle(pcd)
            seeds.append(out[0].copy())
            return out

        def gen(n, total):
            out = orig_sizes(n, total)
            ks.extend(int(k) for k in out)
            return out
        ref._shuffle_pointcloud, ref._gen_random_cluster_sizes = shuffle, gen
        try:
            out = ref.corrupt_dropout_local(clouds[b].copy(), 0)
        finally:
            ref._shuffle_pointcloud, ref._gen_random_cluster_sizes = orig_shuffle, orig_sizes
        nc = len(ks)
        nclusters[b] = nc
        sizes[b, :nc] = ks
        # seed coordinates -> rank among the survivors (oracle state before that cluster)
        for c in range(nc):
            alive = O.dropout_local(clouds[b:b + 1], np.array([c], np.int32), seed_rank[b:b + 1], sizes[b:b + 1])[0]
            hits = np.flatnonzero(alive.astype(bool) & (clouds[b] == seeds[c]).all(1))
            assert len(hits) >= 1, (b, c)
            seed_rank[b, c] = int(alive[:hits[0]].sum())
        alive = O.dropout_local(clouds[b:b + 1], nclusters[b:b + 1], seed_rank[b:b + 1], sizes[b:b + 1])[0]
        got = clouds[b][alive.astype(bool)]
        key = lambda a: a[np.lexsort(a.T[::-1])]
        assert got.shape == out.shape and np.array_equal(key(got), key(out.astype(np.float32))), (b, got.shape, out.shape)
        want[b] = alive
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'dropout_local_ref.npz'), clouds=clouds,
                        nclusters=nclusters, seed_rank=seed_rank, sizes=sizes, alive=want)
    print('dropout_local: oracle == live reference on %d clouds; clusters %s' % (B, nclusters.tolist()))


if __name__ == '__main__':
    main()
