"""Generates tests/golden/ops_oracle.npz: op-level golden vectors (SURVEY.md §8c list, items 1-5) from
the CPU oracle (oracle/pdae_oracle.c), which restates the reference kernels line by line.

    python tests/golden/make_op_fixtures.py

Inputs are regenerated from seeds (tests/conftest.make_clouds), so the file holds only the
expected outputs: indices in full, float outputs as float32.  The reference itself has no vectors
for these ops (its native code is CUDA-only); these pin the oracle -- and through
tests/test_gpu_ops.py::test_op_golden_vectors the HIP kernels -- against drift.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from conftest import make_clouds          # noqa: E402


def special_clouds():
    """(2,1024,3): cloud 0 has 40 points inside the |p|^2 <= 1e-3 sphere that FPS skips, cloud 1
    has every point duplicated (distance ties)."""
    x = make_clouds(5, 2, 1024, 'shapes')
    x[0, 100:140] *= 0.01
    x[1, 512:] = x[1, :512]
    return x


def cases(ops):
    """`ops`: an object with the oracle's call signatures (oracle.ops, or the HIP adapter of
    tests/test_gpu_ops.py) -> {name: array}."""
    out = {}
    for tag, (B, N, m) in {'fps_1024_64': (8, 1024, 64), 'fps_1024_512': (2, 1024, 512),
                           'fps_512_128': (2, 512, 128), 'fps_2048_128': (2, 2048, 128)}.items():
        idx, ctr = ops.furthest_point_sample(make_clouds(31, B, N, 'shapes'), m, return_centres=True)
        out[tag + '_idx'], out[tag + '_ctr'] = idx, ctr
    out['fps_special_idx'] = ops.furthest_point_sample(special_clouds(), 64)
    for tag, (B, N, G, k) in {'knn_1024_64_32': (4, 1024, 64, 32), 'knn_2048_128_32': (2, 2048, 128, 32)}.items():
        x = make_clouds(32, B, N, 'shapes')
        ctr = ops.furthest_point_sample(x, G, return_centres=True)[1]
        dist, idx = ops.knn(x, ctr, k)
        out[tag + '_idx'], out[tag + '_dist'] = idx, dist
    for tag, (N, m, ns, rad) in {'ball_512_32': (1024, 512, 32, 0.2), 'ball_128_64': (512, 128, 64, 0.4)}.items():
        x = make_clouds(33, 2, N, 'shapes')
        ctr = ops.furthest_point_sample(x, m, return_centres=True)[1]
        ctr[0, 0] = 5.0                                     # an empty ball
        out[tag + '_idx'] = ops.ball_query(rad, ns, x, ctr)
    for tag, (B, n, m) in {'cd_32_32': (64, 32, 32), 'cd_1024_1024': (2, 1024, 1024), 'cd_16384_1024': (1, 16384, 1024)}.items():
        a, b = make_clouds(34, B, n, 'uniform'), make_clouds(35, B, m, 'uniform')
        d1, d2, i1, i2 = ops.chamfer_forward(a, b)
        out[tag + '_d1'], out[tag + '_d2'], out[tag + '_i1'], out[tag + '_i2'] = d1, d2, i1, i2
        g1 = np.random.default_rng(36).standard_normal(d1.shape).astype(np.float32)
        g2 = np.random.default_rng(37).standard_normal(d2.shape).astype(np.float32)
        ga, gb = ops.chamfer_backward(a, b, i1, i2, g1, g2)
        if n <= 1024:
            out[tag + '_ga'], out[tag + '_gb'] = ga, gb
        out[tag + '_l2'] = np.float32(ops.chamfer_distance_l2(a, b))
        out[tag + '_l1'] = np.float32(ops.chamfer_distance_l1(a, b))
    two_a = np.array([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]], np.float32)   # extensions/emd/test_emd_loss.py:7-44:
    two_b = np.array([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]], np.float32)   # optimum 0.71 per cloud, /n = 0.355
    out['emd_two_point_cost'] = np.float32(ops.earth_mover_distance(two_a, two_b))
    for tag, n in {'emd_32': 32, 'emd_256': 256}.items():
        a, b = make_clouds(38, 2, n, 'uniform'), make_clouds(39, 2, n, 'uniform')
        match = ops.emd_approxmatch(a, b)
        out[tag + '_match_rowsum'] = match.sum(-1).astype(np.float32)
        out[tag + '_cost'] = ops.emd_matchcost(a, b, match)
    return out


if __name__ == '__main__':
    from oracle import ops
    ops.build()
    data = cases(ops)
    path = os.path.join(HERE, 'ops_oracle.npz')
    np.savez_compressed(path, **data)
    print('wrote', path, os.path.getsize(path), 'bytes;', len(data), 'arrays')
