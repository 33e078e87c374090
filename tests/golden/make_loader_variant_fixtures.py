"""Golden record of EVERY entry of the reference's loader-side corruption table.

Runs only where /root/reference exists.  For each name of `datasets.corrupt_util.corruptions` (:984-1038) and each
level 0..4 the live function is called on one seeded cloud while every draw it makes from `np.random` / `random` is
recorded WITH ITS ARGUMENTS (low / high / size ...): that pins the parameter ranges the product's tables
(point_dae_amd/datasets.py _MAP_PARAMS, _JITTER, _DROPOUT_LOCAL, _SCALE_SINGLE) must carry.  For the affine maps the
drawn values and the function's output are stored too (the product's kernel must reproduce the output from the draw).
The set dispatchers 'affine_r3' / 'affine_r5' / 'affine_r3_v2' / 'affine_r5_v2' of corrupt_data (:1046-1096) are
recorded the same way (which pool, how many maps).  Data only: names, numbers, arrays.

    python tests/golden/make_loader_variant_fixtures.py
"""
import json
import os
import random
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_loader_fixtures as F  # noqa: E402

P = 128


class ArgRecorder(F.Recorder):
    def _wrap(self, tag, fn):
        def inner(*a, **k):
            out = fn(*a, **k)
            rec = a[0].copy() if tag == 'np.shuffle' else (np.copy(out) if isinstance(out, np.ndarray) else out)
            self.log.append((tag, rec, a, k))
            return out
        return inner


def plain(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [plain(x) for x in v]
    return v


def call_args(log):
    """[(tag, positional args, keyword args)] with arrays as lists; the (P, 1) shuffle / noise shapes stay as shapes."""
    out = []
    for tag, rec, a, k in log:
        if tag == 'np.shuffle':
            out.append([tag, [], {}])
        else:
            out.append([tag, plain(list(a)), {kk: plain(vv) for kk, vv in k.items()}])
    return out


def main():
    F.load_reference_dataset()
    import importlib
    cu = importlib.import_module('datasets.corrupt_util')
    from point_dae_amd.synthetic import shapenet_like_clouds
    cloud = shapenet_like_clouds(1, P, seed=77)[0].astype(np.float32)
    meta, arrays = {}, {'cloud': cloud}
    for name, fn in cu.corruptions.items():
        meta[name] = {}
        for level in range(5):
            random.seed(300 + level), np.random.seed(400 + level)
            with ArgRecorder() as r:
                out = fn(cloud.copy(), level)
            meta[name][str(level)] = {'calls': call_args(r.log), 'out_rows': int(out.shape[0]), 'out_dtype': str(out.dtype)}
            if name.startswith('jitter'):                     # out = cloud + sigma * noise: the sigma the function used
                noise = r.log[0][1]
                meta[name][str(level)]['sigma'] = float(np.median((np.asarray(out, np.float64) - cloud) / noise))
            first = r.log[0] if r.log else None
            if first is not None and out.shape == cloud.shape and first[0] in ('np.uniform', 'np.choice') and name != 'add_global':
                arrays['%s/%d/draw' % (name, level)] = np.asarray(first[1], np.float64)
                arrays['%s/%d/out' % (name, level)] = np.asarray(out, np.float64)
    sets = {}
    for item in ('affine_r3', 'affine_r5', 'affine_r3_v2', 'affine_r5_v2'):
        pools, numbers, ks = set(), set(), set()
        for seed in range(40):
            random.seed(seed), np.random.seed(seed)
            with ArgRecorder() as r:
                cu.corrupt_data(cloud.copy(), [item])
            tags = [(t, a) for t, _, a, _ in r.log]
            assert tags[0][0] == 'py.choice' and tags[1][0] == 'py.sample'
            numbers.add(tuple(tags[0][1][0]))
            pools.add(tuple(tags[1][1][0]))
            ks.add(int(tags[1][1][1]))
        assert len(pools) == 1 and len(numbers) == 1
        sets[item] = {'pool': list(pools.pop()), 'numbers': list(numbers.pop()), 'seen_k': sorted(ks)}
    # names the reference's YAMLs use that its dispatcher does not know
    unknown = [n for n in ('affine_r3_tiny', 'affine_r3_middle', 'scan') if n not in cu.corruptions]
    arrays['meta'] = np.array(json.dumps({'table': meta, 'sets': sets, 'unknown_upstream': unknown,
                                          'pass_through': ['clean', 'dropout_patch_pointmae', 'dropout_global*']}))
    path = os.path.join(HERE, 'loader_variants_ref.npz')
    np.savez_compressed(path, **arrays)
    print('recorded %d table entries x 5 levels, %d sets; unknown upstream: %s; wrote %s (%d KiB)' % (
        len(meta), len(sets), unknown, path, os.path.getsize(path) // 1024))


if __name__ == '__main__':
    main()
