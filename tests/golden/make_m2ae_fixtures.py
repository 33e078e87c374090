"""Golden vectors for the Point-M2AE hierarchical grouping from the LIVE reference.

Runs only where /root/reference exists.  Imports models/Point_M2AE_modules.py in place (tests/golden/ref_import.py:
knn_cuda / pointnet2_ops stubs backed by the C oracle, as for every other fixture), runs its `Group` three levels
deep exactly as Point_M2AE.forward does (models/Point_M2AE.py:245-263) and the multi-scale masking lines of
H_Encoder.forward (:107-121) -- executed from the reference's own source text, extracted at run time, never stored --
on seeded clouds with an injected top-level mask.  Stores inputs and expected outputs only.

    python tests/golden/make_m2ae_fixtures.py
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import ref_import  # noqa: E402
from oracle import ops as O  # noqa: E402

NUM_GROUPS, GROUP_SIZES = [512, 256, 64], [16, 8, 8]     # cfgs/pretrain_PointM2AE_transformer_dropout_patch_affine_r3.yaml


def reference_masking(centers, idxs, top):
    """Run the reference's own multi-scale masking statements (Point_M2AE.py:107-121) on CPU tensors."""
    src = open(os.path.join(ref_import.REF, 'models', 'Point_M2AE.py')).read().split('\n')
    start = next(i for i, ln in enumerate(src) if 'Multi-scale Masking by back-propagation' in ln)
    end = next(i for i, ln in enumerate(src) if i > start and 'bool_masked_pos.reverse()' in ln)
    body = '\n'.join(ln[8:] for ln in src[start:end + 1])          # the loop + the reverse, de-indented
    env = {'torch': torch, 'neighborhoods': [None] * len(centers), 'centers': centers, 'idxs': idxs,
           'bool_masked_pos': [top]}
    env['neighborhoods'] = [torch.zeros(c.shape[0], c.shape[1], idxs[i].numel() // (c.shape[0] * c.shape[1]), 3)
                            for i, c in enumerate(centers)]
    exec(compile(body, 'Point_M2AE.py:107-121', 'exec'), env)
    return env['bool_masked_pos']


def main():
    ref_import.setup()
    ref_import.cpu_cuda_noop()
    torch.Tensor.cuda = lambda self, *a, **k: self               # the masking lines call .cuda() on fresh tensors
    # Point_M2AE_modules imports `from utils import misc` / logger etc.: the reference's own utils package
    import importlib
    M = importlib.import_module('models.Point_M2AE_modules')
    from point_dae_amd.synthetic import shapenet_like_clouds
    B, N = 2, 2048
    pts = shapenet_like_clouds(B, N, seed=21)
    pts[1, 7] = pts[1, 3]                                         # a duplicated point (kNN / FPS ties)
    x = torch.from_numpy(pts)
    dividers = [M.Group(num_group=g, group_size=k) for g, k in zip(NUM_GROUPS, GROUP_SIZES)]
    neighborhoods, centers, idxs = [], [], []
    for i, d in enumerate(dividers):                              # Point_M2AE.forward :247-263
        nb, c, idx = d(x if i == 0 else center)
        center = c
        neighborhoods.append(nb), centers.append(c), idxs.append(idx)
    rng = np.random.default_rng(5)
    top = np.zeros((B, NUM_GROUPS[-1]), dtype=bool)
    for b in range(B):
        top[b, rng.permutation(NUM_GROUPS[-1])[:int(0.8 * NUM_GROUPS[-1])]] = True       # mask_ratio 0.8
    masks = reference_masking(centers, idxs, torch.from_numpy(top))
    # the oracle restatement must reproduce the live reference exactly
    o_nb, o_c, o_idx = O.m2ae_hierarchy(pts, NUM_GROUPS, GROUP_SIZES)
    o_masks = O.m2ae_multi_scale_mask(top, o_idx, o_c)
    out = {'pts': pts, 'top_mask': top}
    for i in range(3):
        assert np.array_equal(o_idx[i], idxs[i].numpy()), i
        assert np.array_equal(o_c[i], centers[i].numpy()), i
        assert np.array_equal(o_nb[i], neighborhoods[i].numpy()), i
        assert np.array_equal(o_masks[i], masks[i].numpy()), i
        out['idx%d' % i] = idxs[i].numpy().astype(np.int32)        # < 2^31: stored narrow
        out['center%d' % i] = centers[i].numpy()
        out['nb%d_sample' % i] = neighborhoods[i].numpy()[:, ::37]  # a strided sample; idx + centre pin the rest
        out['mask%d' % i] = masks[i].numpy()
    # token merging (:132) on a seeded feature table
    feat = rng.standard_normal((B, NUM_GROUPS[0], 8)).astype(np.float32)
    merged = torch.from_numpy(feat).reshape(B * NUM_GROUPS[0], -1)[idxs[1], :].reshape(B, NUM_GROUPS[1], GROUP_SIZES[1], -1)
    out['feat'], out['merged'] = feat, merged.numpy()
    path = os.path.join(HERE, 'm2ae_grouping_b2.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes; masked counts',
          [int(m.sum()) for m in masks], 'of', [m.numel() for m in masks])


if __name__ == '__main__':
    main()
