"""Loader-side pipeline: the dropout_local oracle against the fixture generated from the live reference,
the host-side draws, and (GPU) the HIP kernel against the oracle and the dataset end to end."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'dropout_local_ref.npz'))


def test_dropout_local_oracle_reproduces_reference_fixture(oracle_ops):
    fx = _fixture()
    alive = oracle_ops.dropout_local(fx['clouds'], fx['nclusters'], fx['seed_rank'], fx['sizes'])
    assert np.array_equal(alive, fx['alive'])
    # size bookkeeping: survivors = P - dropped
    assert np.array_equal(alive.sum(1), fx['clouds'].shape[1] - fx['sizes'].sum(1))


def test_draws_follow_the_reference_distributions():
    from point_dae_amd.datasets import draw_affine_r3, draw_dropout_local
    rng = np.random.default_rng(0)
    ncl, rank, sizes = draw_dropout_local(rng, 512, 8192)
    assert ncl.min() >= 1 and ncl.max() <= 7
    tot = sizes.sum(1)
    assert tot.min() >= int(0.1 * 8192) - 1 and tot.max() <= int(0.5 * 8192)
    alive_before = 8192 - np.concatenate([np.zeros((512, 1), np.int64), sizes.cumsum(1)[:, :-1]], 1)
    used = np.arange(8)[None, :] < ncl[:, None]
    assert (rank[used] >= 0).all() and (rank[used] < alive_before[used]).all()
    A, t = draw_affine_r3(rng, 256)
    assert A.shape == (256, 3, 3) and t.shape == (256, 3)
    assert np.abs(np.linalg.det(A)).min() > 1e-3          # every map is invertible
    assert (np.abs(t) <= 3.0).all()


def test_unimplemented_loader_corruption_is_refused():
    """A corrupt_type the device pipeline does not implement must raise, never train on clean == corrupted."""
    from point_dae_amd.datasets import ShapeNet
    for name in ('scan', 'affine_r3_tiny', 'affine_r3_middle', 'no_such_corruption'):
        with pytest.raises(NotImplementedError):
            ShapeNet({'corrupt_type': [name], 'device': 'cpu'})
    with pytest.raises(NotImplementedError):
        ShapeNet({'aug_type': ['jitter'], 'device': 'cpu'})


def test_single_map_draws():
    """the parameter ranges of corrupt_util.py's single maps and of the scale / translate augmentations"""
    from point_dae_amd.datasets import draw_affine, sphere_points
    rng = np.random.default_rng(1)
    for name, check in [
            ('aug_scale', lambda A, t: (np.diagonal(A, axis1=1, axis2=2) >= 2 / 3 - 1e-6).all() and (np.diagonal(A, axis1=1, axis2=2) <= 1.5 + 1e-6).all() and (t == 0).all()),
            ('aug_translate', lambda A, t: (np.abs(t) <= 0.2 + 1e-6).all() and np.allclose(A, np.eye(3))),
            ('translate', lambda A, t: (np.abs(t) <= 0.5 + 1e-6).all() and np.abs(t).max() > 0.2),
            ('scale_nonorm', lambda A, t: (np.diagonal(A, axis1=1, axis2=2) >= 0.5 - 1e-6).all() and (np.diagonal(A, axis1=1, axis2=2) <= 2 + 1e-6).all()),
            ('rotate', lambda A, t: np.allclose(A @ A.transpose(0, 2, 1), np.eye(3), atol=1e-5) and np.allclose(np.linalg.det(A), 1, atol=1e-5)),
            ('rotate_z', lambda A, t: np.allclose(A[:, 2, 2], 1) and np.allclose(A @ A.transpose(0, 2, 1), np.eye(3), atol=1e-5)),
            ('reflection', lambda A, t: np.allclose(np.abs(A), np.eye(3))),
            ('shear', lambda A, t: np.allclose(np.diagonal(A, axis1=1, axis2=2), 1) and np.abs(A - np.eye(3)).max() <= 0.5 + 1e-6)]:
        A, t = draw_affine(rng, 200, lambda r, n=name: [n])
        assert check(A, t), name
    pts = sphere_points(rng, 4, 2000)
    r = np.linalg.norm(pts, axis=2)
    assert r.max() <= 1 + 1e-6 and abs((r < 0.5 ** (1 / 3)).mean() - 0.5) < 0.05        # uniform in the ball


@pytest.mark.gpu
def test_dropout_local_kernel_matches_oracle(oracle_ops):
    import torch
    from point_dae_amd.datasets import draw_dropout_local, dropout_local
    fx = _fixture()
    got = dropout_local(torch.from_numpy(fx['clouds']).cuda(), fx['nclusters'], fx['seed_rank'], fx['sizes'])
    assert np.array_equal(got.cpu().numpy().astype(np.uint8), fx['alive'])
    rng = np.random.default_rng(3)
    for B, P in ((16, 8192), (5, 1000), (3, 2048), (2, 16384), (4, 77)):
        x = rng.uniform(-1, 1, (B, P, 3)).astype(np.float32)
        x[0, 1] = x[0, 0]                                   # ties
        ncl, rank, sizes = draw_dropout_local(rng, B, P)
        got = dropout_local(torch.from_numpy(x).cuda(), ncl, rank, sizes).cpu().numpy().astype(np.uint8)
        assert np.array_equal(got, oracle_ops.dropout_local(x, ncl, rank, sizes)), (B, P)


@pytest.mark.gpu
def test_device_dataset_end_to_end(tmp_path):
    """npy directory in the reference layout -> normalised, corrupted, sub-sampled batches on the device."""
    import torch
    from point_dae_amd.datasets import ShapeNet
    rng = np.random.default_rng(0)
    pc, lists = tmp_path / 'pc', tmp_path / 'lists'
    pc.mkdir(), lists.mkdir()
    names = []
    for i in range(12):
        name = '0269%d-model%d.npy' % (i % 3, i)
        np.save(pc / name, (rng.normal(size=(2048, 6)) * [1, 2, 3, 1, 1, 1] + 5).astype(np.float32))
        names.append(name)
    (lists / 'train.txt').write_text('\n'.join(names))
    ds = ShapeNet({'PC_PATH': str(pc), 'DATA_PATH': str(lists), 'subset': 'train', 'npoints': 512, 'N_POINTS': 2048,
                   'bs': 8, 'steps_per_epoch': 3, 'aug_type': ['norm'], 'corrupt_type': ['affine_r3', 'dropout_local'],
                   'device': 'cuda', 'seed': 1})
    n = 0
    for tax, i, corrupted, clean in ds:
        assert corrupted.shape == (8, 512, 3) and clean.shape == (8, 512, 3) and corrupted.is_cuda
        assert clean.norm(dim=-1).max() <= 1.0 + 1e-5 and clean.norm(dim=-1).max() > 0.5      # unit-sphere normalised
        assert torch.isfinite(corrupted).all() and not torch.equal(corrupted, clean)
        assert tax.startswith('0269')
        n += 1
    assert n == 3
    clean_only = ShapeNet({'npoints': 256, 'N_POINTS': 1024, 'bs': 4, 'steps_per_epoch': 1, 'device': 'cuda',
                           'corrupt_type': ['clean']})
    for _, _, corrupted, clean in clean_only:
        assert torch.equal(corrupted, clean)


@pytest.mark.gpu
@pytest.mark.parametrize('aug,cor', [(['norm', 'scale', 'translate'], ['clean']), (['norm'], ['affine_r3', 'jitter']),
                                     (['norm', 'rotate'], ['dropout_local']), (['norm'], ['add_global']),
                                     (['norm', 'rotate_z'], ['nonuniform_density']), (['norm'], ['shear']),
                                     (['norm'], ['scale']), (['norm', 'translate', 'scale'], ['affine_r3', 'dropout_local']), (['norm'], ['add_local']),
                                     (['norm'], ['affine_r3', 'add_local']), (['norm'], ['rotate', 'nonuniform_density']),
                                     (['scale', 'norm'], ['clean']), (['translate', 'scale', 'rotate', 'scale', 'norm'], ['clean'])])
def test_device_dataset_augmentations_and_corruptions(aug, cor):
    """every loader-side augmentation / corruption the device pipeline implements (the reference's pretrain YAMLs
    use these names: aug ['norm','scale','translate'] in 25 of them) -> finite (B, npoints, 3) batches with the
    property the map promises"""
    import torch
    from point_dae_amd.datasets import ShapeNet
    ds = ShapeNet({'npoints': 512, 'N_POINTS': 2048, 'bs': 6, 'steps_per_epoch': 2, 'device': 'cuda', 'seed': 3,
                   'aug_type': aug, 'corrupt_type': cor})
    for _, _, corrupted, clean in ds:
        assert corrupted.shape == (6, 512, 3) and clean.shape == (6, 512, 3)
        assert torch.isfinite(corrupted).all() and torch.isfinite(clean).all()
        r = clean.norm(dim=-1).amax(dim=1)
        if aug == ['norm']:
            assert (r <= 1 + 1e-5).all()
        if aug[-1] == 'norm':                  # augment_data applies the list in order (corrupt_util.py:1155-1175): a closing
            assert (r <= 1 + 1e-5).all() and (r >= 0.85).all()          # 'norm' leaves the unit sphere whatever came before (r: a 512-point subset)
        elif 'scale' in aug:
            assert (r <= 1.5 * 1.001 + 0.2 * 3 ** 0.5).all() and (r >= 2 / 3 - 0.2 * 3 ** 0.5 - 1e-3).all()
        if 'rotate' in aug or 'rotate_z' in aug:
            assert (r <= 1 + 1e-4).all() and (r >= 0.8).all()                 # rotations keep the unit sphere
        if cor == ['clean']:
            assert torch.equal(corrupted, clean)
        else:
            assert not torch.equal(corrupted, clean)
        if cor == ['scale']:
            assert (corrupted.norm(dim=-1).amax(dim=1) <= 1 + 1e-5).all()       # re-normalised (corrupt_scale)
        if cor == ['add_global']:
            assert (corrupted.norm(dim=-1) <= 1 + 1e-5).all()


@pytest.mark.gpu
def test_device_dataset_refuses_orders_it_does_not_implement():
    """a drop in front of an add (or two adds) would need a compaction pass between them: no shipped configuration of
    the reference lists one; the loader raises instead of silently running something else"""
    from point_dae_amd.datasets import ShapeNet
    ds = ShapeNet({'npoints': 256, 'N_POINTS': 1024, 'bs': 2, 'steps_per_epoch': 1, 'device': 'cuda',
                   'aug_type': ['norm'], 'corrupt_type': ['dropout_local', 'add_global']})
    with pytest.raises(NotImplementedError):
        next(iter(ds))


# ---- the whole loader item against the LIVE reference's ShapeNet.__getitem__ (tests/golden/make_loader_fixtures.py) ----
LOADER_CONFIGS = {
    'affine_r3': ['affine_r3'], 'jitter': ['jitter'], 'affine_r3_jitter': ['affine_r3', 'jitter'],
    'add_global': ['add_global'], 'add_local': ['add_local'], 'nonuniform_density': ['nonuniform_density'],
    'affine_r3_dropout_local': ['affine_r3', 'dropout_local'],
}
NPTS = 1024


def _loader_fixture():
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'loader_pipeline_ref.npz'))


def _rec(fx, tag, b):
    pre = '%s/%d/' % (tag, b)
    return {k[len(pre):]: fx[k] for k in fx.files if k.startswith(pre)}


def _maps_of(rec):
    return [(m[:9].reshape(3, 3), m[9:]) for m in rec['maps']] if 'maps' in rec else []


@pytest.mark.parametrize('tag', sorted(LOADER_CONFIGS))
def test_loader_oracle_reproduces_live_reference_items(tag, oracle_ops):
    """oracle/pipeline.py fed with the recorded draws == what the reference's data set class returned."""
    from oracle import pipeline as OP
    fx = _loader_fixture()
    for b in range(fx['clouds'].shape[0]):
        rec = _rec(fx, tag, b)
        data = OP.pc_normalize(fx['clouds'][b]).astype(np.float32)
        assert np.array_equal(data[rec['select_clean']], rec['clean'])
        pc = data
        for M, t in _maps_of(rec):
            pc = np.dot(pc, M) + t
        if 'jitter_noise' in rec:
            pc = OP.jitter(pc, int(rec['jitter_level']), rec['jitter_noise'])
        if 'ball_u' in rec:
            u = rec['ball_u'].astype(np.float64)
            pc = OP.add_global(pc, int(rec['add_level']), u[:, 0:1], u[:, 1:2], u[:, 2:3])
        if 'local_order' in rec:
            full = OP.add_local(pc, int(rec['add_level']), rec['local_order'], rec['local_sizes'], rec['local_sigmas'],
                                rec['local_noise'])
            pc = np.concatenate([pc, full[pc.shape[0]:]])                       # device order: original rows, then the added
        if 'density_v' in rec:
            keep = OP.density_keep(pc, int(rec['density_level']), rec['density_v'], rec['density_r'])
            assert (keep != rec['density_keep']).sum() == 0
        if 'dl_nclusters' in rec:
            alive = oracle_ops.dropout_local(rec['dl_input'][None], rec['dl_nclusters'], rec['dl_rank'], rec['dl_sizes'])[0]
            assert np.array_equal(alive.astype(bool), rec['dl_alive'])
            assert np.abs(pc - rec['dl_input']).max() <= 2e-6
        got = pc[rec['select']].astype(np.float32)
        assert np.abs(got - rec['corrupted']).max() <= 2e-6, (tag, b)


def _keys_for(select, stride, device):
    import torch
    keys = np.full(stride, 1e9, np.float32)
    keys[select] = np.arange(len(select), dtype=np.float32)
    return torch.from_numpy(keys).to(device)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', sorted(LOADER_CONFIGS))
def test_hip_loader_stages_reproduce_live_reference_items(tag):
    """csrc/pipeline.hip on the recorded draws == the reference's items: norm + affine maps + jitter, add_global,
    add_local, nonuniform_density, dropout_local, and the sub-sampling (keys = the recorded permutation)."""
    import torch
    from point_dae_amd import datasets as D
    fx = _loader_fixture()
    clouds = torch.from_numpy(fx['clouds']).cuda()
    B, P, _ = clouds.shape
    recs = [_rec(fx, tag, b) for b in range(B)]
    dev = clouds.device
    data = D.pipeline_norm_affine(clouds, True)
    keys = torch.stack([_keys_for(r['select_clean'], P, dev) for r in recs])
    clean = D.pipeline_subset(data, P, NPTS, keys)
    want = np.stack([r['clean'] for r in recs])
    assert np.abs(clean.cpu().numpy() - want).max() <= 3e-6
    n_add = max([int(r['corrupted'].shape[0] * 0) + (len(r['ball_u']) if 'ball_u' in r else 0) for r in recs] +
                [len(r['local_noise']) if 'local_noise' in r else 0 for r in recs])
    stride = P + n_add
    sigma = noise = None
    if 'jitter_noise' in recs[0]:
        sigma = torch.tensor([0.01 * (int(r['jitter_level']) + 1) for r in recs], dtype=torch.float32, device=dev)
        noise = torch.from_numpy(np.stack([r['jitter_noise'] for r in recs])).cuda()
    y = D.pipeline_norm_affine(data, False, [_maps_of(r) for r in recs], sigma, noise, stride)
    alive, cur = None, P
    if 'ball_u' in recs[0]:
        u = np.zeros((B, n_add, 3), np.float32)
        count = np.zeros(B, np.int32)
        for b, r in enumerate(recs):
            count[b] = int(P * (int(r['add_level']) + 1) * 0.1)
            u[b, :len(r['ball_u'])] = r['ball_u']
        D.pipeline_add_global(y, P, torch.from_numpy(count).cuda(), torch.from_numpy(u).cuda())
        cur = stride
    if 'local_order' in recs[0]:
        seed, sig = np.zeros((B, n_add), np.int32), np.zeros((B, n_add), np.float32)
        nz, count = np.zeros((B, n_add, 3), np.float32), np.zeros(B, np.int32)
        for b, r in enumerate(recs):
            sizes = r['local_sizes']
            count[b] = sizes.sum()
            seed[b, :count[b]] = np.repeat(r['local_order'][:len(sizes)], sizes)      # cluster i sits on shuffled row i
            sig[b, :count[b]] = np.repeat(r['local_sigmas'], sizes)
            nz[b, :count[b]] = r['local_noise']
        D.pipeline_add_local(y, P, torch.from_numpy(count).cuda(), torch.from_numpy(seed).cuda(),
                             torch.from_numpy(sig).cuda(), torch.from_numpy(nz).cuda())
        cur = stride
    if 'density_v' in recs[0]:
        v = np.stack([r['density_v'] / np.linalg.norm(r['density_v']) for r in recs]).astype(np.float32)
        gate = np.array([int(r['density_level']) / 4.0 + 0.1 for r in recs], np.float32)
        alive = torch.ones((B, stride), dtype=torch.uint8, device=dev)
        D.pipeline_density(y, P, torch.from_numpy(v).cuda(), torch.from_numpy(gate).cuda(),
                           torch.from_numpy(np.stack([r['density_r'] for r in recs])).cuda(), alive)
        got = alive.cpu().numpy().astype(bool)
        assert np.array_equal(got, np.stack([r['density_keep'] for r in recs]))
    if 'dl_nclusters' in recs[0]:
        # the affine stage within tolerance of the reference's fp64 chain ...
        assert np.abs(y.cpu().numpy() - np.stack([r['dl_input'] for r in recs])).max() <= 5e-6
        # ... and the drop bit-exact on the reference's own fp32 cloud
        x = torch.from_numpy(np.stack([r['dl_input'] for r in recs])).cuda()
        keep = D.dropout_local(x, np.concatenate([r['dl_nclusters'] for r in recs]),
                               np.concatenate([r['dl_rank'] for r in recs]), np.concatenate([r['dl_sizes'] for r in recs]))
        assert np.array_equal(keep.cpu().numpy(), np.stack([r['dl_alive'] for r in recs]))
        alive = keep.to(torch.uint8)
    keys = torch.stack([_keys_for(r['select'], stride, dev) for r in recs])
    out = D.pipeline_subset(y, cur, NPTS, keys, alive).cpu().numpy()
    # a cloud that kept fewer than n points went through the reference's refill branch (np.random.choice with
    # replacement, then a shuffle): its item repeats points, which no key order expresses -- for it the stages above
    # are what is pinned (mask / added points), the sub-sampled item is compared as a SET of distinct points
    for b, r in enumerate(recs):
        if len(np.unique(r['select'])) == len(r['select']):
            assert np.abs(out[b] - r['corrupted']).max() <= 1e-5, (b, np.abs(out[b] - r['corrupted']).max())
        else:
            assert int(r['n_before_sample']) < NPTS
            d = np.abs(out[b][:, None, :] - r['corrupted'][None, :, :]).max(-1)       # (n, n) max-norm distances
            assert d.min(1).max() <= 1e-5 and d.min(0).max() <= 1e-5, b          # the same set of distinct points


@pytest.mark.gpu
def test_hip_subset_edge_cases():
    """random keys: a uniformly random subset in random order; ties by index; fewer survivors than n: cyclic refill."""
    import torch
    from point_dae_amd import datasets as D
    g = torch.Generator(device='cuda').manual_seed(0)
    B, P, n = 3, 1500, 256
    y = torch.randn((B, P, 3), device='cuda', generator=g)
    keys = torch.rand((B, P), device='cuda', generator=g)
    alive = (torch.rand((B, P), device='cuda', generator=g) < 0.7).to(torch.uint8)
    out = D.pipeline_subset(y, P, n, keys, alive)
    for b in range(B):
        k = keys[b].clone()
        k[alive[b] == 0] = 9.0
        order = torch.argsort(k, stable=True)[:n]
        assert torch.equal(out[b], y[b, order])
    keys0 = torch.zeros((B, P), device='cuda')                       # all ties: index order
    out = D.pipeline_subset(y, P, n, keys0)
    assert torch.equal(out, y[:, :n])
    few = torch.zeros((B, P), dtype=torch.uint8, device='cuda')
    few[:, 10:15] = 1                                                # 5 survivors, n = 12
    out = D.pipeline_subset(y, P, 12, keys, few)
    for b in range(B):
        order = 10 + torch.argsort(keys[b, 10:15])
        assert torch.equal(out[b], y[b, order[torch.arange(12, device='cuda') % 5]])
    none = torch.zeros((B, P), dtype=torch.uint8, device='cuda')
    assert D.pipeline_subset(y, P, 8, keys, none).abs().max().item() == 0.0


# ---- every entry of the reference's corruption table (tests/golden/make_loader_variant_fixtures.py) -------------------
def _variants():
    import json
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'loader_variants_ref.npz'))
    return d, json.loads(str(d['meta']))


def _uniform_call(call):
    """(low, high, size) of a recorded np.random.uniform call."""
    tag, args, kw = call
    assert tag == 'np.uniform', call
    low = kw['low'] if 'low' in kw else args[0]
    high = kw['high'] if 'high' in kw else args[1]
    size = kw['size'] if 'size' in kw else args[2]
    return float(low), float(high), int(np.prod(size))


def test_loader_tables_carry_the_reference_parameters():
    """point_dae_amd/datasets.py's tables against the draws the LIVE reference functions made (arguments recorded):
    every name of corrupt_util.corruptions (:984-1038) is implemented with the reference's ranges, the affine sets use
    the reference's pools and counts, and the YAML names the reference itself cannot dispatch are refused as such."""
    from point_dae_amd import datasets as D
    _, meta = _variants()
    table, P = meta['table'], 128
    assert set(table) <= set(D._CORRUPTIONS), sorted(set(table) - set(D._CORRUPTIONS))
    assert sorted(meta['unknown_upstream']) == sorted(D._UNKNOWN_UPSTREAM)
    for item, rec in meta['sets'].items():
        pool, most = D._AFFINE_SETS[item]
        assert list(pool) == rec['pool'] and list(range(1, most + 1)) == rec['numbers'], item
    assert set(meta['sets']) == set(D._AFFINE_SETS)
    want_size = {'translate': 3, 'scale': 3, 'rotate': 3, 'rotate_z': 1, 'shear': 6}
    for name, (kind, par) in D._MAP_PARAMS.items():
        if name.startswith('aug_'):
            continue
        for level in range(5):
            call = table[name][str(level)]['calls'][0]
            k, c = kind, par
            if kind.endswith('_level'):
                k, c = kind[:-6], par * (level + 1)
            if k == 'reflection':
                assert call[0] == 'np.choice' and sorted(call[1][0]) == [-1, 1] and call[2]['size'] == 3, (name, call)
                continue
            low, high, size = _uniform_call(call)
            lo_w, hi_w = (1.0 / c, c) if k == 'scale' else (-c, c)
            assert size == want_size[k] and abs(low - lo_w) < 1e-12 and abs(high - hi_w) < 1e-12, (name, level, call)
    for level in range(5):
        low, high, size = _uniform_call(table['scale_single'][str(level)]['calls'][0])
        s = D._SCALE_SINGLE[level]
        assert size == 1 and abs(low - 1 / s) < 1e-12 and abs(high - s) < 1e-12
        assert _uniform_call(table['scale'][str(level)]['calls'][0]) == (0.5, 2.0, 3)
    for name, fixed in D._JITTER.items():
        for level in range(5):
            want = 0.01 * (level + 1) if fixed is None else fixed
            assert abs(table[name][str(level)]['sigma'] - want) < 1e-9, (name, level)
    for name, (ratio, hi) in D._DROPOUT_LOCAL.items():
        for level in range(5):
            calls = table[name][str(level)]['calls']
            i = 0
            if ratio is None:
                assert _uniform_call(calls[0]) == (0.1, 0.5, 1), (name, calls[0])
                i = 1
            if hi:
                assert calls[i][0] == 'np.randint' and calls[i][1] == [1, hi], (name, calls[i])
                i += 1
            labels = calls[i]                               # _gen_random_cluster_sizes: randint(num_clusters, size=total)
            assert labels[0] == 'np.randint' and (hi or labels[1] == [1]), (name, labels)
            if ratio is not None:
                assert labels[2]['size'] == int(P * ratio), (name, labels)


def test_affine_sets_draw_from_their_pools():
    from point_dae_amd import datasets as D
    rng = np.random.default_rng(5)
    for item, (pool, most) in D._AFFINE_SETS.items():
        counts = set()
        for _ in range(200):
            number = int(rng.integers(1, most + 1))
            counts.add(number)
        assert counts == set(range(1, most + 1))
        A, t = D.draw_affine_set(rng, 64, item)
        assert np.isfinite(A).all() and np.abs(np.linalg.det(A)).min() > 1e-4


@pytest.mark.gpu
def test_hip_single_maps_reproduce_the_live_reference_functions():
    """every single-map entry of the table, every level: the device kernel applied to the map built from the draw the
    LIVE function made == that function's output (fp64 there, fp32 here: 2e-6 of the cloud's extent)."""
    import torch
    from point_dae_amd import datasets as D
    d, meta = _variants()
    cloud = torch.from_numpy(d['cloud']).cuda()[None]
    checked = 0
    for name in D._MAP_PARAMS:
        if name.startswith('aug_'):
            continue
        for level in range(5):
            draw, want = d['%s/%d/draw' % (name, level)], d['%s/%d/out' % (name, level)]
            got = D.pipeline_norm_affine(cloud, False, [[D.affine_map_from_draw(name, draw)]])[0].cpu().numpy()
            assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max()), (name, level)
            checked += 1
    for name in ('scale', 'scale_single'):                  # a scale, then re-normalised (corrupt_scale / _single)
        for level in range(5):
            draw, want = d['%s/%d/draw' % (name, level)], d['%s/%d/out' % (name, level)]
            M = np.diag(draw) if name == 'scale' else np.eye(3) * draw[0]
            got = D.pipeline_norm_affine(D.pipeline_norm_affine(cloud, False, [[(M, np.zeros(3))]]), True)[0].cpu().numpy()
            assert np.abs(got - want).max() <= 2e-6, (name, level)
            checked += 1
    assert checked == 5 * (len([n for n in D._MAP_PARAMS if not n.startswith('aug_')]) + 2)


@pytest.mark.gpu
@pytest.mark.parametrize('cor', [['affine_r5'], ['affine_r3_v2', 'dropout_local_c5d3'], ['affine_r5_v2', 'dropout_local_c5d3'],
                                 ['jitter_p03'], ['scale_single'], ['shear_small'], ['rotate_z'], ['dropout_local_c1d3'],
                                 ['dropout_local_c8d3'], ['translate_too_large'], ['scale_nonorm_10'], ['rotate_level2']])
def test_device_dataset_runs_every_family_of_the_table(cor):
    """the remaining names of the reference's table through ShapeNet.batch end to end (shapes, finiteness, survivor
    counts of the fixed-ratio drops)"""
    import torch
    from point_dae_amd.datasets import ShapeNet
    ds = ShapeNet({'npoints': 512, 'N_POINTS': 2048, 'bs': 6, 'steps_per_epoch': 2, 'device': 'cuda', 'seed': 4,
                   'aug_type': ['norm'], 'corrupt_type': cor})
    for _, _, corrupted, clean in ds:
        assert corrupted.shape == (6, 512, 3) and clean.shape == (6, 512, 3)
        assert torch.isfinite(corrupted).all() and not torch.equal(corrupted, clean)
        if cor == ['scale_single']:
            assert (corrupted.norm(dim=-1).amax(dim=1) <= 1 + 1e-5).all()
        if cor in (['rotate_z'], ['rotate_level2'], ['dropout_local_c1d3'], ['dropout_local_c8d3']):
            assert (corrupted.norm(dim=-1).amax(dim=1) <= 1 + 1e-4).all()      # rigid maps / drops stay in the unit sphere


@pytest.mark.gpu
def test_loader_does_not_wait_for_the_gpu():
    """Every host -> device transfer of a batch goes through pinned staging: with ~0.3 s of GPU work queued on the stream,
    producing the next batch must return to the host long before that work is done (a pageable .to(device) would block
    until the stream reached it -- the training loop's host would then never run ahead of the GPU)."""
    import time
    import torch
    from point_dae_amd.datasets import ShapeNet
    ds = ShapeNet({'npoints': 512, 'N_POINTS': 2048, 'bs': 8, 'steps_per_epoch': 3, 'device': 'cuda', 'seed': 2,
                   'aug_type': ['norm', 'scale', 'translate'], 'corrupt_type': ['affine_r3', 'dropout_local']})
    it = iter(ds)
    next(it)                                                  # materialise, warm the pinned-memory cache
    torch.cuda.synchronize()
    a = torch.randn(8192, 8192, device='cuda')
    (a @ a).sum().item()                                       # (the first product loads the BLAS library: not timed)
    for _ in range(40):
        a = (a @ a).clamp_(-1, 1)
    t0 = time.perf_counter()
    batch = next(it)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    assert total > 0.1, 'the GPU was not busy: the test proves nothing (%.3f s)' % total
    assert host < 0.3 * total, (host, total)                   # (measured: 1.4 ms of 280)
    assert torch.isfinite(batch[2]).all()
