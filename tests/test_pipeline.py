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
    with pytest.raises(NotImplementedError):
        ShapeNet({'corrupt_type': ['scan'], 'device': 'cpu'})
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
                                     (['norm'], ['scale']), (['norm'], ['add_global', 'dropout_local']), (['norm'], ['add_local'])])
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
        if 'scale' in aug:
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
