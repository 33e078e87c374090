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
        ShapeNet({'corrupt_type': ['jitter'], 'device': 'cpu'})
    with pytest.raises(NotImplementedError):
        ShapeNet({'aug_type': ['rotate'], 'device': 'cpu'})


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
