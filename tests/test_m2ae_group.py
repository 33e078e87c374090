"""Point-M2AE hierarchical grouping: models/Point_M2AE_modules.py:219-248 (`Group` returning idx),
models/Point_M2AE.py:245-263 (the three-level pyramid), :107-121 (multi-scale masking), :132 (token merging).

tests/golden/m2ae_grouping_b2.npz holds the LIVE reference's outputs on two seeded 2048-point clouds
(make_m2ae_fixtures.py).  CPU: the oracle restatement reproduces them.  GPU: the HIP path reproduces the fixture
and -- on other seeds and shapes -- the oracle, bit for bit (indices, centres, neighbourhoods, masks)."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, 'tests', 'golden', 'm2ae_grouping_b2.npz')
NUM_GROUPS, GROUP_SIZES = [512, 256, 64], [16, 8, 8]


def test_oracle_reproduces_live_reference_fixture(oracle_ops):
    f = np.load(FIX)
    nb, c, idx = oracle_ops.m2ae_hierarchy(f['pts'], NUM_GROUPS, GROUP_SIZES)
    masks = oracle_ops.m2ae_multi_scale_mask(f['top_mask'], idx, c)
    for i in range(3):
        assert np.array_equal(idx[i], f['idx%d' % i].astype(np.int64))
        assert np.array_equal(c[i], f['center%d' % i])
        assert np.array_equal(nb[i][:, ::37], f['nb%d_sample' % i])
        assert np.array_equal(masks[i], f['mask%d' % i])
    # the quirk the reference's index arithmetic carries (Point_M2AE.py:114): flat token 0 of every finer level is
    # visible whenever a coarser token is masked
    assert not masks[0].reshape(-1)[0] and not masks[1].reshape(-1)[0]


def test_multi_scale_mask_quirk_and_no_mask_case(oracle_ops):
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1, 1, (2, 256, 3)).astype(np.float32)
    nb, c, idx = oracle_ops.m2ae_hierarchy(pts, [64, 16], [8, 4])
    none = oracle_ops.m2ae_multi_scale_mask(np.zeros((2, 16), bool), idx, c)
    # nothing masked at the top: exactly the children of some coarse token are visible
    want = np.ones(2 * 64, bool)
    want[idx[1]] = False
    assert np.array_equal(none[0].reshape(-1), want)
    allm = oracle_ops.m2ae_multi_scale_mask(np.ones((2, 16), bool), idx, c)
    assert allm[0].reshape(-1)[1:].all() and not allm[0].reshape(-1)[0]      # only the quirk's element 0 is visible


@pytest.mark.gpu
def test_hip_pyramid_reproduces_live_reference_fixture():
    from point_dae_amd.point_m2ae_group import HierarchicalGroup, merge_tokens, multi_scale_mask
    f = np.load(FIX)
    pts = torch.from_numpy(f['pts']).cuda()
    nbs, cs, idxs = HierarchicalGroup(NUM_GROUPS, GROUP_SIZES)(pts)
    masks = multi_scale_mask(torch.from_numpy(f['top_mask']).cuda(), idxs, cs)
    for i in range(3):
        assert idxs[i].dtype == torch.int64 and idxs[i].dim() == 1
        assert np.array_equal(idxs[i].cpu().numpy(), f['idx%d' % i].astype(np.int64)), i
        assert np.array_equal(cs[i].cpu().numpy(), f['center%d' % i]), i
        assert np.array_equal(nbs[i].cpu().numpy()[:, ::37], f['nb%d_sample' % i]), i
        assert masks[i].dtype == torch.bool and np.array_equal(masks[i].cpu().numpy(), f['mask%d' % i]), i
    merged = merge_tokens(torch.from_numpy(f['feat']).cuda(), idxs[1], 2, NUM_GROUPS[1], GROUP_SIZES[1])
    assert np.array_equal(merged.cpu().numpy(), f['merged'])


@pytest.mark.gpu
@pytest.mark.parametrize('B,N,groups,sizes,seed', [
    (3, 1024, [256, 64], [16, 8], 1),
    (5, 2048, [512, 256, 64], [16, 8, 8], 2),
    (2, 333, [100, 37, 9], [7, 5, 3], 3),          # ragged sizes: odd g*k (the un-vectorised index pass), k not 2^n
    (1, 64, [64, 64], [1, 64], 4),                 # every point a centre; k = 1 and k = N
])
def test_hip_pyramid_equals_oracle(oracle_ops, B, N, groups, sizes, seed):
    from point_dae_amd.point_m2ae_group import HierarchicalGroup, multi_scale_mask, rand_mask
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1, 1, (B, N, 3)).astype(np.float32)
    pts[0, 5] = pts[0, 2]                                               # duplicated point: ties
    nb, c, idx = oracle_ops.m2ae_hierarchy(pts, groups, sizes)
    nbs, cs, idxs = HierarchicalGroup(groups, sizes)(torch.from_numpy(pts).cuda())
    for i in range(len(groups)):
        assert np.array_equal(idxs[i].cpu().numpy(), idx[i]), i
        assert np.array_equal(cs[i].cpu().numpy(), c[i]), i
        assert np.array_equal(nbs[i].cpu().numpy(), nb[i]), i
    np.random.seed(seed)
    top = rand_mask(B, groups[-1], 0.8)                                 # the reference's host draw (:85-97)
    assert top.shape == (B, groups[-1]) and int(top.sum()) == B * int(0.8 * groups[-1])
    want = oracle_ops.m2ae_multi_scale_mask(top.numpy(), idx, c)
    got = multi_scale_mask(top.cuda(), idxs, cs)
    for i in range(len(groups)):
        assert np.array_equal(got[i].cpu().numpy(), want[i]), i
    for t in (np.zeros((B, groups[-1]), bool), np.ones((B, groups[-1]), bool)):     # nothing / everything masked
        want = oracle_ops.m2ae_multi_scale_mask(t, idx, c)
        got = multi_scale_mask(torch.from_numpy(t).cuda(), idxs, cs)
        for i in range(len(groups)):
            assert np.array_equal(got[i].cpu().numpy(), want[i]), i


@pytest.mark.gpu
def test_merge_tokens_gradient_is_a_scatter_add():
    from point_dae_amd.point_m2ae_group import Group, merge_tokens
    g = torch.Generator().manual_seed(0)
    pts = torch.rand(2, 128, 3, generator=g).cuda()
    _, center, _ = Group(32, 8)(pts)
    _, _, idx = Group(8, 4)(center)
    x = torch.randn(2, 32, 6, generator=g).cuda().requires_grad_()
    w = torch.randn(2, 8, 4, 6, generator=g).cuda()
    (merge_tokens(x, idx, 2, 8, 4) * w).sum().backward()
    want = torch.zeros(64, 6, device='cuda').index_add_(0, idx, w.reshape(-1, 6))
    assert torch.allclose(x.grad.reshape(64, 6), want, atol=1e-6)
