"""GPU parity tests: the gfx950 kernels, called through the C ABI
(point_dae_amd/_lib.py -> libpdae_hip.so), against the CPU oracle on the same
seeded inputs.  Integer / index outputs and every fp32 value that has a fixed
evaluation order must be BIT-EXACT; only scatter-add gradients (atomic order is
unspecified in the reference as well) and EMD's hardware exp use a tolerance,
written at the assert.
"""
import numpy as np
import pytest
import torch

from conftest import make_clouds

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------ FPS ----
@pytest.mark.parametrize("B,N,m,kind", [
    (8, 1024, 64, "shapes"), (4, 1024, 512, "shapes"), (4, 512, 128, "shapes"),
    (4, 2048, 128, "shapes"), (3, 100, 17, "uniform"), (2, 64, 64, "uniform"),
    (2, 8192, 1024, "shapes"), (2, 1500, 33, "uniform"), (1, 5000, 40, "uniform"),
    (1, 20000, 16, "uniform"), (5, 1, 1, "uniform"),
])
def test_fps_bit_exact(oracle_ops, B, N, m, kind):
    from point_dae_amd import pointnet2_utils as pu
    x = make_clouds(21, B, N, kind)
    want_idx, want_ctr = oracle_ops.furthest_point_sample(x, m, return_centres=True)
    xd = dev(x)
    got = pu.furthest_point_sample(xd, m)
    assert got.dtype == torch.int32
    np.testing.assert_array_equal(host(got), want_idx)
    idx2, ctr = pu.furthest_point_sample_with_centres(xd, m)
    np.testing.assert_array_equal(host(idx2), want_idx)
    np.testing.assert_array_equal(host(ctr), want_ctr)


def test_fps_edge_cases(oracle_ops):
    from point_dae_amd import pointnet2_utils as pu
    x = make_clouds(22, 4, 1024)
    x[0, 5:400] *= 0.01                      # origin-ball points are skipped
    x[1, :] = x[1, :1]                       # all points identical -> tie order decides
    x[2, 100:900] = x[2, 100]                # many duplicates
    x[3, :] = 0.0                            # everything skipped -> zeros
    for m in (64, 512, 1024):
        want = oracle_ops.furthest_point_sample(x, m)
        np.testing.assert_array_equal(host(pu.furthest_point_sample(dev(x), m)), want)
    # tie rule with a non power-of-two N (block size 512, 3 columns)
    y = np.tile(np.float32([[0.3, 0.2, 0.9]]), (1, 1500, 1))
    y[0, [7, 519, 1031, 300, 1400, 812]] = [-0.3, 0.1, -0.9]
    np.testing.assert_array_equal(host(pu.furthest_point_sample(dev(y), 5)),
                                  oracle_ops.furthest_point_sample(y, 5))


def test_fps_full_size_properties():
    """BASELINE shape (B=128, N=1024, m=64): size-independent checks."""
    from point_dae_amd import pointnet2_utils as pu
    x = make_clouds(23, 128, 1024, "shapes")
    xd = dev(x)
    idx, ctr = pu.furthest_point_sample_with_centres(xd, 64)
    idx_h = host(idx)
    assert (idx_h[:, 0] == 0).all()
    assert all(len(set(r)) == 64 for r in idx_h)           # no repeats on tie-free data
    np.testing.assert_array_equal(host(ctr), np.take_along_axis(x, idx_h[..., None].astype(np.int64), 1))
    # greedy property: each new sample is at least as far from the chosen set as any later one
    c = host(ctr).astype(np.float64)
    d = ((c[:, :, None, :] - c[:, None, :, :]) ** 2).sum(-1)
    for j in range(2, 64):
        dj = d[:, j, :j].min(1)
        djm1 = d[:, j - 1, :j - 1].min(1)
        assert (dj <= djm1 + 1e-6).all()
    # idempotence / determinism
    np.testing.assert_array_equal(host(pu.furthest_point_sample(xd, 64)), idx_h)


# ------------------------------------------------------------------ kNN ----
@pytest.mark.parametrize("B,N,G,k,kind", [
    (8, 1024, 64, 32, "shapes"), (3, 2048, 128, 32, "shapes"), (2, 77, 9, 5, "uniform"),
    (2, 40, 3, 40, "uniform"), (2, 512, 128, 64, "uniform"), (1, 3000, 10, 20, "uniform"),
    (2, 64, 64, 1, "uniform"), (2, 200, 1, 16, "uniform"),
])
def test_knn_bit_exact(oracle_ops, B, N, G, k, kind):
    from point_dae_amd.knn_cuda import KNN, knn
    x = make_clouds(31, B, N, kind)
    q = x[:, :G].copy()
    if G > 2:
        q[:, 1] += 0.013            # queries that are not cloud points
    wd, wi, wn = oracle_ops.knn(x, q, k, return_nbr=True)
    d, i, n = knn(dev(x), dev(q), k, with_neighbourhood=True)
    assert i.dtype == torch.int64
    np.testing.assert_array_equal(host(i), wi)
    np.testing.assert_array_equal(host(d), wd)
    np.testing.assert_array_equal(host(n), wn)
    d2, i2 = KNN(k=k, transpose_mode=True)(dev(x), dev(q))
    np.testing.assert_array_equal(host(i2), wi)
    d3, i3 = KNN(k=k, transpose_mode=False)(dev(x).transpose(1, 2), dev(q).transpose(1, 2))
    np.testing.assert_array_equal(host(i3), wi.transpose(0, 2, 1))


def test_knn_ties_and_duplicates(oracle_ops):
    from point_dae_amd.knn_cuda import knn
    x = make_clouds(32, 3, 1024)
    x[0, 100:400] = x[0, 100]                 # 300 equal distances: earlier index first
    x[1, :] = x[1, :1]                        # every distance equal
    x[2] = np.round(x[2] * 4) / 4             # lattice: massive ties
    q = x[:, ::16].copy()
    wd, wi = oracle_ops.knn(x, q, 32)
    d, i = knn(dev(x), dev(q), 32)
    np.testing.assert_array_equal(host(i), wi)
    np.testing.assert_array_equal(host(d), wd)


def test_knn_full_size_properties():
    from point_dae_amd.knn_cuda import knn
    from point_dae_amd import pointnet2_utils as pu
    x = make_clouds(33, 128, 1024, "shapes")
    xd = dev(x)
    _, ctr = pu.furthest_point_sample_with_centres(xd, 64)
    d, i, nbr = knn(xd, ctr, 32, with_neighbourhood=True)
    d, i, nbr = host(d), host(i), host(nbr)
    assert (np.diff(d, axis=2) >= 0).all()                          # sorted
    assert (d[:, :, 0] == 0).all()                                  # centre is its own NN
    assert all(len(set(r)) == 32 for r in i.reshape(-1, 32))        # distinct
    c = host(ctr)
    gathered = np.take_along_axis(x[:, None].repeat(64, 1), i[..., None], 2)
    np.testing.assert_array_equal(nbr, gathered - c[:, :, None, :])
    # the 32nd distance bounds every point that was left out (checked on a slice)
    full = np.sqrt((((x[:4, None] - c[:4, :, None]) ** 2).astype(np.float32)).sum(-1))
    kth = d[:4, :, -1][..., None]
    assert ((full < kth - 1e-6).sum(-1) <= 32).all()


# ----------------------------------------------------------- ball query ----
@pytest.mark.parametrize("B,N,m,r,ns,kind", [
    (4, 1024, 512, 0.2, 32, "shapes"), (4, 512, 128, 0.4, 64, "shapes"),
    (2, 100, 7, 0.05, 8, "uniform"), (2, 1000, 33, 0.3, 100, "uniform"), (1, 70, 70, 10.0, 16, "uniform"),
])
def test_ball_query_bit_exact(oracle_ops, B, N, m, r, ns, kind):
    from point_dae_amd import pointnet2_utils as pu
    x = make_clouds(41, B, N, kind)
    c = x[:, :m].copy()
    c[0, 0] = [5, 5, 5]                        # empty ball -> zeros
    want = oracle_ops.ball_query(r, ns, x, c)
    got = pu.ball_query(r, ns, dev(x), dev(c))
    assert got.dtype == torch.int32
    np.testing.assert_array_equal(host(got), want)


# -------------------------------------------------------- group / gather ----
@pytest.mark.parametrize("B,C,N,npnt,ns", [(4, 3, 1024, 512, 32), (2, 131, 512, 128, 64), (2, 5, 50, 7, 3)])
def test_group_points(oracle_ops, B, C, N, npnt, ns):
    from point_dae_amd import pointnet2_utils as pu
    rng = np.random.default_rng(51)
    f = rng.normal(size=(B, C, N)).astype(np.float32)
    idx = rng.integers(0, N, (B, npnt, ns)).astype(np.int32)
    fd = dev(f).requires_grad_(True)
    out = pu.grouping_operation(fd, dev(idx))
    np.testing.assert_array_equal(host(out), oracle_ops.grouping_operation(f, idx))
    go = rng.normal(size=out.shape).astype(np.float32)
    out.backward(dev(go))
    want = oracle_ops.grouping_operation_grad(go, idx, N)
    # scatter-add: summation order differs (LDS atomics) -> fp32 reassociation only
    np.testing.assert_allclose(host(fd.grad), want, rtol=1e-4, atol=1e-5)


def test_gather_operation(oracle_ops):
    from point_dae_amd import pointnet2_utils as pu
    rng = np.random.default_rng(52)
    f = rng.normal(size=(4, 6, 1024)).astype(np.float32)
    idx = rng.integers(0, 1024, (4, 64)).astype(np.int32)
    fd = dev(f).requires_grad_(True)
    out = pu.gather_operation(fd, dev(idx))
    np.testing.assert_array_equal(host(out), oracle_ops.gather_operation(f, idx))
    go = rng.normal(size=out.shape).astype(np.float32)
    out.backward(dev(go))
    np.testing.assert_allclose(host(fd.grad), oracle_ops.gather_operation_grad(go, idx, 1024),
                               rtol=1e-4, atol=1e-5)


def test_query_and_group_module(oracle_ops):
    from point_dae_amd import pointnet2_utils as pu
    x = make_clouds(53, 2, 512, "shapes")
    new = x[:, :64].copy()
    feats = np.random.default_rng(54).normal(size=(2, 8, 512)).astype(np.float32)
    out = pu.QueryAndGroup(0.3, 16)(dev(x), dev(new), dev(feats))
    idx = oracle_ops.ball_query(0.3, 16, x, new)
    gx = oracle_ops.grouping_operation(x.transpose(0, 2, 1).copy(), idx) - new.transpose(0, 2, 1)[..., None]
    gf = oracle_ops.grouping_operation(feats, idx)
    np.testing.assert_array_equal(host(out), np.concatenate([gx, gf], 1))


# -------------------------------------------------------------- Chamfer ----
@pytest.mark.parametrize("B,n,m", [
    (64, 32, 32), (37, 32, 32), (5, 36, 32), (16, 64, 64), (3, 64, 128), (2, 1024, 1024),
    (2, 600, 1100), (1, 16384, 1024), (4, 3, 1), (2, 300, 7), (300, 8, 8),
])
def test_chamfer_forward_bit_exact(oracle_ops, B, n, m):
    from point_dae_amd import chamfer_dist
    a = make_clouds(61, B, n)
    b = make_clouds(62, B, m)
    wd1, wd2, wi1, wi2 = oracle_ops.chamfer_forward(a, b)
    d1, d2, i1, i2 = chamfer_dist.forward(dev(a), dev(b))
    assert i1.dtype == torch.int32
    np.testing.assert_array_equal(host(i1), wi1)
    np.testing.assert_array_equal(host(i2), wi2)
    np.testing.assert_array_equal(host(d1), wd1)
    np.testing.assert_array_equal(host(d2), wd2)


def test_chamfer_forward_ties(oracle_ops):
    from point_dae_amd import chamfer_dist
    a = np.round(make_clouds(63, 4, 700) * 3) / 3          # lattice: many exact ties
    b = np.round(make_clouds(64, 4, 1300) * 3) / 3
    w = oracle_ops.chamfer_forward(a, b)
    g = chamfer_dist.forward(dev(a), dev(b))
    for x, y in zip(g, w):
        np.testing.assert_array_equal(host(x), y)
    a = np.round(make_clouds(65, 40, 32) * 2) / 2
    b = np.round(make_clouds(66, 40, 32) * 2) / 2
    w = oracle_ops.chamfer_forward(a, b)
    g = chamfer_dist.forward(dev(a), dev(b))
    for x, y in zip(g, w):
        np.testing.assert_array_equal(host(x), y)


@pytest.mark.parametrize("B,n,m,exact", [(64, 32, 32, True), (37, 36, 32, True), (3, 64, 128, True),
                                         (2, 1024, 1024, False), (1, 16384, 1024, False),
                                         # many onto few: the counting-sort scatter (n >= 4 m), both argument orders,
                                         # every prediction on ONE target, more blocks than one per cloud
                                         (3, 4096, 300, False), (2, 257, 2050, False), (5, 1024, 1, False)])
def test_chamfer_backward(oracle_ops, B, n, m, exact):
    from point_dae_amd import chamfer_dist
    rng = np.random.default_rng(67)
    a = make_clouds(68, B, n)
    b = make_clouds(69, B, m)
    g1 = rng.normal(size=(B, n)).astype(np.float32)
    g2 = rng.normal(size=(B, m)).astype(np.float32)
    _, _, i1, i2 = oracle_ops.chamfer_forward(a, b)
    w1, w2 = oracle_ops.chamfer_backward(a, b, i1, i2, g1, g2)
    o1, o2 = chamfer_dist.backward(dev(a), dev(b), dev(i1), dev(i2), dev(g1), dev(g2))
    if exact:      # gather form: same summation order as the oracle
        np.testing.assert_array_equal(host(o1), w1)
        np.testing.assert_array_equal(host(o2), w2)
    else:          # atomic scatter: order unspecified (as in the reference) -> reassociation only
        np.testing.assert_allclose(host(o1), w1, rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(host(o2), w2, rtol=1e-4, atol=1e-5)


def test_chamfer_loss_modules_and_autograd(oracle_ops):
    """North-star bar: Chamfer loss within 1e-5 relative of the reference CPU path."""
    from point_dae_amd.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2, ChamferDistanceL2_split
    a = make_clouds(70, 128 * 41 // 8, 32)     # B' = B*M/8 patches of 32 points
    b = make_clouds(71, a.shape[0], 32)
    ad = dev(a).requires_grad_(True)
    bd = dev(b).requires_grad_(True)
    l2 = ChamferDistanceL2()(ad, bd)
    want = oracle_ops.chamfer_distance_l2(a, b)
    assert abs(l2.item() - want) <= 1e-5 * abs(want)
    l2.backward()
    d1, d2, i1, i2 = oracle_ops.chamfer_forward(a, b)
    w1, w2 = oracle_ops.chamfer_backward(a, b, i1, i2, np.full_like(d1, 1.0 / d1.size),
                                         np.full_like(d2, 1.0 / d2.size))
    np.testing.assert_allclose(host(ad.grad), w1, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(host(bd.grad), w2, rtol=1e-5, atol=1e-9)
    l1 = ChamferDistanceL1()(dev(a), dev(b)).item()
    want1 = oracle_ops.chamfer_distance_l1(a, b)
    assert abs(l1 - want1) <= 1e-5 * abs(want1)
    s1, s2 = ChamferDistanceL2_split()(dev(a), dev(b))
    assert abs(s1.item() + s2.item() - want) <= 1e-5 * abs(want)


def test_chamfer_full_size_properties():
    """cfg2 fine shape (B=128, 16384 x 1024) and cfg3 patch shape (B'=5248, 32 x 32)."""
    from point_dae_amd import chamfer_dist
    for B, n, m in ((128, 16384, 1024), (5248, 32, 32)):
        g = torch.Generator(device="cuda").manual_seed(5)
        a = torch.rand((B, n, 3), device="cuda", generator=g) * 2 - 1
        b = torch.rand((B, m, 3), device="cuda", generator=g) * 2 - 1
        d1, d2, i1, i2 = chamfer_dist.forward(a, b)
        # the reported distance is the distance to the reported index
        nb = torch.gather(b, 1, i1.long()[..., None].expand(-1, -1, 3))
        dx = a - nb
        np.testing.assert_array_equal(host((dx[..., 0] * dx[..., 0] + dx[..., 1] * dx[..., 1]) + dx[..., 2] * dx[..., 2]), host(d1))
        # symmetry: swapping the clouds swaps the outputs
        e1, e2, j1, j2 = chamfer_dist.forward(b, a)
        assert torch.equal(e1, d2) and torch.equal(e2, d1) and torch.equal(j1, i2) and torch.equal(j2, i1)
        # a cloud against itself: zero distance, identity index
        z1, z2, k1, k2 = chamfer_dist.forward(a[:8], a[:8])
        assert (z1 == 0).all() and (z2 == 0).all()
        assert torch.equal(k1.long(), torch.arange(n, device="cuda").expand(8, n))
        # the minimum really is a minimum (spot check against a dense torch computation)
        dense = torch.cdist(a[:2].double(), b[:2].double()) ** 2
        assert torch.allclose(dense.min(2).values.float(), d1[:2], rtol=1e-4, atol=1e-6)


def test_chamfer_gradcheck_fp32_directional():
    """extensions/chamfer_dist/test.py:23-29 (gradcheck) on the GPU path: the
    analytic backward matches central differences of the forward."""
    from point_dae_amd.chamfer_dist import ChamferFunction
    torch.manual_seed(0)
    x = torch.rand(4, 64, 3, device="cuda").double()
    y = torch.rand(4, 128, 3, device="cuda").double()
    w1 = torch.rand(4, 64, device="cuda").double()
    w2 = torch.rand(4, 128, device="cuda").double()

    def f(x, y):
        d1, d2, _, _ = ChamferFunction.apply(x.float(), y.float())
        return (w1 * d1.double()).sum() + (w2 * d2.double()).sum()

    xf = x.float().requires_grad_(True)
    yf = y.float().requires_grad_(True)
    d1, d2, _, _ = ChamferFunction.apply(xf, yf)
    ((w1.float() * d1).sum() + (w2.float() * d2).sum()).backward()
    for arr, g in ((x, xf.grad), (y, yf.grad)):
        v = torch.randn_like(arr)
        v /= v.norm()
        eps = 1e-3
        if arr is x:
            num = (f(x + eps * v, y) - f(x - eps * v, y)) / (2 * eps)
        else:
            num = (f(x, y + eps * v) - f(x, y - eps * v)) / (2 * eps)
        ana = (g.double() * v).sum()
        assert abs(num - ana) <= 2e-2 * max(1.0, abs(num)), (num.item(), ana.item())


# ------------------------------------------------------------------ EMD ----
def test_emd_known_answer_gpu():
    """extensions/emd/test_emd_loss.py:7-44 on the GPU path."""
    from point_dae_amd.emd import earth_mover_distance, EarthMoverDistanceFunction
    p1 = torch.tensor([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]] * 3, device="cuda", requires_grad=True)
    p2 = torch.tensor([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]] * 3, device="cuda", requires_grad=True)
    cost = EarthMoverDistanceFunction.apply(p1, p2)
    np.testing.assert_allclose(host(cost), 0.71, rtol=1e-4)
    cost.sum().backward()
    e1 = 2 * (p1.detach()[:, [0, 1]] - p2.detach()[:, [1, 0]])
    np.testing.assert_allclose(host(p1.grad), host(e1), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(host(p2.grad), host(-e1[:, [1, 0]]), rtol=1e-3, atol=1e-5)
    assert abs(earth_mover_distance()(p1, p2).item() - 0.355) < 1e-4


@pytest.mark.parametrize("B,n,m", [(4, 32, 32), (2, 256, 256), (2, 100, 300), (1, 1024, 1024), (2, 300, 100),
                                   (9, 32, 32), (5, 17, 32), (3, 64, 48), (3, 20, 64), (1, 1, 1), (2, 65, 64), (1, 2048, 512)])
def test_emd_vs_oracle(oracle_ops, B, n, m):
    from point_dae_amd import emd
    a = make_clouds(81, B, n)
    b = make_clouds(82, B, m)
    want_match = oracle_ops.emd_approxmatch(a, b)
    match = emd.approxmatch_forward(dev(a), dev(b))
    # only difference: v_exp_f32 (as the reference's __expf) vs libm expf in the oracle
    np.testing.assert_allclose(host(match), want_match, rtol=2e-3, atol=2e-6)
    # cost / grads are evaluated on the SAME match -> identical summation order -> bit exact
    md = dev(want_match)
    cost = emd.matchcost_forward(dev(a), dev(b), md)
    np.testing.assert_array_equal(host(cost), oracle_ops.emd_matchcost(a, b, want_match))
    gc = np.random.default_rng(83).normal(size=B).astype(np.float32)
    g1, g2 = emd.matchcost_backward(dev(gc), dev(a), dev(b), md)
    w1, w2 = oracle_ops.emd_matchcost_grad(gc, a, b, want_match)
    np.testing.assert_array_equal(host(g1), w1)
    np.testing.assert_array_equal(host(g2), w2)
    # end to end
    got = emd.earth_mover_distance()(dev(a), dev(b)).item()
    want = oracle_ops.earth_mover_distance(a, b)
    assert abs(got - want) <= 1e-3 * abs(want)


def test_emd_forms_agree(oracle_ops):
    """approxmatch's three forms on one input: per-phase launches (scratch given), the one-work-group kernel (no
    scratch: pdae_emd_approxmatch's temp is nullable) -- same `match` up to the summation order of the sliced sums."""
    from point_dae_amd import _lib
    a, b = dev(make_clouds(91, 2, 256)), dev(make_clouds(92, 2, 200))
    B, n, m = 2, 256, 200
    out = []
    for temp in (torch.empty(B, 2 * (n + m), device='cuda'), None):
        match = torch.full((B, m, n), float('nan'), device='cuda')
        _lib.call('pdae_emd_approxmatch', a, B, n, m, _lib.ptr(a), _lib.ptr(b), _lib.ptr(match), _lib.ptr(temp))
        out.append(host(match))
    np.testing.assert_allclose(out[0], out[1], rtol=2e-3, atol=2e-6)


# ------------------------------------------------------ error behaviour ----
def test_ops_reject_bad_inputs():
    from point_dae_amd import chamfer_dist, pointnet2_utils as pu
    from point_dae_amd.knn_cuda import knn
    x = torch.rand(2, 64, 3)
    with pytest.raises(RuntimeError):
        pu.furthest_point_sample(x, 8)                       # CPU tensor: no CPU path
    xd = x.cuda()
    with pytest.raises(RuntimeError):
        pu.furthest_point_sample(xd.double(), 8)             # dtype
    with pytest.raises(RuntimeError):
        pu.furthest_point_sample(xd.transpose(1, 2), 8)      # not contiguous
    with pytest.raises(RuntimeError):
        knn(xd, xd[:, :4].contiguous(), 65)                  # k > 64 unsupported -> loud
    with pytest.raises(RuntimeError):
        knn(xd, xd[:, :4].contiguous(), 100)                 # k > n
    with pytest.raises(RuntimeError):
        chamfer_dist.forward(xd, xd[:, :0].contiguous())     # empty cloud
    # empty batch is fine
    e = torch.empty(0, 64, 3, device="cuda")
    assert pu.furthest_point_sample(e, 8).shape == (0, 8)


# ------------------------------------------------ committed op vectors ----
class _HipOps:
    """The HIP path behind the oracle's call signatures (numpy in, numpy out)."""

    def furthest_point_sample(self, xyz, npoint, return_centres=False):
        from point_dae_amd import pointnet2_utils as pu
        idx, ctr = pu.furthest_point_sample_with_centres(dev(xyz), npoint)
        return (host(idx), host(ctr)) if return_centres else host(idx)

    def knn(self, ref, query, k):
        from point_dae_amd.knn_cuda import KNN
        dist, idx = KNN(k, transpose_mode=True)(dev(ref), dev(query))
        return host(dist), host(idx)

    def ball_query(self, radius, nsample, xyz, new_xyz):
        from point_dae_amd import pointnet2_utils as pu
        return host(pu.ball_query(radius, nsample, dev(xyz), dev(new_xyz)))

    def chamfer_forward(self, a, b):
        from point_dae_amd import chamfer_dist
        return tuple(host(t) for t in chamfer_dist.forward(dev(a), dev(b)))

    def chamfer_backward(self, a, b, i1, i2, g1, g2):
        from point_dae_amd import chamfer_dist
        return tuple(host(t) for t in chamfer_dist.backward(dev(a), dev(b), dev(i1), dev(i2), dev(g1), dev(g2)))

    def chamfer_distance_l2(self, a, b):
        from point_dae_amd.chamfer_dist import ChamferDistanceL2
        return ChamferDistanceL2()(dev(a), dev(b)).item()

    def chamfer_distance_l1(self, a, b):
        from point_dae_amd.chamfer_dist import ChamferDistanceL1
        return ChamferDistanceL1()(dev(a), dev(b)).item()

    def earth_mover_distance(self, a, b):
        from point_dae_amd import emd
        return emd.earth_mover_distance()(dev(a), dev(b)).item()

    def emd_approxmatch(self, a, b):
        from point_dae_amd import emd
        return host(emd.approxmatch_forward(dev(a), dev(b)))

    def emd_matchcost(self, a, b, match):
        from point_dae_amd import emd
        return host(emd.matchcost_forward(dev(a), dev(b), dev(match)))


def test_op_golden_vectors():
    """The committed oracle vectors (tests/golden/ops_oracle.npz) reproduced by
    the HIP kernels without the oracle in the loop: indices and fixed-order fp32 values bit-exact,
    atomic-order gradients and EMD (hardware exp) within the tolerances of the tests above."""
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, here)
    import make_op_fixtures
    want = np.load(os.path.join(here, 'ops_oracle.npz'))
    got = make_op_fixtures.cases(_HipOps())
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        g, w = np.asarray(got[k]), want[k]
        if k.startswith('emd'):
            np.testing.assert_allclose(g, w, rtol=3e-3, atol=1e-5, err_msg=k)
        elif k.endswith(('_l1', '_l2')):
            np.testing.assert_allclose(g, w, rtol=1e-5, err_msg=k)        # north-star bar
        elif k.endswith(('_ga', '_gb')) and '1024' in k:
            np.testing.assert_allclose(g, w, rtol=1e-4, atol=1e-5, err_msg=k)   # atomic scatter order
        else:
            np.testing.assert_array_equal(g, w, err_msg=k)


@pytest.mark.parametrize('B,n,m,c', [(4, 1024, 256, 64), (2, 2048, 128, 256), (3, 77, 5, 3), (1, 300, 2, 8),
                                     (2, 513, 3000, 16)])
def test_three_nn_interpolate_match_oracle(oracle_ops, B, n, m, c):
    """three_nn distances + indices and three_interpolate bit-exact against the oracle (same
    comparison sequence, same left-to-right rounded sum); the gradient within fp32 reassociation."""
    import torch
    from point_dae_amd import pointnet2_utils as P
    rng = np.random.default_rng(B * 1000 + n)
    unknown = rng.uniform(-1, 1, (B, n, 3)).astype(np.float32)
    known = rng.uniform(-1, 1, (B, m, 3)).astype(np.float32)
    if m > 8:
        known[0, 7] = known[0, 3]
    dist, idx = P.three_nn(torch.from_numpy(unknown).cuda(), torch.from_numpy(known).cuda())
    d2, widx = oracle_ops.three_nn(unknown, known)
    assert np.array_equal(idx.cpu().numpy(), widx)
    assert np.array_equal(dist.cpu().numpy(), np.sqrt(d2))
    if m < 3:
        return
    pts = rng.normal(size=(B, c, m)).astype(np.float32)
    w = rng.uniform(0, 1, (B, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    f = torch.from_numpy(pts).cuda().requires_grad_(True)
    out = P.three_interpolate(f, idx, torch.from_numpy(w).cuda())
    assert np.array_equal(out.detach().cpu().numpy(), oracle_ops.three_interpolate(pts, widx, w))
    g = rng.normal(size=(B, c, n)).astype(np.float32)
    out.backward(torch.from_numpy(g).cuda())
    want = oracle_ops.three_interpolate_grad(g, widx, w, m)
    assert np.abs(f.grad.cpu().numpy() - want).max() <= 1e-5 * max(np.abs(want).max(), 1.0)
