"""CPU tests: pin the oracle (oracle/pdae_oracle.c) before anything trusts it.

* against the reference's own known-answer / gradcheck tests where it has any
  (extensions/emd/test_emd_loss.py:7-44, extensions/chamfer_dist/test.py:23-29),
* against brute-force numpy definitions of each operator otherwise (FPS, kNN,
  ball query, grouping have no vectors upstream: "parity unpinned").
"""
import numpy as np
import pytest

from conftest import make_clouds


def _sq(a, b):
    d = a[:, None, :] - b[None, :, :]
    d = (d * d).astype(np.float32)
    return (d[..., 0] + d[..., 1]) + d[..., 2]


# ------------------------------------------------------------------ FPS ----
def _fps_definition(p, m):
    """argmax of the running min distance, start 0, skipping |p|^2 <= 1e-3;
    valid on tie-free data."""
    n = len(p)
    mag = ((p[:, 0] * p[:, 0]) + (p[:, 1] * p[:, 1])) + (p[:, 2] * p[:, 2])
    ok = ~(mag.astype(np.float64) <= 1e-3)
    temp = np.full(n, 1e10, np.float32)
    out, old = [0], 0
    for _ in range(1, m):
        d = _sq(p, p[old:old + 1])[:, 0]
        temp = np.where(ok, np.minimum(temp, d), temp)
        cand = np.where(ok, temp, -1.0)
        old = int(np.argmax(cand))
        out.append(old)
    return np.array(out, np.int32)


@pytest.mark.parametrize("N,m", [(1024, 64), (1024, 512), (512, 128), (2048, 128), (100, 17)])
def test_fps_matches_definition(oracle_ops, N, m):
    x = make_clouds(1, 3, N, "shapes" if N >= 512 else "uniform")
    idx, ctr = oracle_ops.furthest_point_sample(x, m, return_centres=True)
    for b in range(x.shape[0]):
        np.testing.assert_array_equal(idx[b], _fps_definition(x[b], m))
        np.testing.assert_array_equal(ctr[b], x[b][idx[b]])


def test_fps_skips_origin_ball_and_starts_at_zero(oracle_ops):
    x = make_clouds(2, 2, 256)
    x[0, 5:40] *= 0.01          # inside the 1e-3 sphere: never selected
    x[1, 0] = 0.0               # index 0 is always the first sample, even if skipped later
    idx = oracle_ops.furthest_point_sample(x, 200)
    assert (idx[:, 0] == 0).all()
    assert not np.isin(idx[0, 1:], np.arange(5, 40)).any()
    np.testing.assert_array_equal(idx[0], _fps_definition(x[0], 200))


def _tie_rank(k, bs):
    """Order in which the reference's block resolves equal distances: the
    strided scan keeps the first k inside a thread (strict '>'), and the
    shared-memory tree (sampling_gpu.cu:118-171, __update :62-68) pairs
    (tid, tid+s) for s = bs/2 ... 1 keeping the lower slot, so between two
    threads the winner has a 0 at the lowest bit where their tids differ --
    i.e. ascending BIT-REVERSED tid, then ascending k."""
    bits = bs.bit_length() - 1
    tid = k % bs
    rev = int(format(tid, "0{}b".format(bits))[::-1], 2) if bits else 0
    return (rev, k // bs)


def test_fps_tie_rule_is_thread_layout(oracle_ops):
    """Duplicated points: the winner is neither the smallest k nor the smallest
    tid; it follows _tie_rank."""
    N = 1024
    rng = np.random.default_rng(0)
    for trial in range(20):
        x = np.zeros((1, N, 3), np.float32)
        x[0, :] = [0.5, 0.5, 0.5]
        tied = rng.choice(np.arange(1, N), size=rng.integers(2, 6), replace=False)
        x[0, tied] = [-0.5, -0.5, -0.5]
        idx = oracle_ops.furthest_point_sample(x, 2)
        expect = min(tied, key=lambda k: _tie_rank(int(k), 512))
        assert idx[0, 1] == expect, (tied, idx[0, 1], expect)
    # explicit instance: k=188 and k=700 share tid 188 (first k wins inside the
    # thread); tid 300 beats tid 188 because bit 4 is the lowest differing bit
    x = np.zeros((1, N, 3), np.float32)
    x[0, :] = [0.5, 0.5, 0.5]
    x[0, [188, 700]] = [-0.5, -0.5, -0.5]
    assert oracle_ops.furthest_point_sample(x, 2)[0, 1] == 188
    x[0, 300] = [-0.5, -0.5, -0.5]
    assert oracle_ops.furthest_point_sample(x, 2)[0, 1] == 300
    # all points skipped -> index 0 forever
    z = np.zeros((1, 64, 3), np.float32)
    assert (oracle_ops.furthest_point_sample(z, 8) == 0).all()


def test_opt_n_threads(oracle_ops):
    # cuda_utils.h:15-21
    assert [oracle_ops.opt_n_threads(n) for n in (1, 2, 3, 64, 100, 512, 1024, 8192)] == \
        [1, 2, 2, 64, 64, 512, 512, 512]


# ------------------------------------------------------------------ kNN ----
@pytest.mark.parametrize("N,G,k", [(1024, 64, 32), (2048, 128, 32), (77, 9, 5), (40, 3, 40)])
def test_knn_matches_stable_sort(oracle_ops, N, G, k):
    x = make_clouds(3, 2, N, "shapes" if N >= 512 else "uniform")
    ctr = x[:, :G].copy()
    dist, idx, nbr = oracle_ops.knn(x, ctr, k, return_nbr=True)
    assert idx.dtype == np.int64 and idx.shape == (2, G, k)
    for b in range(2):
        d2 = _sq(ctr[b], x[b])
        order = np.argsort(d2, axis=1, kind="stable")[:, :k]
        np.testing.assert_array_equal(idx[b], order)
        np.testing.assert_array_equal(dist[b], np.sqrt(np.take_along_axis(d2, order, 1)))
        np.testing.assert_array_equal(nbr[b], x[b][order] - ctr[b][:, None, :])
    assert (idx[:, :, 0] == np.arange(G)[None]).all()      # the centre is its own 1st neighbour


def test_knn_ties_keep_earlier_index(oracle_ops):
    x = make_clouds(4, 1, 64)
    x[0, 10] = x[0, 3]
    x[0, 50] = x[0, 3]
    _, idx = oracle_ops.knn(x, x[:, 3:4], 8)
    assert list(idx[0, 0, :3]) == [3, 10, 50]


# ----------------------------------------------------------- ball query ----
def _ball_definition(xyz, new_xyz, r, ns):
    out = np.zeros((len(new_xyz), ns), np.int32)
    d2 = _sq(new_xyz, xyz)
    r2 = np.float32(r) * np.float32(r)
    for j in range(len(new_xyz)):
        hits = np.nonzero(d2[j] < r2)[0][:ns]
        if len(hits):
            out[j, :] = hits[0]
            out[j, :len(hits)] = hits
    return out


@pytest.mark.parametrize("N,m,r,ns", [(1024, 512, 0.2, 32), (512, 128, 0.4, 64), (100, 7, 0.05, 8)])
def test_ball_query_matches_definition(oracle_ops, N, m, r, ns):
    x = make_clouds(5, 2, N, "shapes" if N >= 512 else "uniform")
    new_xyz = x[:, :m].copy()
    new_xyz[0, 0] = [5, 5, 5]          # empty ball -> zeros
    idx = oracle_ops.ball_query(r, ns, x, new_xyz)
    for b in range(2):
        np.testing.assert_array_equal(idx[b], _ball_definition(x[b], new_xyz[b], r, ns))
    assert (idx[0, 0] == 0).all()


# -------------------------------------------------------- group / gather ----
def test_group_and_gather(oracle_ops):
    rng = np.random.default_rng(6)
    f = rng.normal(size=(2, 5, 50)).astype(np.float32)
    idx = rng.integers(0, 50, (2, 7, 4)).astype(np.int32)
    out = oracle_ops.grouping_operation(f, idx)
    for b in range(2):
        np.testing.assert_array_equal(out[b], f[b][:, idx[b]])
    go = rng.normal(size=out.shape).astype(np.float32)
    g = oracle_ops.grouping_operation_grad(go, idx, 50)
    ref = np.zeros_like(f, dtype=np.float64)
    for b in range(2):
        for c in range(5):
            np.add.at(ref[b, c], idx[b].ravel(), go[b, c].ravel())
    np.testing.assert_allclose(g, ref, rtol=1e-5, atol=1e-6)
    idx2 = rng.integers(0, 50, (2, 9)).astype(np.int32)
    out2 = oracle_ops.gather_operation(f, idx2)
    for b in range(2):
        np.testing.assert_array_equal(out2[b], f[b][:, idx2[b]])
    g2 = oracle_ops.gather_operation_grad(out2, idx2, 50)
    assert g2.shape == f.shape


# -------------------------------------------------------------- Chamfer ----
@pytest.mark.parametrize("n,m", [(32, 32), (64, 128), (600, 1100), (3, 1)])
def test_chamfer_forward_matches_definition(oracle_ops, n, m):
    a = make_clouds(7, 3, n)
    b = make_clouds(8, 3, m)
    d1, d2, i1, i2 = oracle_ops.chamfer_forward(a, b)
    for c in range(3):
        dd = _sq(a[c], b[c])
        np.testing.assert_array_equal(i1[c], dd.argmin(1))     # lowest index on ties
        np.testing.assert_array_equal(d1[c], dd.min(1))
        np.testing.assert_array_equal(i2[c], dd.argmin(0))
        np.testing.assert_array_equal(d2[c], dd.min(0))


def test_chamfer_ties_lowest_index_across_chunks(oracle_ops):
    a = np.zeros((1, 2, 3), np.float32)
    b = np.ones((1, 1300, 3), np.float32)      # every candidate ties, 3 chunks of 512
    _, _, i1, _ = oracle_ops.chamfer_forward(a, b)
    assert (i1 == 0).all()


def test_chamfer_gradcheck_double(oracle_ops):
    """extensions/chamfer_dist/test.py:23-29: gradcheck of ChamferFunction on
    rand(4,64,3) / rand(4,128,3) in double -- analytic backward vs central
    differences of sum(w1*dist1) + sum(w2*dist2)."""
    rng = np.random.default_rng(9)
    x = rng.random((4, 64, 3))
    y = rng.random((4, 128, 3))
    w1 = rng.random((4, 64))
    w2 = rng.random((4, 128))

    def f(x, y):
        d1, d2, _, _ = oracle_ops.chamfer_forward(x, y)
        return (w1 * d1).sum() + (w2 * d2).sum()

    _, _, i1, i2 = oracle_ops.chamfer_forward(x, y)
    gx, gy = oracle_ops.chamfer_backward(x, y, i1, i2, w1, w2)
    eps = 1e-6
    for (arr, g) in ((x, gx), (y, gy)):
        for _ in range(40):
            pos = tuple(rng.integers(0, s) for s in arr.shape)
            old = arr[pos]
            arr[pos] = old + eps
            fp = f(x, y)
            arr[pos] = old - eps
            fm = f(x, y)
            arr[pos] = old
            num = (fp - fm) / (2 * eps)
            assert abs(num - g[pos]) <= 1e-5 + 1e-3 * abs(num), (pos, num, g[pos])


def test_chamfer_losses(oracle_ops):
    a = make_clouds(10, 2, 40)
    b = make_clouds(11, 2, 50)
    d1, d2, _, _ = oracle_ops.chamfer_forward(a, b)
    assert np.isclose(oracle_ops.chamfer_distance_l2(a, b), d1.mean() + d2.mean())
    assert np.isclose(oracle_ops.chamfer_distance_l1(a, b),
                      (np.sqrt(d1).mean() + np.sqrt(d2).mean()) / 2)


# ------------------------------------------------------------------ EMD ----
def test_emd_known_answer(oracle_ops):
    """extensions/emd/test_emd_loss.py:7-44: 2-point clouds, optimal assignment
    p1[0]<->p2[1], p1[1]<->p2[0], cost 0.30 + 0.41 = 0.71 per cloud, and the
    gradient of that expression."""
    p1 = np.array([[[1.7, -0.1, 0.1], [0.1, 1.2, 0.3]]] * 3, np.float32)
    p2 = np.array([[[0.3, 1.8, 0.2], [1.2, -0.2, 0.3]]] * 3, np.float32)
    match = oracle_ops.emd_approxmatch(p1, p2)
    cost = oracle_ops.emd_matchcost(p1, p2, match)
    np.testing.assert_allclose(cost, 0.71, rtol=1e-5)
    expected = ((p1[:, 0] - p2[:, 1]) ** 2).sum(1) + ((p1[:, 1] - p2[:, 0]) ** 2).sum(1)
    np.testing.assert_allclose(cost, expected, rtol=1e-5)
    g1, g2 = oracle_ops.emd_matchcost_grad(np.ones(3, np.float32), p1, p2, match)
    np.testing.assert_allclose(g1[:, 0], 2 * (p1[:, 0] - p2[:, 1]), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g1[:, 1], 2 * (p1[:, 1] - p2[:, 0]), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g2[:, 0], 2 * (p2[:, 0] - p1[:, 1]), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g2[:, 1], 2 * (p2[:, 1] - p1[:, 0]), rtol=1e-4, atol=1e-6)
    # module semantics: mean over batch of cost / n  (emd.py:46-49)
    assert np.isclose(oracle_ops.earth_mover_distance(p1, p2), 0.71 / 2, rtol=1e-5)


def test_emd_match_is_a_transport_plan(oracle_ops):
    a = make_clouds(12, 2, 64)
    b = make_clouds(13, 2, 64)
    match = oracle_ops.emd_approxmatch(a, b)          # (B, m, n)
    assert (match >= 0).all()
    np.testing.assert_allclose(match.sum(1), 1.0, atol=2e-2)   # every xyz1 point fully shipped
    np.testing.assert_allclose(match.sum(2), 1.0, atol=2e-2)
    # identical clouds -> near-zero cost
    m2 = oracle_ops.emd_approxmatch(a, a)
    assert (oracle_ops.emd_matchcost(a, a, m2) / 64 < 2e-3).all()


def test_oracle_reproduces_committed_op_vectors(oracle_ops):
    """tests/golden/ops_oracle.npz (made by tests/golden/make_op_fixtures.py): the oracle's answers on
    the SURVEY §8c cases -- FPS incl. the origin-skip and duplicate clouds, kNN, ball query with an
    empty ball, Chamfer at the three cfg sizes, EMD incl. the reference's known answer."""
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    sys.path.insert(0, here)
    import make_op_fixtures
    want = np.load(os.path.join(here, 'ops_oracle.npz'))
    got = make_op_fixtures.cases(oracle_ops)
    assert sorted(got) == sorted(want.files)
    for k in want.files:
        np.testing.assert_array_equal(np.asarray(got[k]), want[k], err_msg=k)
    assert abs(float(want['emd_two_point_cost']) - 0.355) < 1e-4     # extensions/emd/test_emd_loss.py
    assert (want['ball_512_32_idx'][0, 0] == 0).all()                  # empty ball: all slots stay 0 (ball_query_gpu.cu)


def test_three_nn_and_interpolate_against_definitions(oracle_ops):
    """three_nn: the three smallest squared distances, ascending, earliest index on ties; interpolate:
    the weighted sum of the three gathered features; its gradient: the adjoint scatter."""
    rng = np.random.default_rng(5)
    B, n, m, c = 3, 70, 41, 6
    unknown = rng.uniform(-1, 1, (B, n, 3)).astype(np.float32)
    known = rng.uniform(-1, 1, (B, m, 3)).astype(np.float32)
    known[0, 7] = known[0, 3]                                  # a duplicated known point: tie
    d2, idx = oracle_ops.three_nn(unknown, known)
    diff = unknown[:, :, None, :] - known[:, None, :, :]
    full = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    order = np.argsort(full, axis=2, kind='stable')[:, :, :3]
    assert np.array_equal(idx, order.astype(np.int32))
    assert np.array_equal(d2, np.take_along_axis(full, order, 2))
    pts = rng.normal(size=(B, c, m)).astype(np.float32)
    w = rng.uniform(0, 1, (B, n, 3)).astype(np.float32)
    out = oracle_ops.three_interpolate(pts, idx, w)
    want = sum(np.take_along_axis(pts, np.broadcast_to(idx[:, None, :, k], (B, c, n)), 2).astype(np.float64) *
               w[:, None, :, k] for k in range(3))
    assert np.allclose(out, want, rtol=1e-6, atol=1e-6)
    g = rng.normal(size=(B, c, n)).astype(np.float32)
    gp = oracle_ops.three_interpolate_grad(g, idx, w, m)
    # adjoint identity: <out(pts), g> == <pts, grad(g)>
    assert abs((out.astype(np.float64) * g).sum() - (pts.astype(np.float64) * gp).sum()) < 1e-3
    # fewer than three known points: +inf and index 0 in the unused slots
    d2s, idxs = oracle_ops.three_nn(unknown[:, :5], known[:, :2])
    assert np.isinf(d2s[..., 2]).all() and (idxs[..., 2] == 0).all()
