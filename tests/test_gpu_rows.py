"""GPU numerics of the Transformer-block GEMM family (csrc/rows_gemm.hip): every tile shape, both
weight layouts, the fused epilogues, split-K slabs and the grouped weight-gradient launch, against
fp64 PyTorch compositions of the same ops (tolerance: fp32 GEMM reassociation, 2e-5 of the output
scale) -- and bit-identical results run to run (no atomics anywhere)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from point_dae_amd import _lib
    return _lib


def _close(a, b, tol=2e-5):
    scale = b.abs().max().item() + 1e-12
    err = (a - b.to(a.dtype)).abs().max().item() / scale
    assert err <= tol, err


def _gemm(L, x, w, w_kn, bias, epi, z, cfg, splits, stream_blocks=0):
    M, K = x.shape
    N = w.shape[1] if w_kn else w.shape[0]
    y = torch.full((max(splits, 1), M, N), float('nan'), device='cuda')
    L.call('pdae_rows_gemm', x, M, N, K, x.data_ptr(), w.data_ptr(), int(w_kn), L.ptr(bias), epi, L.ptr(z),
           y.data_ptr(), cfg, splits, stream_blocks)
    return y


SHAPES = [(2944, 1152, 384), (1664, 384, 384), (4096, 1536, 384), (2944, 384, 1536), (8192, 384, 1152),
          (1000, 96, 384), (37, 128, 132), (300, 1536, 388), (5248, 96, 384), (8192, 384, 128)]


@pytest.mark.parametrize('M,N,K', SHAPES)
@pytest.mark.parametrize('cfg', list(range(8)) + [-1])
def test_rows_gemm_store(M, N, K, cfg):
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5          # (out, in)
    b = torch.randn(N, device='cuda', generator=g)
    ref = F.linear(x.double(), w.double())
    _close(_gemm(L, x, w, 0, None, 0, None, cfg, 1)[0], ref)
    _close(_gemm(L, x, w, 0, b, 0, None, cfg, 1)[0], ref + b.double())
    _close(_gemm(L, x, w, 0, b, 1, None, cfg, 1)[0], F.relu(ref + b.double()))
    # the same weight as the data-gradient operand: dX[M,K] = dY[M,N] . W[N,K]
    dy = torch.randn(M, N, device='cuda', generator=g)
    if N % 4 == 0:
        _close(_gemm(L, dy, w, 1, None, 0, None, cfg, 1)[0], dy.double() @ w.double())


@pytest.mark.parametrize('M,N,K', [(2944, 1536, 384), (8192, 1536, 384), (333, 1536, 384), (2944, 128, 4)])
@pytest.mark.parametrize('cfg', [0, 1, 4, 5, 7, -1])
def test_rows_gemm_gelu_pair(M, N, K, cfg):
    """fc1 + GELU forward (Z and H in one pass) and its backward twin dz = (dh . W2) * GELU'(Z)."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(M, K, device='cuda', generator=g)
    w1 = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b1 = torch.randn(N, device='cuda', generator=g)
    gp = torch.full((M, N), float('nan'), device='cuda')     # GELU'(z): the factor the backward needs
    h = _gemm(L, x, w1, 0, b1, 2, gp, cfg, 1)[0]
    zr = F.linear(x.double(), w1.double(), b1.double()).requires_grad_(True)
    hr = F.gelu(zr)
    _close(h, hr.detach())
    hr.sum().backward()
    _close(gp, zr.grad, 1e-5)
    # backward: da (M, C) -> dh = da . W2 (W2 is (C, N)), dz = dh * gelu'(z)
    C = 384
    da = torch.randn(M, C, device='cuda', generator=g)
    w2 = torch.randn(C, N, device='cuda', generator=g) / N ** 0.5
    dz = _gemm(L, da, w2, 1, None, 3, gp, cfg, 1)[0]
    zz = zr.detach().clone().requires_grad_(True)
    F.gelu(zz).backward(da.double() @ w2.double())
    _close(dz, zz.grad)


@pytest.mark.parametrize('cfg', [0, -1])
def test_gelu_pair_keeps_the_limits_at_plus_infinity(cfg):
    """An fc1 output that overflowed to +inf leaves GELU = inf and GELU' = 1 (what erff / expf give and what nn.GELU's
    autograd gives), not NaN in both (common.h gelu_pair_f).  The overflow enters through the bias, which the
    epilogue adds behind the accumulation in either GEMM arithmetic."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(9)
    M, N, K = 256, 128, 64
    x = torch.randn(M, K, device='cuda', generator=g)
    w1 = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b1 = torch.randn(N, device='cuda', generator=g)
    b1[3] = float('inf')
    b1[5] = 3e38
    gp = torch.full((M, N), float('nan'), device='cuda')
    h = _gemm(L, x, w1, 0, b1, 2, gp, cfg, 1)[0]
    assert torch.isposinf(h[:, 3]).all() and torch.equal(gp[:, 3], torch.ones(M, device='cuda'))
    assert torch.equal(gp[:, 5], torch.ones(M, device='cuda')) and torch.isfinite(h[:, 5]).all()
    keep = [c for c in range(N) if c not in (3, 5)]
    zr = F.linear(x.double(), w1.double(), b1.double())[:, keep]
    _close(h[:, keep], F.gelu(zr))


@pytest.mark.parametrize('M,N,K', [(2944, 384, 1536), (1664, 384, 1152), (8192, 384, 1536), (100, 96, 384)])
@pytest.mark.parametrize('w_kn', [0, 1])
@pytest.mark.parametrize('splits', [2, 3, 4])
def test_rows_gemm_split_slabs(M, N, K, w_kn, splits):
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(11)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(K, N, device='cuda', generator=g) / K ** 0.5 if w_kn else \
        torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    ref = x.double() @ (w.double() if w_kn else w.double().t())
    for cfg in (0, 1, 6, -1):
        y = _gemm(L, x, w, w_kn, None, 0, None, cfg, splits)
        assert torch.isfinite(y).all()
        _close(y.sum(0), ref)
        y2 = _gemm(L, x, w, w_kn, None, 0, None, cfg, splits)
        assert torch.equal(y, y2)                                  # deterministic


@pytest.mark.parametrize('M,N,K', [(2944, 384, 1536), (1664, 384, 384), (8192, 384, 1152), (4096, 384, 1536),
                                   (1000, 96, 384), (300, 384, 388)])
@pytest.mark.parametrize('w_kn', [0, 1])
def test_rows_gemm_stream_k(M, N, K, w_kn):
    """Stream-K: equal contiguous ranges of (tile, k-tile) units per block, pieces of a tile in
    consecutive slabs, unused slabs zero-filled; the planned choice and forced grids."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(13)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(K, N, device='cuda', generator=g) / K ** 0.5 if w_kn else \
        torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    ref = x.double() @ (w.double() if w_kn else w.double().t())
    cfg, S, sb = L.rows_gemm_plan(M, N, K, w_kn, True)
    y = _gemm(L, x, w, w_kn, None, 0, None, cfg, S, sb)
    assert torch.isfinite(y).all()
    _close(y.sum(0), ref)
    for P in (8, 64, 256, 512):
        for cfg2 in (3, 1, 0):
            try:
                y = _gemm(L, x, w, w_kn, None, 0, None, cfg2, 8, P)
            except RuntimeError:            # fewer units than blocks, or more pieces than 8 slabs
                continue
            assert torch.isfinite(y).all()
            _close(y.sum(0), ref)
            assert torch.equal(y, _gemm(L, x, w, w_kn, None, 0, None, cfg2, 8, P))


def test_rows_gemm_plan_and_errors():
    L = _lib()
    for M, N, K in SHAPES:
        for w_kn in (0, 1):
            cfg, s, sb = L.rows_gemm_plan(M, N, K, w_kn, True)
            assert (0 <= cfg < 8 or 16 <= cfg < 20) and 1 <= s <= 4 and sb % 8 == 0
            assert L.rows_gemm_plan(M, N, K, w_kn, False)[1:] == (1, 0)
    x = torch.zeros(8, 6, device='cuda')
    with pytest.raises(RuntimeError):                              # K % 4 != 0
        L.call('pdae_rows_gemm', x, 8, 4, 6, x.data_ptr(), x.data_ptr(), 0, None, 0, None, x.data_ptr(), -1, 1, 0)
    with pytest.raises(RuntimeError):                              # slabs with a bias
        L.call('pdae_rows_gemm', x, 8, 4, 8, x.data_ptr(), x.data_ptr(), 0, x.data_ptr(), 0, None, x.data_ptr(), -1, 2, 0)
    # M = 0 is a no-op
    L.call('pdae_rows_gemm', x, 0, 4, 8, None, None, 0, None, 0, None, None, -1, 1, 0)


def _wgrad(L, M, dims, bias_flags, gen):
    dys = [torch.randn(M, n, device='cuda', generator=gen) for n, _ in dims]
    xs = [torch.randn(M, k, device='cuda', generator=gen) for _, k in dims]
    Ns, Ks = [n for n, _ in dims], [k for _, k in dims]
    floats = L.rows_wgrad_workspace(M, Ns, Ks)
    ws = torch.full((max(floats, 1),), float('nan'), device='cuda')
    dws = [torch.full((n, k), float('nan'), device='cuda') for n, k in dims]
    dbs = [torch.full((n,), float('nan'), device='cuda') if f else None for (n, _), f in zip(dims, bias_flags)]
    L.rows_wgrad(dys[0], M, dys, xs, dws, dbs, ws)
    return dys, xs, dws, dbs, floats


@pytest.mark.parametrize('M', [1664, 2944, 8192, 100, 16, 33])
def test_rows_wgrad_block_group(M):
    """The four Linear layers of a Transformer block in one grouped launch."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(M)
    dims = [(1152, 384), (384, 384), (1536, 384), (384, 1536)]
    flags = [False, False, True, False]
    dys, xs, dws, dbs, splits = _wgrad(L, M, dims, flags, g)
    for dy, x, dw, db in zip(dys, xs, dws, dbs):
        _close(dw, dy.double().t() @ x.double(), 5e-5)
        if db is not None:
            _close(db, dy.double().sum(0), 5e-5)
    g2 = torch.Generator(device='cuda').manual_seed(M)
    again = _wgrad(L, M, dims, flags, g2)
    for a, b in zip(dws, again[2]):
        assert torch.equal(a, b)                                   # fixed reduction order


@pytest.mark.parametrize('dims,flags', [([(96, 384)], [True]), ([(128, 4), (384, 128)], [True, True]),
                                        ([(100, 260)], [False]), ([(4096, 2048)], [True]), ([(512, 256)], [False]),
                                        # the 256-wide tile (every K a multiple of 256, not of 384): FoldingNet's layers
                                        ([(512, 512), (4, 512)], [True, True]), ([(100, 512)], [True]),
                                        ([(128, 256), (256, 1024), (36, 768)], [False, True, True])])
def test_rows_wgrad_odd_shapes(dims, flags):
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(3)
    for M in (5248, 77, 1):
        dys, xs, dws, dbs, _ = _wgrad(L, M, dims, flags, g)
        for dy, x, dw, db in zip(dys, xs, dws, dbs):
            _close(dw, dy.double().t() @ x.double(), 5e-5)
            if db is not None:
                _close(db, dy.double().sum(0), 5e-5)


@pytest.mark.parametrize('M,dims', [(262144, [(128, 4)]), (65536, [(128, 128)]), (8192, [(96, 384)]),
                                    (8192, [(128, 4), (384, 128)]), (131072, [(512, 512)]), (65536, [(4, 256)])])
def test_rows_wgrad_long_reduction_narrow_weight(M, dims):
    """Hundreds of partials per tile (the embedder's first conv: one tile, 512 partials): the lane-parallel
    reduction (4 / 8 partial lanes) gives the same sums as fp64 and the same bits twice."""
    L = _lib()
    flags = [True] * len(dims)
    g = torch.Generator(device='cuda').manual_seed(M)
    dys, xs, dws, dbs, _ = _wgrad(L, M, dims, flags, g)
    for dy, x, dw, db in zip(dys, xs, dws, dbs):
        _close(dw, dy.double().t() @ x.double(), 5e-5)
        _close(db, dy.double().sum(0), 5e-5)
    g2 = torch.Generator(device='cuda').manual_seed(M)
    again = _wgrad(L, M, dims, flags, g2)
    for a, b in zip(dws + dbs, again[2] + again[3]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('blocks,enc_m,dec_m,tail_m', [(2, 1664, 2048, 640), (5, 2944, 0, 0), (13, 96, 128, 64)])
def test_rows_wgrad_multi_layers_with_their_own_rows(blocks, enc_m, dec_m, tail_m):
    """pdae_rows_wgrad_multi: the weight gradients of MANY layers in one launch, every layer with its own row count --
    what a stack's backward issues (encoder blocks at B*T_vis rows, decoder blocks at B*G, the trimmed last block's
    three layers at B*tail).  Against fp64 products; the same bits twice (ordered partial-tile reduction); more
    layers than one launch holds are cut into several."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(blocks)
    dims = [(1152, 384), (384, 384), (1536, 384), (384, 1536)]

    def make():
        jobs = []
        for b in range(blocks):
            for i, (n, k) in enumerate(dims):
                m = enc_m if (b % 2 == 0 or not dec_m) else (tail_m if (b == blocks - 1 and i > 0 and tail_m) else dec_m)
                dy = torch.randn(m, n, device='cuda', generator=g)
                x = torch.randn(m, k, device='cuda', generator=g)
                dw = torch.full((n, k), float('nan'), device='cuda')
                db = torch.full((n,), float('nan'), device='cuda') if i == 2 else None
                jobs.append((dy, x, dw, db))
        return jobs
    st = g.get_state()
    jobs = make()
    assert (len(jobs) > L.WGRAD_MULTI_MAX) == (blocks == 13)
    L.rows_wgrad_multi(jobs)
    for dy, x, dw, db in jobs:
        _close(dw, dy.double().t() @ x.double(), 5e-5)
        if db is not None:
            _close(db, dy.double().sum(0), 5e-5)
    g.set_state(st)
    again = make()
    L.rows_wgrad_multi(again)
    for a, b in zip(jobs, again):
        assert torch.equal(a[2], b[2])


@pytest.mark.parametrize('M,N,K', [(4096, 512, 512), (1000, 512, 4), (333, 64, 128)])
def test_rows_gemm_relu_mask_epilogue(M, N, K):
    """epi 4: dX = (dY . W) masked by the sign of the ReLU output the gradient flows into."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(M + N)
    dy = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(K, N, device='cuda', generator=g) * 0.1          # (out=K, in=N): the [K,N] operand
    h = torch.relu(torch.randn(M, N, device='cuda', generator=g))
    h[0, :8] = 0.0                                                   # exact zeros: relu'(0) = 0
    y = torch.full((M, N), float('nan'), device='cuda')
    L.call('pdae_rows_gemm', dy, M, N, K, dy.data_ptr(), w.data_ptr(), 1, None, 4, h.data_ptr(), y.data_ptr(), -1, 1, 0)
    want = (dy.double() @ w.double()) * (h > 0)
    _close(y, want, 2e-5)
    assert (y[h == 0] == 0).all()
    y2 = torch.full((M, N), float('nan'), device='cuda')              # the same product from the [N,K] layout
    wt = w.t().contiguous()
    L.call('pdae_rows_gemm', dy, M, N, K, dy.data_ptr(), wt.data_ptr(), 0, None, 4, h.data_ptr(), y2.data_ptr(), -1, 1, 0)
    _close(y2, want, 2e-5)
    assert (y2[h == 0] == 0).all()


@pytest.mark.parametrize('clouds,coarse,cells,C', [(3, 40, 16, 512), (2, 7, 16, 128), (1, 5, 9, 64), (5, 129, 4, 256),
                                                   (1, 300, 36, 384), (2, 33, 36, 96)])
def test_fold_input_and_grad(clouds, coarse, cells, C):
    """fold_input / fold_input_grad (csrc/folding.hip) against the broadcast formulation in torch:
    forward bit for bit (same association (a + p) + gd), the sums to fp32 accuracy, partial sets
    reproducible."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(clouds * 100 + coarse)
    a = torch.randn(clouds, C, device='cuda', generator=g)
    p = torch.randn(clouds * coarse, C, device='cuda', generator=g)
    gd = torch.randn(cells, C, device='cuda', generator=g)
    rows = clouds * coarse * cells
    h = torch.full((rows, C), float('nan'), device='cuda')
    L.call('pdae_fold_input', a, clouds, coarse, cells, C, a.data_ptr(), p.data_ptr(), gd.data_ptr(), h.data_ptr())
    want = torch.relu((a.view(clouds, 1, 1, C) + p.view(clouds, coarse, 1, C)) + gd.view(1, 1, cells, C)).reshape(rows, C)
    assert torch.equal(h, want)
    d = torch.randn(rows, C, device='cuda', generator=g) * (h > 0)
    parts = L.lib().pdae_fold_input_grad_parts(clouds, coarse)
    assert parts == (clouds * coarse + 63) // 64
    outs = []
    for _ in range(2):
        dp = torch.full((clouds * coarse, C), float('nan'), device='cuda')
        part = torch.full((parts, cells, C), float('nan'), device='cuda')
        L.call('pdae_fold_input_grad', d, clouds, coarse, cells, C, d.data_ptr(), dp.data_ptr(), part.data_ptr())
        outs.append((dp, part))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    dp, part = outs[0]
    d4 = d.double().view(clouds, coarse, cells, C)
    _close(dp, d4.sum(2).reshape(-1, C), 1e-5)
    _close(part.sum(0), d4.sum((0, 1)), 1e-5)


@pytest.mark.parametrize('rows,C', [(5000, 512), (1024, 128), (77, 64), (3000, 1024), (4000, 384), (333, 96)])
def test_fold_out_backward(rows, C):
    """the 512 -> 3(+1) layer backwards in one pass: masked data gradient + ordered weight-gradient partials"""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(rows)
    dy = torch.randn(rows, 4, device='cuda', generator=g)
    dy[:, 3] = 0.0
    h2 = torch.relu(torch.randn(rows, C, device='cuda', generator=g))
    w = torch.randn(4, C, device='cuda', generator=g) * 0.1
    parts = L.lib().pdae_fold_out_backward_parts(rows)
    assert parts == (rows + 1023) // 1024
    res = []
    for _ in range(2):
        d2 = torch.full((rows, C), float('nan'), device='cuda')
        part = torch.full((parts, 4, C), float('nan'), device='cuda')
        L.call('pdae_fold_out_backward', dy, rows, C, dy.data_ptr(), h2.data_ptr(), w.data_ptr(), d2.data_ptr(), part.data_ptr())
        res.append((d2, part))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    _close(res[0][0], (dy.double() @ w.double()) * (h2 > 0), 2e-5)
    _close(res[0][1].sum(0), dy.double().t() @ h2.double(), 2e-5)
    assert (res[0][0][h2 == 0] == 0).all()


@pytest.mark.parametrize('M,N,K,la,lb,bn,bias', [
    (8192, 384, 512, False, True, True, False),      # conv4: compact dY, listed BatchNorm+ReLU input
    (8192, 512, 256, False, True, False, False),     # conv3's local half on the visible groups
    (4096, 256, 256, True, True, False, False),      # the masked groups' Gram matrix (dY = X, one list)
    (16384, 256, 128, False, False, True, True),     # conv2: every row, BatchNorm+ReLU input, bias sums
    (96, 128, 512, True, False, True, True),         # a reduction shorter than one block's range; K = 512
    (4128, 100, 260, True, True, True, True),        # edges in N and K (narrow tile), M = 129 groups
    (32, 384, 768, False, True, False, True)])       # the 384-wide tile, one group
def test_rows_wgrad_listed(M, N, K, la, lb, bn, bias):
    """pdae_rows_wgrad_listed: the embedder's weight gradients (group-listed operands, BatchNorm + ReLU recomputed on X)
    on the grouped kernel, against fp64, and the same bits twice (ordered reduction)."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(M + N)
    G = M // 32
    stored = (G + 7) * 32
    dY = torch.randn(stored if la else M, N, device='cuda', generator=g)
    X = torch.randn(stored if lb else M, K, device='cuda', generator=g)
    ga = torch.randperm(G + 7, device='cuda', generator=g)[:G].int() if la else None
    gb = torch.randperm(G + 7, device='cuda', generator=g)[:G].int() if lb else None
    if la and lb and N == K:                                   # the Gram form: one matrix, one list
        dY, ga = X, gb
    sc = (torch.rand(K, device='cuda', generator=g) + 0.5) if bn else None
    sh = torch.randn(K, device='cuda', generator=g) * 0.3 if bn else None

    def rows(t, lst):
        return t if lst is None else t.view(-1, 32, t.shape[1])[lst.long()].reshape(M, t.shape[1])
    a, b = rows(dY, ga).double(), rows(X, gb).double()
    if bn:
        b = torch.relu(b * sc.double() + sh.double())
    want_w, want_b = a.t() @ b, a.sum(0)
    ws = torch.full((max(L.rows_wgrad_workspace(M, [N], [K]), 1),), float('nan'), device='cuda')
    outs = []
    for _ in range(2):
        dw = torch.full((N, K), float('nan'), device='cuda')
        db = torch.full((N,), float('nan'), device='cuda') if bias else None
        L.call('pdae_rows_wgrad_listed', dY, M, N, K, L.ptr(dY), L.ptr(ga), L.ptr(X), L.ptr(gb), L.ptr(sc), L.ptr(sh),
               L.ptr(dw), L.ptr(db), L.ptr(ws))
        _close(dw, want_w, 5e-5)
        if bias:
            _close(db, want_b, 5e-5)
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and (not bias or torch.equal(outs[0][1], outs[1][1]))
    with pytest.raises(RuntimeError):                              # a list needs whole groups
        L.call('pdae_rows_wgrad_listed', dY, 40, N, K, L.ptr(dY), L.ptr(gb if gb is not None else torch.zeros(2, dtype=torch.int32, device='cuda')),
               L.ptr(X), None, None, None, L.ptr(dw), None, L.ptr(ws))


@pytest.mark.parametrize('arith', [1, 0], ids=['bf16x3', 'f32mfma'])
@pytest.mark.parametrize('M,widths', [(32, (1024, 1024, 1024, 3072)), (128, (384, 1024, 1024, 192)), (5, (8, 12, 4)),
                                      (7, (16, 10, 6))])
def test_mlp_chain_equals_the_layers_one_by_one(M, widths, arith):
    """nn_ops.mlp_chain (Linear / ReLU / ... / Linear as one node: ReLU masks in the data gradients' epilogues, one grouped
    weight-gradient launch, long reductions on a few rows in split-K slabs) against nn_ops.linear layer by layer and torch in fp64."""
    from point_dae_amd import _lib, nn_ops
    torch.manual_seed(M)
    layers = [torch.nn.Linear(a, b).cuda() for a, b in zip(widths[:-1], widths[1:])]
    x = torch.randn(M, widths[0], device='cuda')
    gy = torch.randn(M, widths[-1], device='cuda')

    def run(fn):
        for l in layers:
            l.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        y = fn(xi)
        y.backward(gy)
        return [y, xi.grad] + [p.grad.clone() for l in layers for p in (l.weight, l.bias)]

    def one_by_one(t):
        for i, l in enumerate(layers):
            t = nn_ops.linear(t, l, 'relu' if i + 1 < len(layers) else None)
        return t

    before = _lib.gemm_arith()
    _lib.set_gemm_arith(arith)
    _lib.set_deterministic(True)
    try:
        a = run(lambda t: nn_ops.mlp_chain(t, layers))
        b = run(one_by_one)
    finally:
        _lib.set_deterministic(False)
        _lib.set_gemm_arith(before)
    for i, (u, v) in enumerate(zip(a, b)):
        # (the chain splits long reductions into slabs and groups the weight gradients: other summation orders)
        assert (u.double() - v.double()).norm().item() <= 2e-6 * max(v.double().norm().item(), 1e-30), i
    ref = [torch.nn.Linear(l.in_features, l.out_features).cuda().double() for l in layers]
    for r, l in zip(ref, layers):
        r.load_state_dict({k: v.double() for k, v in l.state_dict().items()})
    x64 = x.double().requires_grad_(True)
    t = x64
    for i, r in enumerate(ref):
        t = r(t)
        if i + 1 < len(ref):
            t = torch.relu(t)
    t.backward(gy.double())
    want = [t, x64.grad] + [p.grad for r in ref for p in (r.weight, r.bias)]
    for u, v in zip(a, want):
        assert (u.double() - v).norm().item() <= 2e-6 * max(v.norm().item(), 1e-30)


@pytest.mark.parametrize('S,M,N', [(8, 32, 1024), (3, 5, 12), (1, 7, 8)])
def test_slab_sum_epi(S, M, N):
    """pdae_slab_sum_epi: split-K slabs added in slab order, + bias, + ReLU or the ReLU-output mask."""
    from point_dae_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(S * 100 + M)
    slabs = torch.randn(S, M, N, device='cuda', generator=g)
    bias = torch.randn(N, device='cuda', generator=g)
    z = torch.randn(M, N, device='cuda', generator=g)
    want = slabs[0].clone()
    for q in range(1, S):
        want = want + slabs[q]
    for epi, b in ((0, None), (0, bias), (1, bias), (4, None)):
        y = torch.full((M, N), float('nan'), device='cuda')
        _lib.call('pdae_slab_sum_epi', slabs, S, M, N, slabs.data_ptr(), b.data_ptr() if b is not None else None, epi,
                  z.data_ptr() if epi == 4 else None, y.data_ptr())
        ref = want + b if b is not None else want
        ref = torch.relu(ref) if epi == 1 else (torch.where(z > 0, ref, torch.zeros_like(ref)) if epi == 4 else ref)
        assert torch.equal(y, ref)
    with pytest.raises(RuntimeError, match='slab_sum_epi'):
        _lib.call('pdae_slab_sum_epi', slabs, S, M, N, slabs.data_ptr(), None, 2, None, z.data_ptr())


def test_few_rows_plan_splits_a_long_reduction():
    """pdae_rows_gemm_plan(may_split = 8): 32 rows against K = 1024 / 3072 come back in more than 4 slabs (exact-split
    arithmetic), every slab at least one 32-deep tile; may_split = 1 keeps the LayerNorm consumers' limit of 4."""
    from point_dae_amd import _lib
    if _lib.gemm_arith() != _lib.GEMM_BF16X3:
        pytest.skip('exact-split arithmetic only')
    for K in (1024, 3072):
        cfg, splits, sb = _lib.rows_gemm_plan(32, 1024, K, False, 8)
        assert 4 < splits <= 8 and sb == 0 and cfg >= 16
        assert _lib.rows_gemm_plan(32, 1024, K, False, 1)[1] <= 4
        assert _lib.rows_gemm_plan(32, 1024, K, False, 0)[1] == 1
