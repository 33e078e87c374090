"""The DGCNN encoder's gfx950 kernels (csrc/dgcnn.hip, include/pdae.h "DGCNN encoder") against the reference's own
formulation written with framework ops in fp32 (models/dgcnn_util.py:7-34 knn / get_graph_feature, :99-136 the
Conv2d -> BatchNorm2d -> LeakyReLU(0.2) -> max over the 20 neighbours of every EdgeConv, conv5's BatchNorm1d ->
LeakyReLU -> max over the points).  Indices bit-exact (the selection is exact on the reference's expression); floating
point 2e-5 relative on activations, 1e-3 relative L2 on gradients (typically 1e-5; tolerances and their reasons in the tests).
The model-level fixture of the live reference is tests/test_gpu_model.py::test_dgcnn_product_model_reproduces_reference_fixture."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
K = 20


def _lib():
    from point_dae_amd import _lib
    return _lib


def _rel(a, b):
    return (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-30)


@pytest.mark.parametrize('B,N,k', [(32, 1024, 20), (2, 300, 20), (3, 2500, 7), (5, 21, 20), (2, 40, 40)])
def test_xyz_topk_selects_the_reference_s_neighbours(B, N, k):
    """pdae_xyz_topk (the first EdgeConv's graph, distances straight from the point rows: no Gram matrix): the values it
    selected from (pd_out) are dgcnn_util.knn's expression of the fp64 Gram matrix to fp32 rounding, idx is torch.topk's
    of those values -- the SET per row equal, best first (ids differ only inside ties) -- and agrees with the two-kernel
    form up to near-ties of the two Gram roundings."""
    _lib()
    from point_dae_amd.point_cae_dgcnn import feature_knn
    g = torch.Generator(device='cuda').manual_seed(N + k)
    x = torch.randn(B * N, 4, device='cuda', generator=g)
    x[:, 3] = 0
    pd = torch.full((B, N, N), float('nan'), device='cuda')
    idx = feature_knn(x, B, N, k, xyz=True, pd_out=pd).long()
    assert torch.isfinite(pd).all()
    xb = x.view(B, N, 4).double()
    xx64 = xb.square().sum(-1)
    ref = xx64.unsqueeze(2) - 2 * xb @ xb.transpose(1, 2) + xx64.unsqueeze(1)      # = -pd of dgcnn_util.knn (:8-10)
    assert (pd.double() - ref).abs().max().item() <= 4e-7 * (xx64.max() * 4).item()
    want_v, want_i = (-pd).topk(k=k, dim=-1)
    assert torch.equal(-pd.gather(-1, idx), want_v)
    assert (idx.sort(-1)[0] == want_i.sort(-1)[0]).float().mean().item() > 0.999
    assert idx.min().item() >= 0 and idx.max().item() < N
    assert (idx.sort(-1)[0].diff(dim=-1) > 0).all()
    idx2 = feature_knn(x, B, N, k).long()                                          # batched Gram GEMM + gram_topk
    assert (idx.sort(-1)[0] == idx2.sort(-1)[0]).float().mean().item() > 0.995


@pytest.mark.parametrize('B,N,C,k', [(3, 1024, 64, 20), (2, 300, 4, 20), (1, 64, 128, 20), (2, 2500, 8, 7), (5, 21, 4, 20)])
def test_gram_topk_selects_the_reference_s_neighbours(B, N, C, k):
    """idx of dgcnn_util.knn given the same Gram matrix and norms: the SET per row equals torch.topk's and the order is
    best first (compared through the selected distances: equal values may swap ids)."""
    L = _lib()
    from point_dae_amd.point_cae_dgcnn import feature_knn
    g = torch.Generator(device='cuda').manual_seed(N + C)
    x = torch.randn(B * N, C, device='cuda', generator=g)
    if C == 4:
        x[:, 3] = 0
    idx = feature_knn(x, B, N, k).long()
    gram = torch.empty(B, N, N, device='cuda')
    L.call('pdae_rows_gemm_batched', x, B, N, N, C, x.data_ptr(), N * C, x.data_ptr(), N * C, gram.data_ptr(), N * N)
    xx = torch.empty(B * N, device='cuda')
    L.call('pdae_rows_sqnorm', x, B * N, C, x.data_ptr(), xx.data_ptr())
    assert torch.allclose(xx, x.square().sum(-1), rtol=1e-6, atol=0)
    xx = xx.view(B, N)
    xx_ref = xx.unsqueeze(1)                                       # (B, 1, N) as dgcnn_util.knn's keepdim sum over channels
    pd = -xx_ref - (-2 * gram) - xx_ref.transpose(2, 1)            # the reference's expression, operand for operand (:10)
    want_v, want_i = pd.topk(k=k, dim=-1)
    got_v = pd.gather(-1, idx)
    assert torch.equal(got_v, want_v)
    assert (idx.sort(-1)[0] == want_i.sort(-1)[0]).float().mean().item() > 0.999     # (ids differ only inside ties)
    assert idx.min().item() >= 0 and idx.max().item() < N
    # distinct neighbours per row
    assert (idx.sort(-1)[0].diff(dim=-1) > 0).all()


@pytest.mark.parametrize('B,N,k', [(3, 1024, 20), (2, 77, 5), (1, 2048, 20), (2, 4096, 3)])
def test_reverse_graph_lists_every_arriving_edge_in_ascending_order(B, N, k):
    L = _lib()
    rng = np.random.default_rng(N)
    idx = np.stack([np.stack([rng.choice(N, size=k, replace=False) for _ in range(N)]) for _ in range(B)]).astype(np.int32)
    idx[:, :, 0] = np.minimum(idx[:, :, 0], 3)                      # hubs: a few points with hundreds of arriving edges
    for b in range(B):                                              # (keep a row's neighbours distinct)
        for r in range(N):
            if len(set(idx[b, r])) < k:
                idx[b, r] = rng.choice(N, size=k, replace=False)
    t = torch.from_numpy(idx).cuda()
    start = torch.full((B, N + 1), -1, dtype=torch.int32, device='cuda')
    src = torch.full((B, N * k), -1, dtype=torch.int32, device='cuda')
    L.call('pdae_knn_reverse', t, B, N, k, t.data_ptr(), start.data_ptr(), src.data_ptr())
    start, src = start.cpu().numpy(), src.cpu().numpy()
    for b in range(B):
        rows, cols = np.nonzero(np.ones((N, k), dtype=bool))
        order = np.lexsort((rows, idx[b].reshape(-1)))             # by target, then by source row
        want_src = rows[order]
        counts = np.bincount(idx[b].reshape(-1), minlength=N)
        assert np.array_equal(start[b], np.concatenate([[0], np.cumsum(counts)]))
        assert np.array_equal(src[b], want_src)


def _edge_reference(x, idx, w, bn, training=True):
    """get_graph_feature + Conv2d + BatchNorm2d + LeakyReLU + max, as the reference computes it (dense edge tensor)."""
    B, N, k = idx.shape
    C = x.shape[1]
    flat = (idx.long() + torch.arange(B, device=x.device).view(-1, 1, 1) * N).reshape(-1)
    nb = x.index_select(0, flat).view(B * N, k, C)
    feat = torch.cat([nb - x.unsqueeze(1), x.unsqueeze(1).expand(-1, k, -1)], dim=2)          # (R, k, 2C)
    e = feat.reshape(B * N * k, 2 * C) @ w.t()
    y = F.batch_norm(e, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps)
    return F.leaky_relu(y, 0.2).view(B * N, k, -1).max(dim=1)[0]


@pytest.mark.parametrize('B,N', [(4, 256), (2, 1024), (3, 100), (32, 64), (2, 716), (5, 333)])
def test_encoder_equals_the_dense_edge_formulation(B, N, monkeypatch):
    """The four EdgeConvs + conv5 pool through _Encoder's kernels vs the dense formulation on the same graphs: features,
    running estimates, and every gradient; gamma with positive, negative and zero entries (max / min / first edge)."""
    torch.manual_seed(B * 1000 + N)
    from point_dae_amd import point_cae_dgcnn as D
    from point_dae_amd.point_cae_dgcnn import dgcnn_encoder, feature_knn
    enc = dgcnn_encoder(channel=3).cuda().train()
    ref = dgcnn_encoder(channel=3).cuda().train()
    ref.load_state_dict(enc.state_dict())
    ref = ref.double()                       # the dense formulation in fp64: the yardstick for both fp32 summation orders
    for m in (enc, ref):
        with torch.no_grad():
            for bn in (m.bn1, m.bn2, m.bn3, m.bn4, m.bn5):
                bn.weight.copy_(torch.linspace(-1.0, 1.5, bn.weight.numel()))
                bn.weight[::7] = 0.0
                bn.bias.copy_(torch.linspace(-0.3, 0.3, bn.bias.numel()))
    x = torch.randn(B, 3, N, device='cuda')
    graphs = []
    monkeypatch.setattr(D, 'feature_knn', lambda *a, **kw: graphs.append(feature_knn(*a, **kw)) or graphs[-1])
    feat = enc(x)
    tgt = torch.randn_like(feat)
    (feat * tgt).sum().backward()

    # the dense formulation, on the graphs the product built (a near-tie may legitimately pick another neighbour)
    rows = x.double().transpose(1, 2).reshape(B * N, 3)
    feats = []
    for conv, idx in zip((ref.conv1, ref.conv2, ref.conv3, ref.conv4), graphs):
        rows = _edge_reference(rows, idx, conv[0].weight.flatten(1), conv[1])
        feats.append(rows)
    y = torch.cat(feats, 1) @ ref.conv5[0].weight.squeeze(-1).t()
    y = F.leaky_relu(F.batch_norm(y, ref.bn5.running_mean, ref.bn5.running_var, ref.bn5.weight, ref.bn5.bias, True,
                                  ref.bn5.momentum, ref.bn5.eps), 0.2)
    want = y.view(B, N, -1).max(dim=1)[0]
    (want * tgt.double()).sum().backward()
    assert _rel(feat, want) <= 2e-5, _rel(feat, want)
    for (name, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, name
        # (typically 1e-5; the bound leaves room for LeakyReLU's kink: a winner whose pre-activation is within fp32
        # rounding of 0 takes slope 1 in one computation and 0.2 in the other -- one such entry moves a BatchNorm bias
        # gradient by 2-4e-4 of its norm; seen with PDAE_GEMM=f32mfma, whose products sit further from the fp64 ones)
        assert _rel(p.grad, q.grad) <= 1e-3, (name, _rel(p.grad, q.grad))
    for (name, a), (_, b) in zip(enc.named_buffers(), ref.named_buffers()):
        if a.dtype.is_floating_point:
            assert torch.allclose(a, b.float(), rtol=1e-4, atol=1e-6), name
        else:
            assert a.item() == 1, name                              # num_batches_tracked (F.batch_norm leaves the reference's at 0)


def test_eval_mode_uses_the_running_estimates(monkeypatch):
    torch.manual_seed(5)
    from point_dae_amd import point_cae_dgcnn as D
    from point_dae_amd.point_cae_dgcnn import dgcnn_encoder, feature_knn
    B, N = 2, 200
    enc = dgcnn_encoder(channel=3).cuda()
    with torch.no_grad():
        for bn in (enc.bn1, enc.bn2, enc.bn3, enc.bn4, enc.bn5):
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.normal_(0, 1)
    enc.eval()
    x = torch.randn(B, 3, N, device='cuda')
    graphs = []
    monkeypatch.setattr(D, 'feature_knn', lambda *a, **kw: graphs.append(feature_knn(*a, **kw)) or graphs[-1])
    with torch.no_grad():
        feat = enc(x)
        rows = x.transpose(1, 2).reshape(B * N, 3)
        feats = []
        for conv, idx in zip((enc.conv1, enc.conv2, enc.conv3, enc.conv4), graphs):
            rows = _edge_reference(rows, idx, conv[0].weight.flatten(1), conv[1], training=False)
            feats.append(rows)
        y = torch.cat(feats, 1) @ enc.conv5[0].weight.squeeze(-1).t()
        y = F.leaky_relu(F.batch_norm(y, enc.bn5.running_mean, enc.bn5.running_var, enc.bn5.weight, enc.bn5.bias, False,
                                      0.1, enc.bn5.eps), 0.2)
        want = y.view(B, N, -1).max(dim=1)[0]
    assert _rel(feat, want) <= 2e-5
    assert enc.bn1.num_batches_tracked.item() == 0


def test_eval_mode_gradients_have_no_batch_statistic_terms(monkeypatch):
    """Fine-tuning with frozen (eval) BatchNorm: the backward of an eval-mode forward is d x = scale dy through the
    winners only -- the c1 / c2 corrections of the training-mode formula must not be applied (dgcnn_util.py:117-136 with
    the modules in .eval()).  Every parameter gradient against the dense fp64 formulation with training=False."""
    torch.manual_seed(9)
    from point_dae_amd import point_cae_dgcnn as D
    from point_dae_amd.point_cae_dgcnn import dgcnn_encoder, feature_knn
    B, N = 3, 160
    enc = dgcnn_encoder(channel=3).cuda()
    with torch.no_grad():
        for bn in (enc.bn1, enc.bn2, enc.bn3, enc.bn4, enc.bn5):
            bn.running_mean.normal_(0, 0.1)
            bn.running_var.uniform_(0.5, 1.5)
            bn.weight.copy_(torch.linspace(-1.0, 1.5, bn.weight.numel()))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, bn.bias.numel()))
    ref = dgcnn_encoder(channel=3).cuda()
    ref.load_state_dict(enc.state_dict())
    ref = ref.double().eval()
    enc.eval()
    x = torch.randn(B, 3, N, device='cuda')
    graphs = []
    monkeypatch.setattr(D, 'feature_knn', lambda *a, **kw: graphs.append(feature_knn(*a, **kw)) or graphs[-1])
    feat = enc(x)
    tgt = torch.randn_like(feat)
    (feat * tgt).sum().backward()
    rows = x.double().transpose(1, 2).reshape(B * N, 3)
    feats = []
    for conv, idx in zip((ref.conv1, ref.conv2, ref.conv3, ref.conv4), graphs):
        rows = _edge_reference(rows, idx, conv[0].weight.flatten(1), conv[1], training=False)
        feats.append(rows)
    y = torch.cat(feats, 1) @ ref.conv5[0].weight.squeeze(-1).t()
    y = F.leaky_relu(F.batch_norm(y, ref.bn5.running_mean, ref.bn5.running_var, ref.bn5.weight, ref.bn5.bias, False,
                                  0.1, ref.bn5.eps), 0.2)
    want = y.view(B, N, -1).max(dim=1)[0]
    (want * tgt.double()).sum().backward()
    assert _rel(feat, want) <= 2e-5
    for (name, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, name
        assert _rel(p.grad, q.grad) <= 1e-3, (name, _rel(p.grad, q.grad))


def test_entries_refuse_what_they_do_not_implement():
    L = _lib()
    t = torch.zeros(16, device='cuda')
    with pytest.raises(RuntimeError, match='k > n'):
        L.call('pdae_gram_topk', t, 1, 4, 5, t.data_ptr(), t.data_ptr(), t.data_ptr())
    with pytest.raises(RuntimeError, match='4096'):
        L.call('pdae_knn_reverse', t, 1, 5000, 2, t.data_ptr(), t.data_ptr(), t.data_ptr())
    with pytest.raises(RuntimeError, match='powers of two'):
        L.call('pdae_edge_gather_stats', t, 1, 8, 2, 48, *([t.data_ptr()] * 8))


@pytest.mark.parametrize('B,N,C', [(32, 1024, 64), (3, 1024, 128), (2, 300, 64), (1, 64, 128), (2, 2500, 8), (5, 20, 4), (2, 68, 4)])
def test_gram_matrices_on_the_upper_triangle_equal_the_full_product(B, N, C):
    """pdae_rows_gemm_batched with W = X (a Gram matrix) computes the 64 x 64 tiles on and above the diagonal and stores
    each one a second time transposed: bit for bit the matrix the general path (a separate copy of X as W) computes,
    symmetric, and X X^T to fp32 rounding."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(N + C)
    x = torch.randn(B * N, C, device='cuda', generator=g)
    sym = torch.full((B, N, N), float('nan'), device='cuda')
    L.call('pdae_rows_gemm_batched', x, B, N, N, C, x.data_ptr(), N * C, x.data_ptr(), N * C, sym.data_ptr(), N * N)
    w = x.clone()
    full = torch.full((B, N, N), float('nan'), device='cuda')
    L.call('pdae_rows_gemm_batched', x, B, N, N, C, x.data_ptr(), N * C, w.data_ptr(), N * C, full.data_ptr(), N * N)
    assert torch.isfinite(sym).all()
    assert torch.equal(sym, full)
    assert torch.equal(sym, sym.transpose(1, 2))
    xb = x.view(B, N, C).double()
    ref = xb @ xb.transpose(1, 2)
    assert (sym.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
