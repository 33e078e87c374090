"""GPU numerics of the hand-written fp32 MFMA GEMM kernels and their fused
producer / epilogue variants against plain PyTorch fp32 compositions of the
same ops (tolerance: fp32 GEMM reassociation, 2e-5 of the output scale)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from point_dae_amd import _lib
    return _lib


def _close(a, b, tol=2e-5):
    scale = b.abs().max().item() + 1e-12
    err = (a - b).abs().max().item() / scale
    assert err <= tol, err


SHAPES = [(4096, 384, 512), (131072, 384, 512), (2944, 1152, 384), (8192, 384, 1536), (1000, 96, 384),
          (37, 128, 132), (65536, 512, 256), (5248, 96, 384), (300, 1536, 384)]


@pytest.mark.parametrize('M,N,K', SHAPES)
@pytest.mark.parametrize('act', [0, 1, 2])
def test_linear_forward(M, N, K, act):
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(M + N + K)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b = torch.randn(N, device='cuda', generator=g)
    y = torch.empty(M, N, device='cuda')
    L.call('pdae_linear_forward', x, M, N, K, x.data_ptr(), w.data_ptr(), b.data_ptr(), act, y.data_ptr())
    ref = F.linear(x.double(), w.double(), b.double())
    ref = [ref, F.relu(ref), F.gelu(ref)][act].float()
    _close(y, ref)
    L.call('pdae_linear_forward', x, M, N, K, x.data_ptr(), w.data_ptr(), None, 0, y.data_ptr())
    _close(y, F.linear(x.double(), w.double()).float())


@pytest.mark.parametrize('M,N,K', SHAPES)
def test_linear_backward(M, N, K):
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    dy = torch.randn(M, N, device='cuda', generator=g)
    wt = w.t().contiguous()
    dx = torch.empty(M, K, device='cuda')
    L.call('pdae_linear_backward_data', x, M, N, K, dy.data_ptr(), wt.data_ptr(), dx.data_ptr())
    _close(dx, (dy.double() @ w.double()).float())
    dw = torch.full((N, K), 7.0, device='cuda')          # must be overwritten, not accumulated
    db = torch.full((N,), 7.0, device='cuda')
    L.call('pdae_linear_backward_weight', x, M, N, K, dy.data_ptr(), x.data_ptr(), dw.data_ptr(), db.data_ptr())
    _close(dw, (dy.double().t() @ x.double()).float(), 5e-5)     # fp32 atomics over M-splits
    _close(db, dy.double().sum(0).float(), 5e-5)


@pytest.mark.parametrize('G,N,K', [(8192, 256, 128), (512, 256, 128), (33, 128, 64), (4096, 384, 512)])
def test_embed_conv_store_groupmax(G, N, K):
    L = _lib()
    M = G * 32
    g = torch.Generator(device='cuda').manual_seed(1)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b = torch.randn(N, device='cuda', generator=g)
    y = torch.empty(M, N, device='cuda')
    gm = torch.empty(G, N, device='cuda')
    ga = torch.empty(G, N, device='cuda', dtype=torch.uint8)
    L.call('pdae_embed_conv_store_groupmax', x, M, N, K, x.data_ptr(), w.data_ptr(), b.data_ptr(),
           y.data_ptr(), gm.data_ptr(), ga.data_ptr())
    ref = F.linear(x, w, b)
    _close(y, ref)
    # max / argmax must be consistent with the Y the kernel itself produced (bit exact)
    mx, am = y.reshape(G, 32, N).max(dim=1)
    assert torch.equal(gm, mx)
    picked = torch.gather(y.reshape(G, 32, N), 1, ga.long().unsqueeze(1)).squeeze(1)
    assert torch.equal(picked, mx)
    first = (y.reshape(G, 32, N) == mx.unsqueeze(1)).float().argmax(dim=1)
    assert torch.equal(ga.long(), first)


@pytest.mark.parametrize('G,N,K', [(8192, 512, 256), (100, 512, 256), (1024, 128, 64)])
def test_embed_conv_groupbias_stats(G, N, K):
    L = _lib()
    M = G * 32
    g = torch.Generator(device='cuda').manual_seed(2)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    gb = torch.randn(G, N, device='cuda', generator=g)
    y = torch.empty(M, N, device='cuda')
    st = torch.full((8, 2, N), 3.0, device='cuda')
    L.call('pdae_embed_conv_groupbias_stats', x, M, N, K, x.data_ptr(), w.data_ptr(), gb.data_ptr(),
           y.data_ptr(), st.data_ptr())
    ref = (F.linear(x, w).reshape(G, 32, N) + gb.unsqueeze(1)).reshape(M, N)
    _close(y, ref)
    s = st.sum(0)
    _close(s[0], y.double().sum(0).float(), 1e-4)
    _close(s[1], (y.double() ** 2).sum(0).float(), 1e-5)


@pytest.mark.parametrize('G,N,K', [(8192, 384, 512), (77, 384, 512), (256, 128, 64)])
def test_embed_bnrelu_conv_groupmax(G, N, K):
    L = _lib()
    M = G * 32
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(M, K, device='cuda', generator=g)
    sc = torch.rand(K, device='cuda', generator=g) + 0.5
    sh = torch.randn(K, device='cuda', generator=g) * 0.3
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    b = torch.randn(N, device='cuda', generator=g)
    gm = torch.empty(G, N, device='cuda')
    ga = torch.empty(G, N, device='cuda', dtype=torch.uint8)
    L.call('pdae_embed_bnrelu_conv_groupmax', x, M, N, K, x.data_ptr(), sc.data_ptr(), sh.data_ptr(),
           w.data_ptr(), b.data_ptr(), gm.data_ptr(), ga.data_ptr(), None)
    a = F.relu(x * sc + sh)
    ref = F.linear(a.double(), w.double(), b.double()).float().reshape(G, 32, N)
    mx, am = ref.max(dim=1)
    _close(gm, mx)
    # argmax may differ only where two rows tie within rounding: check the value at the reported row
    picked = torch.gather(ref, 1, ga.long().unsqueeze(1)).squeeze(1)
    _close(picked, mx, 1e-5)
    assert (ga.long() == am).float().mean() > 0.999
    # weight gradient with the activation recomputed in the producer
    dy = torch.randn(M, N, device='cuda', generator=g)
    dw = torch.empty(N, K, device='cuda')
    db = torch.empty(N, device='cuda')
    L.call('pdae_bnrelu_linear_backward_weight', x, M, N, K, dy.data_ptr(), x.data_ptr(), sc.data_ptr(),
           sh.data_ptr(), dw.data_ptr(), db.data_ptr(), None)
    _close(dw, (dy.double().t() @ a.double()).float(), 5e-5)
    _close(db, dy.double().sum(0).float(), 5e-5)       # bias gradient from the same kernel
    # group list: only some groups go through the layer, outputs compact in list order
    sel = torch.randperm(G, device='cuda', generator=g)[:max(1, G // 3)].sort()[0].to(torch.int32)
    Gs = sel.numel()
    gm2 = torch.empty(Gs, N, device='cuda')
    ga2 = torch.empty(Gs, N, device='cuda', dtype=torch.uint8)
    L.call('pdae_embed_bnrelu_conv_groupmax', x, Gs * 32, N, K, x.data_ptr(), sc.data_ptr(), sh.data_ptr(),
           w.data_ptr(), b.data_ptr(), gm2.data_ptr(), ga2.data_ptr(), sel.data_ptr())
    assert torch.equal(gm2, gm[sel.long()]) and torch.equal(ga2, ga[sel.long()])
    dyc = torch.randn(Gs * 32, N, device='cuda', generator=g)
    L.call('pdae_bnrelu_linear_backward_weight', x, Gs * 32, N, K, dyc.data_ptr(), x.data_ptr(), sc.data_ptr(),
           sh.data_ptr(), dw.data_ptr(), None, sel.data_ptr())
    a_sel = a.reshape(G, 32, K)[sel.long()].reshape(Gs * 32, K)
    _close(dw, (dyc.double().t() @ a_sel.double()).float(), 5e-5)


def _embed_reference(points, first_conv, second_conv):
    """Plain PyTorch composition of Encoder.forward (PointCAE_transformer.py:37-51)."""
    bs_g, n, _ = points.shape
    f = first_conv(points.transpose(2, 1))
    fg = f.max(dim=2, keepdim=True)[0]
    f = second_conv(torch.cat([fg.expand(-1, -1, n), f], dim=1))
    return f.max(dim=2)[0]


@pytest.mark.parametrize('BG', [256, 8192])
def test_fused_patch_embed_matches_pytorch(BG):
    """Fused embedder (forward + backward) against the plain PyTorch composition.
    Ground truth is the float64 composition; the fused fp32 path must be as
    accurate as PyTorch's own fp32 path (this backward is ill-conditioned: both
    fp32 paths sit ~1e-3 from float64)."""
    import copy
    import torch.nn as nn
    from point_dae_amd.patch_embed import patch_embed
    torch.manual_seed(0)
    first = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True), nn.Conv1d(128, 256, 1)).cuda()
    second = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, 384, 1)).cuda()
    for m in list(first) + list(second):
        if isinstance(m, nn.BatchNorm1d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    first_r, second_r = copy.deepcopy(first), copy.deepcopy(second)
    first_d, second_d = copy.deepcopy(first).double(), copy.deepcopy(second).double()
    pts = torch.randn(BG, 32, 3, device='cuda') * 0.2
    go = torch.randn(BG, 384, device='cuda')
    out = patch_embed(pts, first, second, True)
    out.backward(go)
    ref = _embed_reference(pts, first_r, second_r)
    ref.backward(go)
    refd = _embed_reference(pts.double(), first_d, second_d)
    refd.backward(go.double())
    _close(out, refd.float(), 2e-5)
    mine = list(first.parameters()) + list(second.parameters())
    t32 = list(first_r.parameters()) + list(second_r.parameters())
    t64 = list(first_d.parameters()) + list(second_d.parameters())
    gmax = max(c.grad.abs().max().item() for c in t64)
    for a, b, c in zip(mine, t32, t64):
        scale = c.grad.abs().max().item()
        if scale < 1e-6:           # conv bias feeding a training-mode BatchNorm: exactly zero gradient,
            assert a.grad.abs().max().item() < 1e-4 * gmax      # fp32 rounding noise on both paths
            continue
        e_mine = (a.grad.double() - c.grad).abs().max().item() / scale
        e_t32 = (b.grad.double() - c.grad).abs().max().item() / scale
        l2_mine = ((a.grad.double() - c.grad).norm() / c.grad.norm()).item()
        # An arg-max that flips between an fp32 path and fp64 (near-tied rows of a patch) reroutes
        # one element's gradient: a percent-level spike in the max norm of either fp32 path, nothing
        # in the L2 norm.  Max norm as tight as PyTorch's own fp32 path, or L2-tight with a bounded spike.
        ok_max = e_mine <= 1e-2 and e_mine <= 5 * e_t32 + 3e-3
        ok_l2 = l2_mine <= 2e-3 and e_mine <= 5e-2
        ok_like_torch = e_mine <= 1.5 * e_t32 and e_mine <= 5e-2      # PyTorch's fp32 path has the same flip
        assert ok_max or ok_l2 or ok_like_torch, (tuple(a.shape), e_mine, e_t32, l2_mine)
    for a, b in zip(list(first.buffers()) + list(second.buffers()), list(first_d.buffers()) + list(second_d.buffers())):
        assert torch.allclose(a.double(), b.double(), rtol=1e-4, atol=1e-5)
    # visible-groups path == all groups + selection, forward and backward
    import copy as _copy
    fa, sa = _copy.deepcopy(first_r), _copy.deepcopy(second_r)
    fb, sb = _copy.deepcopy(first_r), _copy.deepcopy(second_r)
    for mods in ((fa, sa), (fb, sb)):
        for mm in mods:
            mm.zero_grad()
    sel = torch.arange(0, BG, 3, device='cuda', dtype=torch.int32)
    gsel = torch.randn(sel.numel(), 384, device='cuda')
    out_a = patch_embed(pts, fa, sa, True)[sel.long()]
    out_a.backward(gsel)
    out_b = patch_embed(pts, fb, sb, True, sel)
    out_b.backward(gsel)
    # the same with the complementary list given: the masked groups' share of the backward by algebra
    fc, sc_ = _copy.deepcopy(first_r), _copy.deepcopy(second_r)
    for mm in (fc, sc_):
        mm.zero_grad()
    rest = torch.tensor([i for i in range(BG) if i % 3 != 0], device='cuda', dtype=torch.int32)
    out_c = patch_embed(pts, fc, sc_, True, sel, rest)
    out_c.backward(gsel)
    _close(out_c, out_b.detach(), 1e-5)
    for pb_, pc_ in zip(list(fb.parameters()) + list(sb.parameters()), list(fc.parameters()) + list(sc_.parameters())):
        if pb_.grad.abs().max().item() < 1e-4 * gmax:
            continue
        rel = (pb_.grad - pc_.grad).norm().item() / (pb_.grad.norm().item() + 1e-12)
        assert rel <= 5e-3, ('algebra', tuple(pb_.shape), rel)
    _close(out_b, out_a.detach(), 1e-5)      # BatchNorm sums use atomics: last-bit run-to-run noise
    # The two runs draw BatchNorm sums through atomics (last-bit differences), which can flip
    # a near-tied arg-max row and reroute that element's gradient: compare in the L2 sense.
    for pa, pb in zip(list(fa.parameters()) + list(sa.parameters()), list(fb.parameters()) + list(sb.parameters())):
        if pa.grad.abs().max().item() < 1e-4 * gmax:
            continue
        rel = (pa.grad - pb.grad).norm().item() / (pa.grad.norm().item() + 1e-12)
        assert rel <= 5e-3, (tuple(pa.shape), rel)
    # eval mode: running statistics
    first.eval(), second.eval(), first_d.eval(), second_d.eval()
    with torch.no_grad():
        _close(patch_embed(pts, first, second, False), _embed_reference(pts.double(), first_d, second_d).float(), 2e-5)


@pytest.mark.parametrize('R,C', [(262144, 128), (4096, 64), (1000, 128), (37, 256)])
def test_conv1_stats_and_bn_finalize(R, C):
    """embed_conv1_stats + bn_finalize against nn.Conv1d(3, C, 1) -> nn.BatchNorm1d(C) in training mode
    (values, batch statistics, running estimates, counter, the scale/shift the next kernel consumes)."""
    import torch.nn as nn
    L = _lib()
    torch.manual_seed(R + C)
    x = torch.randn(R, 3, device='cuda') * 0.3
    conv = nn.Conv1d(3, C, 1).cuda()
    bn = nn.BatchNorm1d(C).cuda()
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.normal_(0, 0.2)
    bn.running_mean.normal_(0, 0.1)
    bn.running_var.uniform_(0.5, 2.0)
    ref_bn = nn.BatchNorm1d(C).cuda().double()
    ref_bn.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    w = conv.weight.detach().squeeze(-1).contiguous()
    y = torch.empty(R, C, device='cuda')
    st = torch.zeros(2, C, device='cuda', dtype=torch.float64)
    L.call('pdae_embed_conv1_stats', x, R, C, x.data_ptr(), w.data_ptr(), conv.bias.data_ptr(), y.data_ptr(),
           st.data_ptr())
    y64 = x.double() @ w.double().t() + conv.bias.detach().double()
    _close(y, y64.float(), 1e-6)
    _close(st[0], y64.sum(0), 1e-6)
    _close(st[1], (y64 * y64).sum(0), 1e-6)
    scale, shift, mean, invstd = (torch.empty(C, device='cuda') for _ in range(4))
    L.call('pdae_bn_finalize', x, C, R, st.data_ptr(), None, 0, bn.weight.data_ptr(), bn.bias.data_ptr(),
           float(bn.eps), 0.1, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
           bn.num_batches_tracked.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
           invstd.data_ptr())
    ref_bn.train()
    out64 = ref_bn(y64)
    _close(y * scale + shift, out64.float(), 2e-5)
    _close(mean, y64.mean(0).float(), 1e-5)
    _close(invstd, (y64.var(0, unbiased=False) + bn.eps).rsqrt().float(), 1e-5)
    assert torch.allclose(bn.running_mean.double(), ref_bn.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), ref_bn.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn.num_batches_tracked) == int(ref_bn.num_batches_tracked) == 1
    # the fp32 partial-set form (what embed_conv_groupbias_stats leaves)
    parts = torch.zeros(8, 2, C, device='cuda')
    parts[3, 0], parts[3, 1] = st[0].float(), st[1].float()
    parts[5, 0] += 1.0
    parts[6, 0] -= 1.0
    L.call('pdae_bn_finalize', x, C, R, None, parts.data_ptr(), 8, bn.weight.data_ptr(), bn.bias.data_ptr(),
           float(bn.eps), 0.1, None, None, None, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
           invstd.data_ptr())
    _close(mean, y64.mean(0).float(), 1e-5)


def test_listed_group_entries():
    """The group-list entries of the algebraic embedder backward against torch on gathered rows:
    linear_backward_weight_listed (gathers on either operand; Gram matrix), group_gemm_scatter (gather, per-group
    bias, scatter; untouched rows stay), group_sum_listed, masked_group_sums, and their argument checks."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(5)
    G, C, K = 40, 256, 128
    f = torch.randn(G * 32, C, device='cuda', generator=g)
    lst = torch.tensor([3, 0, 17, 39, 8, 21], device='cuda', dtype=torch.int32)
    rows = (lst.long().unsqueeze(1) * 32 + torch.arange(32, device='cuda')).reshape(-1)
    M = rows.numel()
    # Gram matrix of the listed rows
    gram = torch.full((C, C), float('nan'), device='cuda')
    L.call('pdae_linear_backward_weight_listed', f, M, C, C, f.data_ptr(), lst.data_ptr(), f.data_ptr(), lst.data_ptr(),
           gram.data_ptr(), None)
    _close(gram, f[rows].double().t() @ f[rows].double(), 2e-5)
    # compact dY against gathered X, with the bias-gradient column sums
    dy = torch.randn(M, K, device='cuda', generator=g)
    dw, db = torch.empty(K, C, device='cuda'), torch.empty(K, device='cuda')
    L.call('pdae_linear_backward_weight_listed', f, M, K, C, dy.data_ptr(), None, f.data_ptr(), lst.data_ptr(),
           dw.data_ptr(), db.data_ptr())
    _close(dw, dy.double().t() @ f[rows].double(), 2e-5)
    _close(db, dy.double().sum(0), 2e-5)
    # gather -> product -> per-group bias -> scatter
    w = torch.randn(K, C, device='cuda', generator=g) * 0.1
    gbias = torch.randn(lst.numel(), K, device='cuda', generator=g)
    y = torch.full((G * 32, K), 7.0, device='cuda')
    L.call('pdae_group_gemm_scatter', f, M, K, C, f.data_ptr(), lst.data_ptr(), w.data_ptr(), gbias.data_ptr(),
           y.data_ptr(), K, lst.data_ptr())
    want = torch.full((G * 32, K), 7.0, device='cuda', dtype=torch.float64)
    want[rows] = f[rows].double() @ w.double().t() + gbias.double().repeat_interleave(32, 0)
    _close(y, want, 2e-5)
    assert (y[torch.ones(G * 32, dtype=torch.bool, device='cuda').index_fill_(0, rows, False)] == 7.0).all()
    # compact input, no bias
    y2 = torch.zeros(G * 32, K, device='cuda')
    xc = f[rows].contiguous()
    L.call('pdae_group_gemm_scatter', f, M, K, C, xc.data_ptr(), None, w.data_ptr(), None, y2.data_ptr(), K, lst.data_ptr())
    _close(y2[rows], xc.double() @ w.double().t(), 2e-5)
    # group sums
    gs = torch.empty(lst.numel(), C, device='cuda')
    L.call('pdae_group_sum_listed', f, lst.numel(), C, f.data_ptr(), lst.data_ptr(), gs.data_ptr())
    _close(gs, f.view(G, 32, C)[lst.long()].double().sum(1), 1e-5)
    hs, xe, v = (torch.randn(lst.numel(), C, device='cuda', generator=g), torch.randn(lst.numel(), C, device='cuda', generator=g),
                 torch.randn(C, device='cuda', generator=g))
    dgb = torch.zeros(G, C, device='cuda')
    L.call('pdae_masked_group_sums', f, lst.numel(), C, hs.data_ptr(), xe.data_ptr(), v.data_ptr(), lst.data_ptr(), dgb.data_ptr())
    assert torch.allclose(dgb[lst.long()], v * hs + 32 * xe, rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError, match='multiple of 32'):
        L.call('pdae_linear_backward_weight_listed', f, 40, C, C, f.data_ptr(), None, f.data_ptr(), None, gram.data_ptr(), None)
    with pytest.raises(RuntimeError, match='multiple of 32'):
        L.call('pdae_group_gemm_scatter', f, 40, K, C, f.data_ptr(), None, w.data_ptr(), None, y.data_ptr(), K, lst.data_ptr())
    with pytest.raises(RuntimeError, match='null pointer'):
        L.call('pdae_group_gemm_scatter', f, M, K, C, f.data_ptr(), None, w.data_ptr(), None, y.data_ptr(), K, None)
    with pytest.raises(RuntimeError, match='n_listed <= G'):
        L.call('pdae_bnrelu_backward_listed', f, 2, C, f.data_ptr(), f.data_ptr(), v.data_ptr(), v.data_ptr(), v.data_ptr(),
               v.data_ptr(), v.data_ptr(), gram.data_ptr(), None, 0, None, 5, lst.data_ptr())


@pytest.mark.parametrize('R,C', [(262144, 128), (5000, 128), (77, 64)])
def test_conv1_backward_weight(R, C):
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(R)
    d = torch.randn(R, C, device='cuda', generator=g)
    x = torch.randn(R, 3, device='cuda', generator=g)
    parts = L.lib().pdae_embed_conv1_backward_weight_parts(R)
    part = torch.full((parts, 3, C), float('nan'), device='cuda')
    L.call('pdae_embed_conv1_backward_weight', d, R, C, d.data_ptr(), x.data_ptr(), part.data_ptr())
    _close(part.sum(0), x.double().t() @ d.double(), 2e-5)
    part2 = torch.empty_like(part)
    L.call('pdae_embed_conv1_backward_weight', d, R, C, d.data_ptr(), x.data_ptr(), part2.data_ptr())
    assert torch.equal(part, part2)


@pytest.mark.parametrize('M,N,K,listed', [(69632, 512, 384, True), (8192, 128, 256, False), (4000, 260, 256, False),
                                          (4096, 64, 128, True), (352, 512, 384, True), (140000, 64, 64, False)])
@pytest.mark.parametrize('arith', [1, 0])
def test_data_gradient_with_batchnorm_backward_sums(M, N, K, listed, arith):
    """pdae_rows_gemm_bnrelu_stats: T = relu'(bn(X)) ? dY . W : 0 and S = (sum T, sum T xhat) out of ONE launch on the
    exact-split arithmetic (gemm3_kernel's EPI_BNRELU_STATS epilogue + the ordered fp64 finish), the two-launch form on the
    fp32-input arithmetic -- against the fp64 composition; X rows through a group list; ragged M and N; run to run
    bit-identical (no atomics on the fused path)."""
    from point_dae_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(M + N)
    prev = _lib.gemm_arith()
    _lib.set_gemm_arith(arith)
    try:
        dy = torch.randn(M, K, device='cuda', generator=g)
        w = torch.randn(K, N, device='cuda', generator=g) / K ** 0.5
        groups = None
        if listed:
            G = M // 32
            perm = torch.randperm(2 * G, device='cuda', generator=g)[:G].sort().values.to(torch.int32)
            X = torch.randn(2 * G * 32, N, device='cuda', generator=g)
            groups = perm
            xr = X.view(2 * G, 32, N)[perm.long()].reshape(M, N)
        else:
            X = torch.randn(M, N, device='cuda', generator=g)
            xr = X
        gamma = torch.rand(N, device='cuda', generator=g) + 0.5
        beta = torch.randn(N, device='cuda', generator=g) * 0.3
        mean, var = X.mean(0), X.var(0, unbiased=False)
        invstd = (var + 1e-5).rsqrt()
        scale = (gamma * invstd).contiguous()
        shift = (beta - mean * scale).contiguous()

        def run():
            t = torch.full((M, N), float('nan'), device='cuda')
            S = torch.full((2, N), float('nan'), device='cuda')
            ws = torch.empty(max(_lib.lib().pdae_rows_gemm_bnrelu_stats_workspace(M, N), 1), device='cuda')
            _lib.call('pdae_rows_gemm_bnrelu_stats', dy, M, N, K, _lib.ptr(dy), _lib.ptr(w), _lib.ptr(X), _lib.ptr(groups),
                      _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(t), _lib.ptr(S), _lib.ptr(ws))
            return t, S
        t, S = run()
        prod = dy.double() @ w.double()
        on = (xr * scale + shift) > 0
        t_ref = torch.where(on, prod, torch.zeros_like(prod))
        tm = torch.where(on, t.double(), torch.zeros_like(prod))             # (the two-launch form leaves T unmasked)
        assert (tm - t_ref).abs().max().item() <= 2e-5 * prod.abs().max().item()
        if arith == 1:
            assert torch.equal(t.double(), tm)                               # masked elements are exact zeros already
        xhat = (xr.double() - mean.double()) * invstd.double()
        S_ref = torch.stack([tm.sum(0), (tm * xhat).sum(0)])
        scale_s = S_ref.abs().max().item() + 1e-12
        assert (S.double() - S_ref).abs().max().item() <= 2e-5 * scale_s, (S.double() - S_ref).abs().max().item() / scale_s
        t2, S2 = run()
        assert torch.equal(t, t2)
        if arith == 1:
            assert torch.equal(S, S2)
    finally:
        _lib.set_gemm_arith(prev)
