"""Set-abstraction shared MLP + max-pool (point_dae_amd/sa_mlp.py: conv_stats, bnrelu_group_max,
group_max_scatter_n + the embedder's BatchNorm-backward kernels) against the reference's layer stack
(pointnet2_modules SharedMLP: Conv2d 1x1 no bias -> BatchNorm2d -> ReLU, then max over nsample) run by
PyTorch in fp64: outputs, every parameter gradient, the input gradient, running statistics."""
import copy

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    scale = b.abs().max().item() + 1e-12
    err = (a.double() - b.double()).abs().max().item() / scale
    assert err <= tol, err


def _level(spec, seed):
    from point_dae_amd.point_cae_pointnetv2 import SharedMLP
    torch.manual_seed(seed)
    mlp = SharedMLP(spec).cuda().train()
    with torch.no_grad():
        for layer in mlp:
            layer.bn.bn.weight.uniform_(0.5, 1.5)
            layer.bn.bn.bias.uniform_(-0.3, 0.3)
    return mlp


def _reference(x64, mlp64, groups, ns):
    """(R, K) rows -> the reference's (B=1, C, groups, ns) tensor through Conv2d / BatchNorm2d / ReLU / max."""
    t = x64.reshape(1, groups, ns, -1).permute(0, 3, 1, 2)
    for layer in mlp64:
        t = layer(t)
    return t.max(dim=3)[0][0].t()                               # (groups, C)


@pytest.mark.parametrize('groups,ns,spec', [(512, 32, [4, 64, 64, 128]), (96, 64, [132, 128, 128, 256]),
                                            (3, 128, [260, 256, 512, 1024]), (40, 16, [8, 32, 64])])
def test_shared_mlp_max_matches_torch_fp64(groups, ns, spec):
    from point_dae_amd import sa_mlp
    mlp = _level(spec, groups)
    mlp64 = copy.deepcopy(mlp).double()
    g = torch.Generator(device='cuda').manual_seed(ns)
    x = torch.randn(groups * ns, spec[0], device='cuda', generator=g).requires_grad_()
    x64 = x.detach().double().requires_grad_()
    go = torch.randn(groups, spec[-1], device='cuda', generator=g)
    out = sa_mlp.shared_mlp_max(x, list(mlp), ns)
    out.backward(go)
    ref = _reference(x64, mlp64, groups, ns)
    ref.backward(go.double())
    _close(out, ref, 2e-5)
    _close(x.grad, x64.grad, 2e-4)
    for (n, a), (_, b) in zip(mlp.named_parameters(), mlp64.named_parameters()):
        _close(a.grad, b.grad, 2e-4)
    for (n, a), (_, b) in zip(mlp.named_buffers(), mlp64.named_buffers()):
        if a.dtype.is_floating_point:
            assert torch.allclose(a.double(), b, rtol=1e-4, atol=1e-6), n
        else:
            assert int(a) == int(b), n


def test_group_max_kernels():
    from point_dae_amd import _lib
    G, ns, C = 37, 64, 128
    g = torch.Generator(device='cuda').manual_seed(1)
    y = torch.randn(G * ns, C, device='cuda', generator=g)
    sc = torch.randn(C, device='cuda', generator=g)              # both signs
    sh = torch.randn(C, device='cuda', generator=g) * 0.2
    out = torch.empty(G, C, device='cuda')
    arg = torch.empty(G, C, device='cuda', dtype=torch.uint8)
    _lib.call('pdae_bnrelu_group_max', y, G, ns, C, y.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(),
              arg.data_ptr())
    a = torch.relu(y * sc + sh).view(G, ns, C)
    want, widx = a.max(dim=1)
    assert torch.equal(out, want)
    # ties (rows clamped to 0): the first index wins
    first = (a == want.unsqueeze(1)).float().argmax(dim=1)
    assert torch.equal(arg.long(), first)
    go = torch.randn(G, C, device='cuda', generator=g)
    dense = torch.full((G * ns, C), float('nan'), device='cuda')
    _lib.call('pdae_group_max_scatter_n', y, G, ns, C, go.data_ptr(), arg.data_ptr(), dense.data_ptr())
    want_d = torch.zeros(G, ns, C, device='cuda').scatter_(1, arg.long().unsqueeze(1), go.unsqueeze(1))
    assert torch.equal(dense.view(G, ns, C), want_d)


@pytest.mark.parametrize('G,ns,C', [(300, 32, 128), (70, 64, 256), (5, 128, 1024), (1000, 16, 64)])
def test_pool_bn_backward_equals_dense_form(G, ns, C):
    """pool_bn_backward (through the max-pool) == group_max_scatter_n + bnrelu_backward (dense), and twice the
    same bits."""
    from point_dae_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(G)
    R = G * ns
    y = torch.randn(R, C, device='cuda', generator=g)
    gamma = torch.rand(C, device='cuda', generator=g) + 0.5
    gamma[::7] *= -1                                               # negative scales too
    beta = torch.randn(C, device='cuda', generator=g) * 0.3
    mean = y.mean(0)
    invstd = torch.rsqrt(y.var(0, unbiased=False) + 1e-5)
    sc = (gamma * invstd).contiguous()
    sh = (beta - mean * sc).contiguous()
    out = torch.empty(G, C, device='cuda')
    arg = torch.empty(G, C, device='cuda', dtype=torch.uint8)
    _lib.call('pdae_bnrelu_group_max', y, G, ns, C, y.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(),
              arg.data_ptr())
    go = torch.randn(G, C, device='cuda', generator=g)
    # dense form
    d = torch.empty(R, C, device='cuda')
    _lib.call('pdae_group_max_scatter_n', y, G, ns, C, go.data_ptr(), arg.data_ptr(), d.data_ptr())
    S_d = torch.empty(2, C, device='cuda')
    _lib.call('pdae_bnrelu_backward', y, R // 32, C, d.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(),
              mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), S_d.data_ptr(), None, R // 32, None, None, None)
    res = []
    for _ in range(2):
        S = torch.full((2, C), float('nan'), device='cuda')
        dy = torch.full((R, C), float('nan'), device='cuda')
        ws = torch.empty(max(_lib.lib().pdae_pool_bn_backward_workspace(G, C), 1), device='cuda')
        _lib.call('pdae_pool_bn_backward', y, G, ns, C, go.data_ptr(), arg.data_ptr(), out.data_ptr(), y.data_ptr(),
                  mean.data_ptr(), invstd.data_ptr(), gamma.data_ptr(), S.data_ptr(), ws.data_ptr(), dy.data_ptr())
        res.append((S, dy))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    _close(res[0][0], S_d, 2e-5)
    _close(res[0][1], d, 2e-5)


def test_entries_refuse_bad_arguments():
    """error behaviour of the round-2 entries: status + message, never a launch on bad sizes / null pointers"""
    from point_dae_amd import _lib
    x = torch.zeros(64, 8, device='cuda')
    w = torch.zeros(16, 8, device='cuda')
    y = torch.zeros(64, 16, device='cuda')
    st = torch.zeros(8, 2, 16, device='cuda')
    sc = torch.ones(8, device='cuda')
    with pytest.raises(RuntimeError, match='multiple of 4'):          # K % 4 != 0
        _lib.call('pdae_conv_stats', x, 64, 16, 6, x.data_ptr(), None, None, w.data_ptr(), y.data_ptr(), st.data_ptr())
    with pytest.raises(RuntimeError, match='scale and shift'):
        _lib.call('pdae_conv_stats', x, 64, 16, 8, x.data_ptr(), sc.data_ptr(), None, w.data_ptr(), y.data_ptr(), st.data_ptr())
    with pytest.raises(RuntimeError, match='null pointer'):
        _lib.call('pdae_conv_stats', x, 64, 16, 8, x.data_ptr(), None, None, w.data_ptr(), y.data_ptr(), None)
    _lib.call('pdae_conv_stats', x, 0, 16, 8, None, None, None, None, None, st.data_ptr())        # M = 0: zeroed statistics
    assert float(st.abs().sum()) == 0.0
    out = torch.zeros(2, 16, device='cuda')
    arg = torch.zeros(2, 16, device='cuda', dtype=torch.uint8)
    with pytest.raises(RuntimeError, match='nsample'):
        _lib.call('pdae_bnrelu_group_max', y, 2, 300, 16, y.data_ptr(), sc.data_ptr(), sc.data_ptr(), out.data_ptr(), arg.data_ptr())
    with pytest.raises(RuntimeError, match='multiple of 4'):
        _lib.call('pdae_bnrelu_group_max', y, 2, 32, 6, y.data_ptr(), sc.data_ptr(), sc.data_ptr(), out.data_ptr(), arg.data_ptr())
    with pytest.raises(RuntimeError, match='C/4 must divide 256'):
        _lib.call('pdae_pool_bn_backward', y, 2, 32, 24, out.data_ptr(), arg.data_ptr(), out.data_ptr(), y.data_ptr(),
                  sc.data_ptr(), sc.data_ptr(), sc.data_ptr(), st.data_ptr(), st.data_ptr(), y.data_ptr())
    with pytest.raises(RuntimeError, match='at most 1024'):
        _lib.call('pdae_fold_input_grad', y, 1, 2, 4, 2048, y.data_ptr(), y.data_ptr(), y.data_ptr())
    with pytest.raises(RuntimeError, match='positive multiple of 4'):
        _lib.call('pdae_fold_input', y, 1, 2, 4, 6, y.data_ptr(), y.data_ptr(), y.data_ptr(), y.data_ptr())
    with pytest.raises(RuntimeError, match='needs Z'):                # masking epilogue without the kept output
        _lib.call('pdae_rows_gemm', x, 64, 16, 8, x.data_ptr(), torch.zeros(8, 16, device='cuda').data_ptr(), 1, None, 4,
                  None, y.data_ptr(), -1, 1, 0)
    with pytest.raises(RuntimeError, match='1 MiB'):
        _lib._check(_lib.lib(), 'pdae_set_deterministic', _lib.lib().pdae_set_deterministic(x.data_ptr(), 1024))
    assert not _lib.deterministic()


@pytest.mark.parametrize('B,N,npoint,ns,C', [(3, 512, 128, 64, 128), (2, 1024, 512, 32, 0), (2, 100, 7, 5, 8), (1, 64, 3, 4, 36)])
def test_sa_group_rows_matches_the_index_select_form(B, N, npoint, ns, C):
    """pdae_sa_group_rows / _grad (QueryAndGroup in row layout, pointnet2_utils.py:345-361) against the PyTorch form it
    replaces: index_select of coordinates and features, minus the centre, a zero column, cat -- rows bit-equal, the
    feature gradient to fp32 summation order (repeated indices: ball query pre-fills a ball with its first hit)."""
    from point_dae_amd.point_cae_pointnetv2 import _GroupRows
    g = torch.Generator(device='cuda').manual_seed(B * N + C)
    xyz = torch.rand(B, N, 3, device='cuda', generator=g)
    new_xyz = torch.rand(B, npoint, 3, device='cuda', generator=g)
    idx = torch.randint(0, N, (B, npoint, ns), device='cuda', generator=g, dtype=torch.int32)
    idx[:, :, ns // 2:] = idx[:, :, :1]                                  # duplicates inside a ball
    idx[0, 0] = 0                                                        # an empty ball: all zeros
    feats = torch.randn(B * N, C, device='cuda', generator=g, requires_grad=True) if C else None
    out = _GroupRows.apply(xyz, new_xyz, idx, feats)
    flat = (idx.long() + torch.arange(B, device='cuda').view(B, 1, 1) * N).reshape(-1)
    ref_f = feats.detach().clone().requires_grad_(True) if C else None
    gx = (xyz.reshape(B * N, 3).index_select(0, flat).reshape(B, npoint, ns, 3) - new_xyz.unsqueeze(2)).reshape(-1, 3)
    ref = torch.cat([gx, gx.new_zeros(gx.shape[0], 1)] + ([ref_f.index_select(0, flat)] if C else []), dim=1)
    assert out.shape == ref.shape and torch.equal(out, ref)
    if C:
        w = torch.randn(out.shape, device='cuda', generator=g)
        (out * w).sum().backward()
        (ref * w).sum().backward()
        scale = ref_f.grad.abs().max().item()
        assert (feats.grad - ref_f.grad).abs().max().item() <= 1e-5 * scale


@pytest.mark.parametrize('M,N', [(65536, 64), (2048 * 3 + 96, 128), (32, 16), (4096, 1024)])
@pytest.mark.parametrize('det', [False, True])
def test_first_layer_on_coordinates_alone(M, N, det):
    """conv_stats with K = 4 (xyz - centre | 0: the first layer of a level without features) runs on its own pass
    (csrc/gemm.hip conv_k4_stats_kernel): values against fp64, statistics against the sums of the values it stored."""
    from point_dae_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(M + N)
    x = torch.randn(M, 4, device='cuda', generator=g)
    x[:, 3] = 0
    w = torch.randn(N, 4, device='cuda', generator=g)
    y = torch.full((M, N), float('nan'), device='cuda')
    st = torch.full((8, 2, N), float('nan'), device='cuda')
    prev = _lib.deterministic()
    _lib.set_deterministic(det)
    try:
        _lib.call('pdae_conv_stats', x, M, N, 4, x.data_ptr(), None, None, w.data_ptr(), y.data_ptr(), st.data_ptr())
        if det:
            y2, st2 = torch.empty_like(y), torch.empty_like(st)
            _lib.call('pdae_conv_stats', x, M, N, 4, x.data_ptr(), None, None, w.data_ptr(), y2.data_ptr(), st2.data_ptr())
            assert torch.equal(y, y2) and torch.equal(st, st2)
    finally:
        _lib.set_deterministic(prev)
    want = x.double() @ w.double().t()
    assert float((y.double() - want).abs().max()) <= 1e-6 * max(float(want.abs().max()), 1.0)
    s = st.double().sum(0)
    assert torch.allclose(s[0], y.double().sum(0), rtol=1e-5, atol=1e-5 * float(y.double().abs().sum(0).max()))
    assert torch.allclose(s[1], (y.double() ** 2).sum(0), rtol=1e-5, atol=0)
