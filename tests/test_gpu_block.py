"""GPU numerics of the Transformer-block kernels (attention core, LayerNorm with
fused adds, GELU, bias+DropPath+residual) against plain PyTorch fp32 / fp64."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, tol):
    scale = b.abs().max().item() + 1e-12
    err = (a.double() - b.double()).abs().max().item() / scale
    assert err <= tol, err


@pytest.mark.parametrize('B,T,H', [(128, 23, 6), (64, 64, 6), (8, 13, 6), (5, 32, 2), (3, 100, 6), (2, 128, 6), (7, 1, 3),
                                   (16, 47, 6), (4, 33, 2), (128, 64, 6), (9, 63, 3)])
def test_attention_forward_backward(B, T, H):
    from point_dae_amd import nn_ops
    D, scale = 64, 64 ** -0.5
    g = torch.Generator(device='cuda').manual_seed(B * 1000 + T)
    qkv = torch.randn(B * T, 3 * H * D, device='cuda', generator=g, requires_grad=True)
    go = torch.randn(B * T, H * D, device='cuda', generator=g)
    o = nn_ops.attention_core(qkv, B, T, H, scale)
    o.backward(go)
    q64 = qkv.detach().double().requires_grad_(True)
    t = q64.reshape(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
    a = ((t[0] @ t[1].transpose(-2, -1)) * scale).softmax(dim=-1)
    ref = (a @ t[2]).transpose(1, 2).reshape(B * T, H * D)
    ref.backward(go.double())
    _close(o, ref, 2e-6)
    _close(qkv.grad, q64.grad, 5e-6)


@pytest.mark.parametrize('M,C', [(2944, 384), (8192, 384), (37, 128), (100, 2048), (5, 4)])
@pytest.mark.parametrize('with_pos', [True, False])
def test_add_layernorm(M, C, with_pos):
    from point_dae_amd import nn_ops
    ln = torch.nn.LayerNorm(C).cuda()
    ln.weight.data.uniform_(0.5, 1.5)
    ln.bias.data.normal_()
    x = torch.randn(M, C, device='cuda', requires_grad=True)
    pos = torch.randn(M, C, device='cuda', requires_grad=True) if with_pos else None
    gs, gy = torch.randn(M, C, device='cuda'), torch.randn(M, C, device='cuda')
    s, y = nn_ops.add_layer_norm(x, pos, ln)
    (s * gs).sum().backward(retain_graph=True) if with_pos else None
    (y * gy).sum().backward()
    got = [x.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()] + ([pos.grad.clone()] if with_pos else [])
    x64 = x.detach().double().requires_grad_(True)
    p64 = pos.detach().double().requires_grad_(True) if with_pos else None
    ln64 = torch.nn.LayerNorm(C).cuda().double()
    ln64.load_state_dict(ln.state_dict())
    s64 = x64 + p64 if with_pos else x64
    y64 = ln64(s64)
    loss = (y64 * gy.double()).sum() + ((s64 * gs.double()).sum() if with_pos else 0)
    loss.backward()
    _close(y, y64, 2e-6)
    _close(s, s64, 1e-6)
    want = [x64.grad, ln64.weight.grad, ln64.bias.grad] + ([p64.grad] if with_pos else [])
    for a, b in zip(got, want):
        _close(a, b, 2e-5)


def test_gelu_and_scale_residual():
    from point_dae_amd import nn_ops
    z = torch.randn(8192, 1536, device='cuda', requires_grad=True)
    gh = torch.randn_like(z)
    h = nn_ops.gelu(z)
    h.backward(gh)
    bias = torch.randn(1536, device='cuda', requires_grad=True)
    z2 = z.detach().clone().requires_grad_(True)
    nn_ops.begin_step(z.device)
    h2 = nn_ops.bias_gelu(z2, bias)
    h2.backward(gh)
    zb = (z.detach().double() + bias.detach().double()).requires_grad_(True)
    F.gelu(zb).backward(gh.double())
    _close(h2, F.gelu(zb), 1e-6)
    _close(z2.grad, zb.grad, 2e-6)
    _close(bias.grad, zb.grad.sum(0), 2e-5)
    z64 = z.detach().double().requires_grad_(True)
    F.gelu(z64).backward(gh.double())
    _close(h, F.gelu(z64), 1e-6)
    _close(z.grad, z64.grad, 2e-6)
    B, T, C = 16, 23, 384
    a = torch.randn(B * T, C, device='cuda', requires_grad=True)
    bias = torch.randn(C, device='cuda', requires_grad=True)
    res = torch.randn(B * T, C, device='cuda', requires_grad=True)
    keep = (torch.rand(B, device='cuda') > 0.3).float() / 0.7
    gy = torch.randn(B * T, C, device='cuda')
    for kp in (keep, None):
        for t in (a, bias, res):
            t.grad = None
        y = nn_ops._ScaleResidual.apply(a, bias, kp, res, T)
        y.backward(gy)
        k = kp.repeat_interleave(T).unsqueeze(1) if kp is not None else 1.0
        ref = res + k * (a + bias)
        _close(y, ref.detach(), 1e-6)
        _close(a.grad, (gy * k), 1e-6)
        _close(res.grad, gy, 0)
        _close(bias.grad, (gy * k).double().sum(0), 1e-5)


def test_transformer_block_matches_pytorch():
    """nn_ops.transformer_block against the reference composition (oracle/model.py Block)."""
    from oracle.model import Block as RefBlock
    from point_dae_amd import nn_ops
    from point_dae_amd.point_cae_transformer import Block
    torch.manual_seed(0)
    B, T, C = 32, 23, 384
    ref = RefBlock(C, 6, 0.0).cuda().double()
    mine = Block(C, 6, 0.0).cuda()
    mine.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    x = torch.randn(B * T, C, device='cuda', requires_grad=True)
    pos = torch.randn(B * T, C, device='cuda', requires_grad=True)
    gy = torch.randn(B * T, C, device='cuda')
    y = mine(x, pos, B, T, (None, None))
    y.backward(gy)
    x64, p64 = x.detach().double().requires_grad_(True), pos.detach().double().requires_grad_(True)
    y64 = ref((x64 + p64).reshape(B, T, C)).reshape(B * T, C)
    y64.backward(gy.double())
    _close(y, y64, 5e-6)
    _close(x.grad, x64.grad, 2e-5)
    _close(pos.grad, p64.grad, 2e-5)
    for (n, a), (_, b) in zip(sorted(mine.named_parameters()), sorted(ref.named_parameters())):
        _close(a.grad, b.grad, 5e-5)


def test_flat_adamw_matches_torch():
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.optim import FlatAdamW
    import copy
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 64), torch.nn.LayerNorm(64), torch.nn.Linear(64, 13)).cuda()
    ref = copy.deepcopy(net)
    model = FlatDataParallel(net)
    opt = FlatAdamW(model, lr=1e-2, weight_decay=0.05)
    decay = [p for n, p in ref.named_parameters() if p.dim() > 1]
    no_decay = [p for n, p in ref.named_parameters() if p.dim() <= 1]
    ropt = torch.optim.AdamW([{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': 0.05}], lr=1e-2)
    for step in range(5):
        x = torch.randn(16, 37, device='cuda')
        for mdl, o in ((model, opt), (ref, ropt)):
            loss = (mdl(x) ** 2).mean()
            loss.backward()
            o.step()
            o.zero_grad()
        if step == 2:
            for g in opt.param_groups + ropt.param_groups:
                g['lr'] = 3e-3
    for a, b in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a - b).abs().max()
    # the optimiser state travels both ways in torch.optim.AdamW's layout
    sd = opt.state_dict()
    assert len(sd['param_groups']) == 2 and all(int(v['step']) == 5 for v in sd['state'].values())
    rsd = ropt.state_dict()
    assert sorted(sd['state']) == sorted(rsd['state'])
    assert [g['params'] for g in sd['param_groups']] == [g['params'] for g in rsd['param_groups']]
    for i in rsd['state']:
        assert torch.allclose(sd['state'][i]['exp_avg'], rsd['state'][i]['exp_avg'], rtol=1e-5, atol=1e-7)
        assert torch.allclose(sd['state'][i]['exp_avg_sq'], rsd['state'][i]['exp_avg_sq'], rtol=1e-5, atol=1e-9)
    net2, ref2 = copy.deepcopy(ref), copy.deepcopy(net)          # swap: torch state -> flat, flat state -> torch
    model2 = FlatDataParallel(net2)
    opt2 = FlatAdamW(model2, lr=1e-2, weight_decay=0.05)
    opt2.load_state_dict(rsd)
    ropt2 = torch.optim.AdamW([{'params': [p for p in ref2.parameters() if p.dim() <= 1], 'weight_decay': 0.},
                               {'params': [p for p in ref2.parameters() if p.dim() > 1], 'weight_decay': 0.05}], lr=1e-2)
    ropt2.load_state_dict(sd)
    assert opt2.steps == 5 and opt2.param_groups[0]['lr'] == 3e-3 and ropt2.param_groups[1]['lr'] == 3e-3
    x = torch.randn(16, 37, device='cuda')
    for mdl, o in ((model2, opt2), (ref2, ropt2), (model, opt), (ref, ropt)):
        (mdl(x) ** 2).mean().backward()
        o.step()
        o.zero_grad()
    for a, b, c, d in zip(net2.parameters(), ref2.parameters(), net.parameters(), ref.parameters()):
        assert torch.allclose(a, d, rtol=1e-5, atol=1e-6) and torch.allclose(b, c, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('M,N', [(262144, 128), (8192, 64), (2944, 384), (100, 1536), (5, 4), (8192, 512)])
def test_colsum(M, N):
    from point_dae_amd import _lib
    x = torch.randn(M, N, device='cuda')
    out = torch.empty(N, device='cuda')
    _lib.call('pdae_colsum', x, M, N, x.data_ptr(), out.data_ptr(), 0)
    _close(out, x.double().sum(0), 1e-5)
    _lib.call('pdae_colsum', x, M, N, x.data_ptr(), out.data_ptr(), 1)       # accumulate
    _close(out, 2 * x.double().sum(0), 1e-5)


@pytest.mark.parametrize('B,T,C', [(16, 23, 384), (128, 64, 384), (3, 5, 128), (2, 7, 2048)])
@pytest.mark.parametrize('with_keep,with_pos', [(True, True), (True, False), (False, True), (False, False)])
def test_residual_layernorm_equals_unfused(B, T, C, with_keep, with_pos):
    """The fused tail (bias + DropPath + residual (+ pos) + LayerNorm, one launch each way) against
    the unfused pair scale_residual -> add_layer_norm: identical forward bits, same gradients."""
    from point_dae_amd import nn_ops
    torch.manual_seed(B * 100 + T)
    M = B * T
    ln = torch.nn.LayerNorm(C).cuda()
    ln.weight.data.uniform_(0.5, 1.5)
    ln.bias.data.normal_()
    a0, res0 = torch.randn(M, C, device='cuda'), torch.randn(M, C, device='cuda')
    bias0 = torch.randn(C, device='cuda')
    pos0 = torch.randn(M, C, device='cuda') if with_pos else None
    keep = (torch.rand(B, device='cuda') > 0.3).float() / 0.7 if with_keep else None
    gs, gy = torch.randn(M, C, device='cuda'), torch.randn(M, C, device='cuda')
    outs = []
    for fused in (True, False):
        a, res, bias = (t.clone().requires_grad_(True) for t in (a0, res0, bias0))
        pos = pos0.clone().requires_grad_(True) if with_pos else None
        ln.zero_grad()
        nn_ops.begin_step(a.device)
        if fused:
            s, y = nn_ops.residual_layer_norm(nn_ops.Pending(a, bias, keep, res, T), pos, ln)
        else:
            s, y = nn_ops.add_layer_norm(nn_ops._ScaleResidual.apply(a, bias, keep, res, T), pos, ln)
        (s * gs).sum().backward(retain_graph=True)
        (y * gy).sum().backward()
        outs.append([s.detach(), y.detach(), a.grad, res.grad, bias.grad, pos.grad if with_pos else None,
                     ln.weight.grad.clone(), ln.bias.grad.clone()])
    f, u = outs
    assert torch.equal(f[0], u[0]) and torch.equal(f[1], u[1])
    for x, y_ in zip(f[2:], u[2:]):
        if x is not None:
            _close(x, y_, 2e-5)


@pytest.mark.parametrize('B,T,tail,C', [(128, 64, 41, 384), (3, 7, 7, 8), (5, 9, 1, 12)])
def test_tail_rows_gather_and_scatter(B, T, tail, C):
    """The decoder's last block works on the returned (last `tail`) tokens of every sample only
    (models/PointCAE_transformer.py:225-232): one gather launch for up to two tensors, one scatter launch back into
    full-size gradients that are ZERO outside the tail -- against plain slicing."""
    from point_dae_amd import _lib
    g = torch.Generator(device='cuda').manual_seed(B * 100 + T)
    a = torch.randn(B * T, C, device='cuda', generator=g)
    b = torch.randn(B * T, C, device='cuda', generator=g)
    a_t = torch.full((B * tail, C), float('nan'), device='cuda')
    b_t = torch.full((B * tail, C), float('nan'), device='cuda')
    _lib.call('pdae_tail_rows_gather', a, B, T, tail, C, a.data_ptr(), b.data_ptr(), a_t.data_ptr(), b_t.data_ptr())
    assert torch.equal(a_t, a.view(B, T, C)[:, T - tail:].reshape(B * tail, C))
    assert torch.equal(b_t, b.view(B, T, C)[:, T - tail:].reshape(B * tail, C))
    one = torch.full((B * tail, C), float('nan'), device='cuda')
    _lib.call('pdae_tail_rows_gather', a, B, T, tail, C, a.data_ptr(), None, one.data_ptr(), None)
    assert torch.equal(one, a_t)
    da = torch.full((B * T, C), float('nan'), device='cuda')
    db = torch.full((B * T, C), float('nan'), device='cuda')
    _lib.call('pdae_tail_rows_scatter', a, B, T, tail, C, a_t.data_ptr(), b_t.data_ptr(), da.data_ptr(), db.data_ptr())
    want = torch.zeros(B, T, C, device='cuda')
    want[:, T - tail:] = a_t.view(B, tail, C)
    assert torch.equal(da, want.view(B * T, C))
    want[:, T - tail:] = b_t.view(B, tail, C)
    assert torch.equal(db, want.view(B * T, C))
    with pytest.raises(RuntimeError, match='tail'):
        _lib.call('pdae_tail_rows_gather', a, B, T, T + 1, C, a.data_ptr(), None, one.data_ptr(), None)
