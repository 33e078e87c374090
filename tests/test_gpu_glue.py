"""csrc/glue.hip: the single launches that stand where groups of framework launches stood in the graphed step.  Every entry is
data movement or a fixed-order sum, so the comparisons with torch are exact unless stated."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(*shape, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed + sum(shape))
    return torch.randn(*shape, device='cuda', generator=g)


@pytest.mark.parametrize('P,K,C', [(1024, 3, 128), (1, 3, 128), (37, 5, 33), (0, 3, 16)])
def test_partials_sum_transposed(P, K, C):
    from point_dae_amd import _lib
    part = _rand(P, K, C)
    out = torch.full((C, K), 7.0, device='cuda')
    _lib.call('pdae_partials_sum_t', out, P, K, C, _lib.ptr(part), _lib.ptr(out))
    want = part.double().sum(0).t()                        # (a fixed order of fp32 adds: 16 strided lane sums, then the lanes)
    assert torch.allclose(out.double(), want, rtol=0, atol=2e-6 * max(float(want.abs().max()), 1.0) if P else 0)
    again = torch.empty_like(out)
    _lib.call('pdae_partials_sum_t', out, P, K, C, _lib.ptr(part), _lib.ptr(again))
    assert torch.equal(out, again)


def test_multi_copy_many_ragged_tensors():
    from point_dae_amd import _lib
    sizes = [1, 3, 4, 5, 384, 1536, 2047, 2048, 2049, 196608, 262144 + 1] + [384] * 150     # 161 entries: two launches
    src = [_rand(n, seed=i) for i, n in enumerate(sizes)]
    flat = torch.full((sum(sizes) + 3,), -5.0, device='cuda')
    dst, o = [], 3                                       # (views at odd offsets: the unaligned path)
    for n in sizes:
        dst.append(flat[o:o + n])
        o += n
    _lib.multi_copy(list(zip(dst, src)))
    assert torch.equal(flat[3:], torch.cat(src))
    assert (flat[:3] == -5.0).all()


def test_multi_copy_takes_the_leading_columns_of_a_wider_tile():
    from point_dae_amd import _lib
    tile, wide = _rand(128, 4), _rand(37, 640, seed=3)
    a, b, c = torch.empty(128, 3, device='cuda'), torch.empty(37, 639, device='cuda'), torch.empty(999, device='cuda')
    whole = torch.empty(128, 4, device='cuda')
    s = _rand(999, seed=5)
    z = torch.full((5000,), 3.0, device='cuda')
    _lib.multi_copy([(c, s), (z[1:4098], None)], [(a, tile, 3), (b, wide, 639), (whole, tile, 4)])
    assert torch.equal(a, tile[:, :3]) and torch.equal(b, wide[:, :639]) and torch.equal(c, s) and torch.equal(whole, tile)
    assert float(z[0]) == 3.0 and (z[1:4098] == 0).all() and (z[4098:] == 3.0).all()      # (no source: zero fill)


def test_multi_copy_refuses_what_it_cannot_copy():
    from point_dae_amd import _lib
    a, b = torch.empty(8, device='cuda'), torch.empty(9, device='cuda')
    with pytest.raises(ValueError):
        _lib.multi_copy([(a, b)])
    with pytest.raises(ValueError):
        _lib.multi_copy([(a, torch.empty(8, device='cuda', dtype=torch.float64))])
    with pytest.raises(ValueError):
        _lib.multi_copy([], [(a, torch.empty(4, 4, device='cuda'), 5)])
    _lib.multi_copy([])                                   # nothing to do


@pytest.mark.parametrize('B,G,Tv,C', [(128, 64, 17, 384), (3, 64, 63, 384), (2, 5, 1, 8), (1, 4, 4, 4)])
def test_assemble_tokens_and_its_gradient(B, G, Tv, C):
    from point_dae_amd import nn_ops
    M = G - Tv
    x = _rand(B, Tv, C).requires_grad_(True)
    tok = _rand(1, 1, C, seed=1).requires_grad_(True)
    x2, tok2 = x.detach().clone().requires_grad_(True), tok.detach().clone().requires_grad_(True)
    if M:
        out = nn_ops.assemble_tokens(x, tok, B, Tv, M)
        want = torch.cat([x2, tok2.expand(B, M, -1)], dim=1)
    else:
        pytest.skip('the model skips the assembly when nothing is masked')
    assert torch.equal(out, want)
    w = _rand(B, G, C, seed=2)
    (out * w).sum().backward()
    (want * w).sum().backward()
    assert torch.equal(x.grad, x2.grad)
    # the token's gradient: a column sum over B * M rows in the colsum kernel's order vs torch's
    assert torch.allclose(tok.grad, tok2.grad, rtol=2e-5, atol=2e-5 * float(tok2.grad.abs().max()))


@pytest.mark.parametrize('N,K2', [(512, 256), (12, 5)])
def test_split_conv3_weight(N, K2):
    from point_dae_amd import _lib
    w = _rand(N, 2 * K2)
    wg, wl, wlt = torch.empty(N, K2, device='cuda'), torch.empty(N, K2, device='cuda'), torch.empty(K2, N, device='cuda')
    _lib.call('pdae_embed_split_conv3_weight', w, N, K2, _lib.ptr(w), _lib.ptr(wg), _lib.ptr(wl), _lib.ptr(wlt))
    assert torch.equal(wg, w[:, :K2]) and torch.equal(wl, w[:, K2:]) and torch.equal(wlt, w[:, K2:].t())
    wg.zero_(), wl.zero_()
    _lib.call('pdae_embed_split_conv3_weight', w, N, K2, _lib.ptr(w), _lib.ptr(wg), _lib.ptr(wl), None)
    assert torch.equal(wg, w[:, :K2]) and torch.equal(wl, w[:, K2:])


@pytest.mark.parametrize('Gm,BG,C3,C2', [(6016, 8192, 512, 256), (1, 4, 8, 4), (0, 4, 8, 4)])
def test_masked_prep(Gm, BG, C3, C2):
    from point_dae_amd import _lib
    uv, gb, wl = _rand(2, C3), _rand(BG, C3, seed=1), _rand(C3, C2, seed=2)
    masked = torch.randperm(BG, device='cuda')[:Gm].to(torch.int32)
    xe, wv = torch.empty(Gm, C3, device='cuda'), torch.empty(C3, C2, device='cuda')
    _lib.call('pdae_embed_masked_prep', uv, Gm, C3, C2, _lib.ptr(uv), _lib.ptr(gb), _lib.ptr(masked), _lib.ptr(wl), _lib.ptr(xe),
              _lib.ptr(wv))
    want = (uv[0].double() + gb.index_select(0, masked.long()).double() * uv[1].double())
    assert torch.allclose(xe.double(), want, rtol=0, atol=1e-6 * float(want.abs().max()) if Gm else 0)
    assert torch.equal(wv, wl * uv[1].unsqueeze(1))


@pytest.mark.parametrize('C3,C2', [(512, 256), (8, 4)])
def test_dw3_assemble(C3, C2):
    from point_dae_amd import _lib
    dwg, dwl, wgram, xterm, v = _rand(C3, C2), _rand(C3, C2, seed=1), _rand(C3, C2, seed=2), _rand(C3, C2, seed=3), _rand(C3, seed=4)
    out = torch.empty(C3, 2 * C2, device='cuda')
    _lib.call('pdae_embed_dw3_assemble', out, C3, C2, _lib.ptr(dwg), _lib.ptr(dwl), _lib.ptr(v), _lib.ptr(wgram), _lib.ptr(xterm),
              _lib.ptr(out))
    want = torch.cat([dwg.double(), dwl.double() + v.double().unsqueeze(1) * wgram.double() + xterm.double()], dim=1)
    assert torch.equal(out[:, :C2], dwg)
    assert torch.allclose(out.double(), want, rtol=0, atol=1e-6 * float(want.abs().max()))
    _lib.call('pdae_embed_dw3_assemble', out, C3, C2, _lib.ptr(dwg), _lib.ptr(dwl), None, None, None, _lib.ptr(out))
    assert torch.equal(out, torch.cat([dwg, dwl], dim=1))


def test_entries_refuse_bad_arguments():
    from point_dae_amd import _lib
    t = torch.empty(64, device='cuda')
    for name, args in [('pdae_assemble_tokens', (1, 4, 5, 4, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t))),
                       ('pdae_assemble_tokens', (1, 4, 2, 6, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t))),
                       ('pdae_partials_sum_t', (1, 3, 4, None, _lib.ptr(t))),
                       ('pdae_embed_masked_prep', (1, 6, 4, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), _lib.ptr(t))),
                       ('pdae_embed_dw3_assemble', (4, 4, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), None, None, _lib.ptr(t)))]:
        with pytest.raises(RuntimeError):
            _lib.call(name, t, *args)


def test_a_step_with_the_single_launches_equals_the_step_with_framework_glue(monkeypatch):
    """One forward + backward of the graphed step's body (depth 2 + 1, B = 8, the same mask and affine draws) with the glue as
    library launches (csrc/glue.hip) and with the framework launches they replaced: the losses agree bit for bit (the forward
    glue is data movement) and every gradient within 2e-6 of its tensor's largest entry (the backward's sums are re-associated:
    partials_sum_t, dw3_assemble, the mask token's column sums)."""
    import random
    import numpy as np
    from point_dae_amd import builder, graph_step, nn_ops, patch_embed
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedTrainStep
    from point_dae_amd.synthetic import shapenet_like_clouds
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    config = cfg_from_yaml_file(os.path.join(root, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    config.model.transformer_config.drop_path_rate = 0.0
    config.model.transformer_config.depth = 2
    config.model.transformer_config.decoder_depth = 1
    B = 8
    x = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=9)).cuda()
    out = []
    for glue in (True, False):
        monkeypatch.setattr(patch_embed, 'GLUE', glue)
        monkeypatch.setattr(nn_ops, 'ASSEMBLE', glue)
        monkeypatch.setattr(nn_ops, 'PREDRAW', glue)
        monkeypatch.setattr(graph_step, 'MULTI_COPY', glue)
        torch.manual_seed(0)
        model = FlatDataParallel(builder.model_builder(config.model).cuda().train())
        opt, _ = builder.build_opti_sche(model, config)
        step = GraphedTrainStep(model, opt, config, B, 1024, warmup_eager=1)
        step.pts.copy_(x)
        random.seed(4), np.random.seed(4), torch.manual_seed(4)
        lx, ln = step._fwd_bwd(step._draw())
        torch.cuda.synchronize()
        out.append((lx.clone(), ln.clone(), model.flat_grad.clone(), [(n, o, c) for n, (o, c) in zip(model.names, model.offsets)]))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    gmax = float(out[1][2].abs().max())          # (a conv bias that feeds a training-mode BatchNorm has an analytically zero
    for name, off, n in out[0][3]:               #  gradient: rounding noise of either sign on both sides -> an absolute floor)
        a, b = out[0][2][off:off + n], out[1][2][off:off + n]
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * scale + 1e-7 * gmax, (name, float((a - b).abs().max()), scale, gmax)


@pytest.mark.parametrize('R,C,pr,pc', [(384, 3, 0, 1), (3, 384, 1, 0), (5, 6, 3, 2), (131072, 3, 0, 1)])
def test_pad2d_and_its_gradient(R, C, pr, pc):
    import torch.nn.functional as F
    from point_dae_amd import nn_ops
    x = _rand(R, C).requires_grad_(True)
    y = nn_ops.pad2d(x, pr, pc)
    assert torch.equal(y, F.pad(x.detach(), (0, pc, 0, pr)))
    w = _rand(R + pr, C + pc, seed=1)
    (y * w).sum().backward()
    assert torch.equal(x.grad, w[:R, :C])
    # a column slice of a wider matrix (read through its row stride) and a 1-D tensor
    wide = _rand(R, C + 5, seed=2)
    assert torch.equal(nn_ops.pad2d(wide[:, 2:2 + C], pr, pc), F.pad(wide[:, 2:2 + C], (0, pc, 0, pr)))
    b = _rand(C, seed=3).requires_grad_(True)
    yb = nn_ops.pad2d(b, 0, pc)
    assert torch.equal(yb, F.pad(b.detach(), (0, pc)))
    yb.sum().backward()
    assert torch.equal(b.grad, torch.ones(C, device='cuda'))


@pytest.mark.parametrize('B,T,C', [(128, 23, 384), (3, 1, 8), (2, 255, 5)])
def test_max_plus_mean_and_its_gradient(B, T, C):
    from point_dae_amd import nn_ops
    x = _rand(B, T, C).requires_grad_(True)
    x2 = x.detach().clone().requires_grad_(True)
    out = nn_ops.max_plus_mean(x)
    want = x2.max(dim=1)[0] + x2.mean(1)
    assert torch.allclose(out, want, rtol=0, atol=2e-6 * float(want.abs().max()))
    w = _rand(B, C, seed=1)
    (out * w).sum().backward()
    (want * w).sum().backward()
    assert torch.allclose(x.grad, x2.grad, rtol=0, atol=2e-7 * float(x2.grad.abs().max()))


@pytest.mark.parametrize('R,bounds', [(384, [(0, 384), (384, 386)]), (512, [(0, 2), (2, 5), (5, 1029)]), (8, [(0, 3)]),
                                      (16, [(0, 4), (4, 5), (5, 9), (9, 12)])])
def test_split_weight_cols_and_its_gradient(R, bounds):
    import torch.nn.functional as F
    from point_dae_amd import nn_ops
    C = bounds[-1][1]
    w = _rand(R, C).requires_grad_(True)
    w2 = w.detach().clone().requires_grad_(True)
    got = nn_ops.split_weight_cols(w, bounds)
    want = [F.pad(w2[:, b0:b1], (0, (-(b1 - b0)) % 4)) for b0, b1 in bounds]
    loss_a = loss_b = 0
    for i, (g, x) in enumerate(zip(got, want)):
        assert torch.equal(g, x)
        m = _rand(*g.shape, seed=i)
        if i != 1:                                        # (one block without a gradient: its columns of dW are zero)
            loss_a = loss_a + (g * m).sum()
            loss_b = loss_b + (x * m).sum()
    loss_a.backward()
    loss_b.backward()
    assert torch.equal(w.grad, w2.grad)


@pytest.mark.parametrize('R,C,at', [(64, 3, 3), (128, 131, 3), (256, 259, 3), (5, 7, 2)])
def test_insert_zero_col_and_its_gradient(R, C, at):
    from point_dae_amd import nn_ops
    w = _rand(R, C).requires_grad_(True)
    w2 = w.detach().clone().requires_grad_(True)
    got = nn_ops.insert_zero_col(w, at)
    want = torch.cat([w2[:, :at], w2.new_zeros(R, 1), w2[:, at:]], dim=1)
    assert torch.equal(got, want)
    m = _rand(R, C + 1, seed=1)
    (got * m).sum().backward()
    (want * m).sum().backward()
    assert torch.equal(w.grad, w2.grad)
