"""Deterministic mode (include/pdae.h: pdae_set_deterministic; PDAE_DETERMINISTIC=1).

With the workspace registered every float-atomic reduction of the Transformer pretraining step
(batch-norm statistics, the embedder's weight / bias gradients, LayerNorm parameter gradients,
column sums) becomes per-block partials + an ordered pass, so
  * the same launch twice gives the same bits,
  * the hipGraph-replayed step equals the eager launch of the same step body bit for bit,
  * the visible-groups embedder path equals all-groups-then-select: forward bit for bit,
    gradients to <= 1e-5 in the max norm.
The default (atomic) mode is held to the looser bounds in test_gpu_gemm.py / test_gpu_model.py.
"""
import copy
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def det():
    from point_dae_amd import _lib
    _lib.set_deterministic(True)
    assert _lib.deterministic()
    yield
    _lib.set_deterministic(False)
    assert not _lib.deterministic()


def _embedder(seed):
    import torch.nn as nn
    torch.manual_seed(seed)
    first = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True), nn.Conv1d(128, 256, 1)).cuda()
    second = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, 384, 1)).cuda()
    return first.train(), second.train()


def _embed_run(pts, first, second, go, sel=None):
    from point_dae_amd.patch_embed import patch_embed
    for m in (first, second):
        m.zero_grad()
    out = patch_embed(pts, first, second, True, sel) if sel is not None else patch_embed(pts, first, second, True)
    out.backward(go)
    return out.detach().clone(), [p.grad.clone() for p in list(first.parameters()) + list(second.parameters())]


@pytest.mark.parametrize('BG', [64 * 128, 200])
def test_embedder_bit_reproducible(det, BG):
    """Patch embedder forward + backward (conv1 fp64 statistics, conv3 statistics epilogue, the four
    split-M weight gradients and their bias column sums, both BatchNorm backward reductions): the
    same call twice gives identical bits."""
    first, second = _embedder(1)
    torch.manual_seed(2)
    pts = torch.randn(BG, 32, 3, device='cuda') * 0.2
    go = torch.randn(BG, 384, device='cuda')
    f2, s2 = copy.deepcopy(first), copy.deepcopy(second)
    out_a, g_a = _embed_run(pts, first, second, go)
    out_b, g_b = _embed_run(pts, f2, s2, go)
    assert torch.equal(out_a, out_b)
    for a, b in zip(g_a, g_b):
        assert torch.equal(a, b)
    for a, b in zip(list(first.buffers()) + list(second.buffers()), list(f2.buffers()) + list(s2.buffers())):
        assert torch.equal(a, b)


def test_embedder_matches_atomic_mode(det):
    """Same arithmetic per block in both modes: the deterministic results sit inside the atomic
    mode's own noise band."""
    from point_dae_amd import _lib
    first, second = _embedder(3)
    torch.manual_seed(4)
    BG = 2048
    pts = torch.randn(BG, 32, 3, device='cuda') * 0.2
    go = torch.randn(BG, 384, device='cuda')
    f2, s2 = copy.deepcopy(first), copy.deepcopy(second)
    out_d, g_d = _embed_run(pts, first, second, go)
    _lib.set_deterministic(False)
    out_a, g_a = _embed_run(pts, f2, s2, go)
    _lib.set_deterministic(True)
    assert (out_d - out_a).abs().max().item() <= 1e-5 * out_a.abs().max().item()
    gmax = max(g.abs().max().item() for g in g_a)
    for d, a in zip(g_d, g_a):
        if a.abs().max().item() < 1e-4 * gmax:
            continue
        assert ((d - a).norm() / a.norm()).item() <= 5e-3


def test_visible_groups_equal_all_groups(det):
    """patch_embed over the visible groups only == all groups then select: forward bit for bit, every
    gradient to 1e-5 of its max (the two paths add the same terms in different orders -- 2/3 of the
    rows are zeros on one side and absent on the other; measured 1.3e-6).  The atomic mode needs
    1e-5 forward and an L2 bound of 5e-3: a last-bit difference in the BatchNorm sums can flip a
    tied arg-max."""
    first, second = _embedder(5)
    torch.manual_seed(6)
    BG = 64 * 32
    pts = torch.randn(BG, 32, 3, device='cuda') * 0.2
    sel = torch.arange(0, BG, 3, device='cuda', dtype=torch.int32)
    gsel = torch.randn(sel.numel(), 384, device='cuda')
    f2, s2 = copy.deepcopy(first), copy.deepcopy(second)
    go_all = torch.zeros(BG, 384, device='cuda')
    go_all[sel.long()] = gsel
    out_a, g_a = _embed_run(pts, first, second, go_all)
    out_b, g_b = _embed_run(pts, f2, s2, gsel, sel)
    assert torch.equal(out_a[sel.long()], out_b)          # the forward statistics are the same sums
    # ... and with the masked groups' share of the backward done by algebra (patch_embed's `masked` list)
    from point_dae_amd.patch_embed import patch_embed
    f3, s3 = copy.deepcopy(first), copy.deepcopy(second)
    for m in (f3, s3):
        m.zero_grad()
    rest = torch.tensor([i for i in range(BG) if i % 3 != 0], device='cuda', dtype=torch.int32)
    out_c = patch_embed(pts, f3, s3, True, sel, rest)
    out_c.backward(gsel)
    assert torch.equal(out_c.detach(), out_b)
    g_c = [p.grad.clone() for p in list(f3.parameters()) + list(s3.parameters())]
    gmax = max(g.abs().max().item() for g in g_a)
    for a, b in zip(g_a, g_b):
        if a.abs().max().item() < 1e-4 * gmax:     # conv bias feeding a training-mode BatchNorm: exactly zero
            continue                               # gradient, rounding residue on both paths
        scale = a.abs().max().item()
        assert (a - b).abs().max().item() <= 1e-5 * scale, (tuple(a.shape), (a - b).abs().max().item(), scale)
    for a, c in zip(g_a, g_c):
        if a.abs().max().item() < 1e-4 * gmax:
            continue
        scale = a.abs().max().item()
        assert (a - c).abs().max().item() <= 1e-5 * scale, ('algebra', tuple(a.shape), (a - c).abs().max().item(), scale)


@pytest.mark.parametrize('M,C', [(2944, 384), (8192, 384), (777, 1024)])
def test_layernorm_backward_bit_reproducible(det, M, C):
    from point_dae_amd import nn_ops
    torch.manual_seed(M)
    x = torch.randn(M, C, device='cuda')
    ln = torch.nn.LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.normal_(), ln.bias.normal_()
    go = torch.randn(M, C, device='cuda')
    outs = []
    for _ in range(2):
        xx = x.clone().requires_grad_()
        ln.zero_grad()
        nn_ops.layer_norm(xx, ln).backward(go)
        outs.append((xx.grad.clone(), ln.weight.grad.clone(), ln.bias.grad.clone()))
    for a, b_ in zip(*outs):
        assert torch.equal(a, b_)
    ref_w = (go.double() * torch.nn.functional.layer_norm(x.double(), (C,))).sum(0)
    assert (outs[0][1].double() - ref_w).abs().max().item() <= 1e-5 * ref_w.abs().max().item()
    assert (outs[0][2].double() - go.double().sum(0)).abs().max().item() <= 1e-5 * go.double().sum(0).abs().max().item()


def test_workspace_too_small_is_refused():
    from point_dae_amd import _lib, nn_ops
    _lib.set_deterministic(True, megabytes=1)
    try:
        x = torch.randn(65536, 1024, device='cuda', requires_grad=True)
        y = nn_ops.layer_norm(x, torch.nn.LayerNorm(1024).cuda())
        with pytest.raises(RuntimeError, match='workspace is too small'):
            y.backward(torch.ones_like(y))
    finally:
        _lib.set_deterministic(False)


def _six_updates_setup():
    from point_dae_amd import builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.synthetic import shapenet_like_clouds
    config = cfg_from_yaml_file(os.path.join(
        ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    config.model.transformer_config.drop_path_rate = 0.0
    config.model.transformer_config.depth = 3
    config.model.transformer_config.decoder_depth = 2
    torch.manual_seed(0)
    net = builder.model_builder(config.model).cuda().train()
    B = 16
    x = torch.from_numpy(shapenet_like_clouds(B * 2, 1024, seed=4)).cuda().split(B)
    return config, net, B, x


def _seed(s):
    random.seed(s), np.random.seed(s), torch.manual_seed(s)


def test_graphed_step_equals_eager_step_exactly(det):
    """hipGraph replays vs eager launches of THE SAME step body (the cfg3 optimisation step, FPS ... AdamW), six
    updates: the same kernels in the same order on the same inputs, so every loss and every parameter agree BIT FOR
    BIT -- the header's promise for the deterministic mode (the atomic mode's 2e-3 / 2e-2 bounds in test_gpu_model.py
    are summation-order noise amplified by AdamW, nothing else)."""
    from point_dae_amd import builder
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedTrainStep
    config, net_a, B, x = _six_updates_setup()
    net_b = copy.deepcopy(net_a)
    model_a = FlatDataParallel(net_a)
    opt_a, _ = builder.build_opti_sche(model_a, config)
    model_a.zero_grad()
    # the step body launched kernel by kernel (never captured) ...
    eager_step = GraphedTrainStep(model_a, opt_a, config, B, 1024, warmup_eager=1000)
    _seed(123)
    eager = [eager_step(x[i % 2])[0].item() for i in range(6)]
    assert not eager_step.graphs
    # ... against its hipGraph replays
    model_b = FlatDataParallel(net_b)
    opt_b, _ = builder.build_opti_sche(model_b, config)
    step = GraphedTrainStep(model_b, opt_b, config, B, 1024, warmup_eager=1)
    _seed(123)
    graphed = [step(x[i % 2])[0].item() for i in range(6)]
    assert len(step.graphs) >= 1
    assert eager == graphed, (eager, graphed)
    assert torch.equal(model_a.flat_param, model_b.flat_param)


def test_plain_autograd_step_tracks_the_graphed_step_within_amplified_rounding(det):
    """A TOLERANCE test, not an identity: runner_pretrain.train_step (plain autograd: weight gradients reduced per
    block) against the graphed step (per stack) -- two fixed but DIFFERENT summation orders, so the first update agrees
    to fp32 rounding and six AdamW updates amplify that.  Bound derived from the arithmetic: one update moves a weight
    by lr = 1e-3 times a sign-like ratio m / sqrt(v), so a relative gradient difference of ~1e-6 (exact-split products,
    fp64 BatchNorm sums) becomes ~1e-5 of the loss per update; measured 5.4e-5 at the sixth update, bound 1e-4."""
    from point_dae_amd import builder
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedTrainStep
    from point_dae_amd.runner_pretrain import train_step
    config, net_b, B, x = _six_updates_setup()
    net_c = copy.deepcopy(net_b)
    model_b = FlatDataParallel(net_b)
    opt_b, _ = builder.build_opti_sche(model_b, config)
    step = GraphedTrainStep(model_b, opt_b, config, B, 1024, warmup_eager=1)
    _seed(123)
    graphed = [step(x[i % 2])[0].item() for i in range(6)]
    model_c = FlatDataParallel(net_c)
    opt_c, _ = builder.build_opti_sche(model_c, config)
    model_c.zero_grad()
    _seed(123)
    plain = [train_step(model_c, opt_c, config, x[i % 2], x[i % 2])[0].item() for i in range(6)]
    assert abs(plain[0] - graphed[0]) <= 2e-6 * abs(plain[0]), (plain, graphed)       # before any update: rounding only
    for a, b in zip(plain, graphed):
        assert abs(a - b) <= 1e-4 * abs(a), (plain, graphed)


@pytest.mark.parametrize('name', ['PointCAE_transformer', 'PointCAE_transformer_fc_global_folding_local'])
def test_two_phase_step_equals_single_phase(det, name):
    """GraphedTrainStep(split=True) -- graph 1 stops at a leaf copy of the patch tokens, graph 2 is the
    embedder's backward -- gives the same parameters as the one-graph step, bit for bit, for both
    Transformer model classes (one rank: the collectives in between are skipped)."""
    from point_dae_amd import builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedTrainStep
    from point_dae_amd.synthetic import shapenet_like_clouds
    config = cfg_from_yaml_file(os.path.join(
        ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    config.model.NAME = name
    config.model.transformer_config.depth = 2
    config.model.transformer_config.decoder_depth = 2
    torch.manual_seed(0)
    net_a = builder.model_builder(config.model).cuda().train()
    net_b = copy.deepcopy(net_a)
    B = 8
    x = torch.from_numpy(shapenet_like_clouds(B * 2, 1024, seed=9)).cuda().split(B)
    out = []
    for net, split in ((net_a, False), (net_b, True)):
        model = FlatDataParallel(net)
        assert model.late_ranges and model.early_range[1] - model.early_range[0] > 0
        opt, _ = builder.build_opti_sche(model, config)
        step = GraphedTrainStep(model, opt, config, B, 1024, warmup_eager=1, split=split)
        assert step.split == split
        random.seed(5), np.random.seed(5), torch.manual_seed(5)
        losses = [step(x[i % 2])[0].item() for i in range(5)]
        out.append((losses, model.flat_param.clone()))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])


def test_deterministic_flag_of_the_cli_switches_the_library_mode():
    """`--deterministic` (main.py -> misc.set_random_seed) registers the ordered-reduction workspace."""
    from point_dae_amd import _lib
    from point_dae_amd.misc import set_random_seed
    assert not _lib.deterministic()
    try:
        set_random_seed(3, deterministic=True)
        assert _lib.deterministic()
    finally:
        _lib.set_deterministic(False)
        torch.backends.cudnn.deterministic = False


def test_deferred_layernorm_reductions():
    """pdae_deferred_begin / _flush: LayerNorm backward calls park their parameter-gradient partials and ONE
    launch adds them -- same sums as the atomic path (1e-6), same bits twice, outputs accumulate (+=), and a
    call that does not fit the workspace falls back to the usual path."""
    from point_dae_amd import _lib, nn_ops
    torch.manual_seed(7)
    lns = [torch.nn.LayerNorm(384).cuda() for _ in range(5)]
    xs = [torch.randn(m, 384, device='cuda') for m in (3584, 8192, 1664, 100, 16)]
    gos = [torch.randn_like(x) for x in xs]

    def run(deferred, small=False):
        for ln in lns:
            ln.zero_grad()
        ys = [nn_ops.layer_norm(x.clone().requires_grad_(), ln) for x, ln in zip(xs, lns)]
        if deferred and small:           # 1 MiB: only some of the calls fit, the others take the usual path
            _lib._check(_lib.lib(), 'pdae_deferred_begin', _lib.lib().pdae_deferred_begin(_lib._def_ws.data_ptr(), 1 << 20))
        elif deferred:
            _lib.deferred_begin()
        for y, go in zip(ys, gos):
            y.backward(go)
        if deferred:
            _lib.deferred_flush(xs[0])
        return [(ln.weight.grad.clone(), ln.bias.grad.clone()) for ln in lns]

    ref = run(False)
    a, b = run(True), run(True)
    for (rw, rb), (aw, ab), (bw, bb) in zip(ref, a, b):
        assert torch.equal(aw, bw) and torch.equal(ab, bb)
        assert (aw - rw).abs().max().item() <= 2e-6 * rw.abs().max().item()
        assert (ab - rb).abs().max().item() <= 2e-6 * rb.abs().max().item()
    again = run(False)                                               # the mode is off again after the flush
    small = run(True, small=True)
    for (rw, rb), (gw, gb_), (sw, sb) in zip(ref, again, small):
        assert (gw - rw).abs().max().item() <= 2e-6 * rw.abs().max().item()
        assert (sw - rw).abs().max().item() <= 2e-6 * rw.abs().max().item()
        assert (sb - rb).abs().max().item() <= 2e-6 * rb.abs().max().item()


@pytest.mark.parametrize('mode', ['atomic', 'deterministic'])
def test_graphed_step_gradients_equal_eager_gradients(mode):
    """One step's gradients, every parameter: the graphed step's path (gradient sink for the blocks' weights,
    parked LayerNorm / weight-gradient reductions, one flush) against plain eager autograd into the flat buffer.
    Catches a gradient that was read (or cloned by autograd) before its deferred reduction ran."""
    from point_dae_amd import _lib, builder
    from point_dae_amd.config import cfg_from_yaml_file
    from point_dae_amd.data_parallel import FlatDataParallel
    from point_dae_amd.graph_step import GraphedTrainStep
    from point_dae_amd.synthetic import shapenet_like_clouds
    config = cfg_from_yaml_file(os.path.join(
        ROOT, 'cfgs', 'pretrain_PointCAE_transformer_dropout_patch_affine_r3_maskpatch_p0005_whole.yaml'))
    config.model.transformer_config.drop_path_rate = 0.0
    config.model.transformer_config.depth = 3
    config.model.transformer_config.decoder_depth = 2
    if mode == 'deterministic':
        _lib.set_deterministic(True)
    try:
        torch.manual_seed(0)
        net_a = builder.model_builder(config.model).cuda().train()
        net_b = copy.deepcopy(net_a)
        B = 16
        x = torch.from_numpy(shapenet_like_clouds(B, 1024, seed=4)).cuda()
        model_a, model_b = FlatDataParallel(net_a), FlatDataParallel(net_b)
        opt_b, _ = builder.build_opti_sche(model_b, config)
        step = GraphedTrainStep(model_b, opt_b, config, B, 1024, warmup_eager=1)
        random.seed(9), np.random.seed(9), torch.manual_seed(9)
        step.pts.copy_(x)
        tvis = step._draw()
        step._phase1(tvis)                                           # eager launches of the graphed step's body
        nv = B * tvis
        model_a.zero_grad()
        la, lna = model_a(step.pts, step.pts, steps=step.steps, rows=(step.vis[:nv], step.msk[:B * (step.G - tvis)]))
        (la + step.normal_weight * lna.sum()).backward()
        tol = 1e-6 if mode == 'deterministic' else 2e-3      # (atomic mode: summation-order noise, measured 2.5e-4)
        for name, (off, n) in zip(model_a.names, model_a.offsets):
            ga, gb = model_a.flat_grad[off:off + n], model_b.flat_grad[off:off + n]
            scale = ga.abs().max().item()
            if scale < 1e-7:
                assert gb.abs().max().item() < 1e-5, name
                continue
            assert (ga - gb).abs().max().item() <= tol * scale, (name, (ga - gb).abs().max().item(), scale)
    finally:
        if mode == 'deterministic':
            _lib.set_deterministic(False)


def test_two_threads_two_streams_two_contexts():
    """include/pdae.h contexts: the library's mutable state (deterministic workspace, GEMM arithmetic, parked
    reductions) belongs to the calling thread's current context.  Two host threads, each on its own stream and its
    own context -- one deterministic on the fp32-input kernels, one atomic on the exact-split kernels -- run the
    embedder's statistics reductions, a grouped weight gradient and a row GEMM at the same time: each sees its own
    settings throughout, the deterministic thread's results are bit-identical to a single-threaded deterministic run,
    and the default context is untouched.  (Forward and direct calls only: autograd runs backward nodes on ITS thread,
    which is on the default context unless told otherwise -- INTEGRATION.md.)"""
    import threading
    from point_dae_amd import _lib, nn_ops
    from point_dae_amd.patch_embed import patch_embed
    first0, second0 = _embedder(1)
    torch.manual_seed(3)
    pts = torch.randn(64 * 32, 32, 3, device='cuda') * 0.2
    w = torch.randn(1152, 384, device='cuda') / 20
    a = torch.randn(3584, 384, device='cuda')
    dy = torch.randn(3584, 1152, device='cuda')
    torch.cuda.synchronize()

    def work(ctx_det, arith, out, rounds):
        stream = torch.cuda.Stream()
        ctx = _lib.Context()
        first, second = copy.deepcopy(first0), copy.deepcopy(second0)
        try:
            with torch.cuda.stream(stream), ctx, torch.no_grad():
                if ctx_det:
                    ctx.set_deterministic(64)
                _lib.lib().pdae_set_gemm_arith(arith)
                seen = []
                for _ in range(rounds):
                    seen.append((_lib.deterministic(), _lib.gemm_arith(), _lib.rows_gemm_plan(3584, 1152, 384, False, False)[0] >= 16))
                    y = patch_embed(pts, first, second, True)
                    z = nn_ops.rows_gemm(a, w)
                    dws, dbs = nn_ops.rows_wgrad([dy], [a], [True])
                out.update(seen=seen, y=y.clone(), z=z.clone(), dw=dws[0].clone(), db=dbs[0].clone(),
                           rm=second[1].running_mean.clone())
                stream.synchronize()
        except Exception as e:          # surfaced by the main thread
            out['error'] = e
        finally:
            ctx.close()

    single = {}
    work(True, _lib.GEMM_F32MFMA, single, 6)
    assert 'error' not in single, single.get('error')
    before = (_lib.deterministic(), _lib.gemm_arith())
    r1, r2 = {}, {}
    t1 = threading.Thread(target=work, args=(True, _lib.GEMM_F32MFMA, r1, 6))
    t2 = threading.Thread(target=work, args=(False, _lib.GEMM_BF16X3, r2, 6))
    t1.start(); t2.start(); t1.join(); t2.join()
    for r in (r1, r2):
        assert 'error' not in r, r.get('error')
    assert all(s_ == (True, _lib.GEMM_F32MFMA, False) for s_ in r1['seen']), r1['seen']
    assert all(s_ == (False, _lib.GEMM_BF16X3, True) for s_ in r2['seen']), r2['seen']
    for k in ('y', 'z', 'dw', 'db', 'rm'):
        assert torch.equal(r1[k], single[k]), k
    assert (r2['z'] - r1['z']).abs().max().item() <= 2e-6 * r1['z'].abs().max().item()
    assert (_lib.deterministic(), _lib.gemm_arith()) == before
    assert _lib.lib().pdae_ctx_current() is None
