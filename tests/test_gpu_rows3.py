"""The exact-split bf16 arithmetic of the row-GEMM family (csrc/rows3_kernel.h, include/pdae.h PDAE_GEMM_BF16X3) against
the fp32-input MFMA kernels it replaces as the default (the reference runs these products through cuBLAS sgemm:
models/PointCAE_transformer.py:94-158, models/PointCAE_pointnetv2.py:135-173).

Admission gate: on EVERY product of the cfg3, published-variant and cfg2 optimisation steps (tests/golden/
gemm_shapes.json, recorded at the C boundary by tools/dump_gemm_shapes.py) the error against an fp64 product is at or
below the fp32-input MFMA kernel's on the same inputs.  Two statistics of |C - C64| / max |C64|: the RMS over the
output (stable: measured ratio exact-split / fp32-input 0.5-0.65 on the row GEMMs) must not exceed the fp32-input
kernel's; the MAXIMUM (an extreme value of ~1e6 samples: +-30 % from seed to seed for either kernel, tools/lab notes in
DESIGN.md) must not exceed it by more than that noise (x 1.25)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, 'tests', 'golden', 'gemm_shapes.json')) as _f:
    _REC = json.load(_f)
GEMMS = sorted({tuple(s[:4]) for wl in _REC.values() for s in wl['gemm']})          # (M, N, K, w_kn)
WGRADS = sorted({tuple(s[:3]) for wl in _REC.values() for s in wl['wgrad']})         # (M, N, K)
F32, BF16X3 = 0, 1


def _lib():
    from point_dae_amd import _lib
    return _lib


@pytest.fixture(autouse=True)
def _restore_arith():
    L = _lib()
    before = L.gemm_arith()
    yield
    L.set_gemm_arith(before)


def _err(c, ref):
    return (c.double() - ref).abs().max().item() / ref.abs().max().item()


def _rms(c, ref):
    return (c.double() - ref).pow(2).mean().sqrt().item() / ref.abs().max().item()


def _gemm(L, arith, x, w, w_kn, cfg=-1):
    L.set_gemm_arith(arith)
    M, K = x.shape
    N = w.shape[1] if w_kn else w.shape[0]
    y = torch.full((M, N), float('nan'), device='cuda')
    L.call('pdae_rows_gemm', x, M, N, K, x.data_ptr(), w.data_ptr(), int(w_kn), None, 0, None, y.data_ptr(), cfg, 1, 0)
    return y


@pytest.mark.parametrize('M,N,K,w_kn', GEMMS)
def test_every_step_product_is_at_least_as_accurate_as_the_fp32_mfma_kernel(M, N, K, w_kn):
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(M * 7 + N * 3 + K + w_kn)
    rows = min(M, 32768)                     # the error of a row does not depend on the number of rows: bound the fp64 work
    x = torch.randn(rows, K, device='cuda', generator=g)
    w = torch.randn((K, N) if w_kn else (N, K), device='cuda', generator=g) / K ** 0.5
    ref = x.double() @ (w.double() if w_kn else w.double().t())
    y32 = _gemm(L, F32, x, w, w_kn)
    e32, r32 = _err(y32, ref), _rms(y32, ref)
    y3 = _gemm(L, BF16X3, x, w, w_kn)
    e3, r3 = _err(y3, ref), _rms(y3, ref)
    if K % 32 != 0:                          # (the exact-split kernels take whole 32-deep tiles: same kernel either way)
        assert torch.equal(y3, y32)
    else:
        assert L.rows_gemm_plan(rows, N, K, bool(w_kn), False)[0] >= 16, 'the default plan is an exact-split tile shape'
        assert r3 <= r32, (r3, r32)
        assert e3 <= 1.25 * e32, (e3, e32)
    assert e3 <= 2e-6


@pytest.mark.parametrize('M,N,K', WGRADS)
def test_every_step_weight_gradient_is_at_least_as_accurate(M, N, K):
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(M + N * 5 + K * 11)
    dy = torch.randn(M, N, device='cuda', generator=g)
    x = torch.randn(M, K, device='cuda', generator=g)
    ref = dy.double().t() @ x.double()
    refb = dy.double().sum(0)
    errs, rms = {}, {}
    for arith in (F32, BF16X3):
        L.set_gemm_arith(arith)
        dw = torch.full((N, K), float('nan'), device='cuda')
        db = torch.full((N,), float('nan'), device='cuda')
        L.rows_wgrad_multi([(dy, x, dw, db)])
        errs[arith], rms[arith] = _err(dw, ref), _rms(dw, ref)
        assert _err(db, refb) <= 2e-6
    # a single layer's rows are dealt to a whole residency of blocks: chains of 32-64 rows, whose partial tiles the
    # SAME ordered fp32 reduction then adds in both arithmetics -- the shared reduction dominates and the two errors
    # nearly tie (measured ratio 0.5-0.96); the step's own grouped launches are the next test
    assert rms[BF16X3] <= 1.05 * rms[F32], rms
    assert errs[BF16X3] <= 1.25 * errs[F32], errs


@pytest.mark.parametrize('M,blocks', [(3584, 12), (8192, 4), (1664, 12)])
def test_a_stack_s_grouped_weight_gradients_are_more_accurate(M, blocks):
    """The launch the step issues: every Linear of every block of a stack in ONE grouped launch -- a tile's rows then
    lie in one or two blocks' ranges (chains of thousands of rows), where the exact-split kernel's two accumulator
    sets pay: every layer's error at or below the fp32-input kernel's, the worst one by a third or more."""
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(M)
    layers = [(1152, 384), (384, 384), (1536, 384), (384, 1536)] * blocks
    jobs = [(torch.randn(M, n, device='cuda', generator=g), torch.randn(M, k, device='cuda', generator=g),
             torch.empty(n, k, device='cuda'), torch.empty(n, device='cuda') if n == 1536 else None) for n, k in layers]
    errs = {}
    for arith in (F32, BF16X3):
        L.set_gemm_arith(arith)
        L.rows_wgrad_multi(jobs)
        errs[arith] = [_rms(dw, dy.double().t() @ x.double()) for dy, x, dw, _ in jobs[::7]]
    assert all(a <= b for a, b in zip(errs[BF16X3], errs[F32])), errs
    assert max(errs[BF16X3]) <= 0.67 * max(errs[F32]), errs


@pytest.mark.parametrize('cfg', [16, 17, 18, 19])
@pytest.mark.parametrize('M,N,K,w_kn', [(3584, 1152, 384, 0), (300, 100, 64, 0), (2944, 384, 1536, 1), (129, 388, 96, 1)])
def test_every_tile_shape_against_fp64(cfg, M, N, K, w_kn):
    L = _lib()
    g = torch.Generator(device='cuda').manual_seed(cfg + M)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn((K, N) if w_kn else (N, K), device='cuda', generator=g) / K ** 0.5
    ref = x.double() @ (w.double() if w_kn else w.double().t())
    assert _err(_gemm(L, BF16X3, x, w, w_kn, cfg), ref) <= 1e-6


def test_identity_weight_returns_the_operand_bit_for_bit():
    """x = h + m + l exactly and 1.0 = (1, 0, 0): the products hh + (mh + lh) reassemble every element of A -- over the
    whole exponent range, signs, and values with long runs of zero or one bits in the mantissa."""
    L = _lib()
    K = 256
    g = torch.Generator(device='cuda').manual_seed(3)
    bits = torch.randint(0, 2 ** 31 - 1, (1024, K), device='cuda', generator=g, dtype=torch.int64)
    expo = torch.randint(20, 235, (1024, K), device='cuda', generator=g, dtype=torch.int64)      # finite, normal
    mant = bits & 0x7fffff
    mant[::3] &= 0x7f00ff                    # sparse mantissas: residuals with leading zeros
    mant[1::3] |= 0x00ffff
    word = ((bits >> 23) & 1) << 31 | expo << 23 | mant
    x = word.to(torch.int32).view(torch.float32).contiguous()
    eye = torch.eye(K, device='cuda')
    for w_kn in (0, 1):
        y = _gemm(L, BF16X3, x, eye, w_kn)
        assert torch.equal(y.view(torch.int32), x.view(torch.int32))


def test_fp32_mfma_kernels_stay_selectable():
    """pdae_set_gemm_arith / PDAE_GEMM=f32mfma: plans answer with the fp32-input tile shapes (cfg < 16), the weight
    gradients with their 128 x 384 tiles, and a block's products agree with the default arithmetic to rounding."""
    L = _lib()
    L.set_gemm_arith(F32)
    assert L.gemm_arith() == F32 and L.rows_gemm_plan(3584, 1152, 384, False, False)[0] < 16
    x = torch.randn(3584, 384, device='cuda')
    w = torch.randn(1152, 384, device='cuda') / 384 ** 0.5
    a = _gemm(L, F32, x, w, 0)
    b = _gemm(L, BF16X3, x, w, 0)
    assert L.rows_gemm_plan(3584, 1152, 384, False, False)[0] >= 16
    assert (a - b).abs().max().item() <= 2e-6 * a.abs().max().item()
    code = ('import sys; sys.path.insert(0, %r); from point_dae_amd import _lib; '
            'print(_lib.gemm_arith(), _lib.rows_gemm_plan(3584, 1152, 384, False, False)[0])' % ROOT)
    env = dict(os.environ, PDAE_GEMM='f32mfma')
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out[0] == '0' and int(out[1]) < 16


def test_split_k_slabs_without_a_tile_of_the_reduction_are_refused():
    """pdae_rows_gemm with more split-K slabs than 32-deep tiles of K (K = 128, 8 slabs: slabs 4..7 would start past
    the end of the rows) returns a bad-argument status instead of launching; the largest legal count still works and
    its slabs add up to the unsplit product."""
    L = _lib()
    L.set_gemm_arith(BF16X3)
    M, N, K = 256, 384, 128
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g)
    y = torch.empty(8, M, N, device='cuda')
    with pytest.raises(RuntimeError, match='split-K slabs'):
        L.call('pdae_rows_gemm', x, M, N, K, x.data_ptr(), w.data_ptr(), 0, None, 0, None, y.data_ptr(), 16, 8, 0)
    L.call('pdae_rows_gemm', x, M, N, K, x.data_ptr(), w.data_ptr(), 0, None, 0, None, y.data_ptr(), 16, 4, 0)
    one = _gemm(L, BF16X3, x, w, False, 16)
    assert (y[:4].sum(0) - one).abs().max().item() <= 1e-5 * one.abs().max().item()
