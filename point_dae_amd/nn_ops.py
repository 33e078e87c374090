"""Dense layers of the pretraining step on flat (rows, channels) activations.

The model code (point_cae_transformer.py, point_cae_pointnetv2.py) only calls
the functions below.  Every dense layer runs on the hand-written gfx950 kernels:
the Linear layers of the Transformer blocks, pos_embed and the heads on the row
GEMM family of csrc/rows_gemm.hip (fp32 MFMA, fused bias / GELU / GELU'
epilogues, split-K slabs, grouped weight gradients -- no BLAS library in the
step), the patch embedder on csrc/{gemm,embed}.hip (patch_embed.py), and the
attention core, LayerNorm with the position / residual adds, DropPath on
csrc/{attention,block}.hip.  A whole pre-LN block is ONE autograd Function
(_TransformerBlock) whose backward issues the data-gradient chain and then the
block's four weight gradients as one grouped launch.
Parameters are read from the reference-layout nn.Modules that own them.
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib
from .arena import ZeroArena, arena, begin_step  # noqa: F401
from .patch_embed import patch_embed  # noqa: F401  (fused gfx950 embedder)
from .probe import Probe, probed_family, set_probe  # noqa: F401


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


def _colsum(x):
    out, _ = arena.take(x.shape[1], x)
    _lib.call('pdae_colsum', x, x.shape[0], x.shape[1], _lib.ptr(x), _lib.ptr(out), 1)
    return out


class _AddLayerNorm(torch.autograd.Function):
    """(s, y) = (x + pos, LayerNorm(x + pos)); pos may be None (then s is x)."""

    @staticmethod
    def forward(ctx, x, pos, gamma, beta, eps):
        x = x.contiguous()
        if pos is not None:
            pos = pos.contiguous()
        s, y, mean, rstd = _add_ln_forward(x, pos, gamma, beta, eps)
        ctx.save_for_backward(s, mean, rstd, gamma)
        ctx.has_pos = pos is not None
        ctx.mark_non_differentiable(mean, rstd)
        ctx.set_materialize_grads(False)      # an unused `s` must not cost a zero-fill + a read of zeros
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        s, mean, rstd, gamma = ctx.saved_tensors
        M, C = s.shape
        if dy is None:                        # only the pass-through sum was used
            return ds, (ds if ctx.has_pos else None), None, None, None
        dx, dg, db = _ln_backward(dy.contiguous(), s, mean, rstd, gamma, ds.contiguous() if ds is not None else None)
        return dx, (dx if ctx.has_pos else None), dg, db, None


def _slabs(t):
    """Slab count of a GEMM output: (S, M, C) = S split-K slabs of partial products, (M, C) = 1."""
    return t.shape[0] if t.dim() == 3 else 1


def _to_slabs(g, slabs):
    """Gradient w.r.t. S slabs that are summed = the same gradient for each (a view, no copy)."""
    return g if slabs == 1 else g.unsqueeze(0).expand(slabs, *g.shape)


def _from_slabs(g):
    """The (M, C) gradient out of what _to_slabs made (autograd may have materialised it)."""
    return g if g.dim() == 2 else g[0]


def _add_ln_forward(x, pos, gamma, beta, eps):
    M, C = x.shape
    y = torch.empty_like(x)
    mean, rstd = _empty((M,), x), _empty((M,), x)
    s = torch.empty_like(x) if pos is not None else x
    _lib.call('pdae_add_layernorm_forward', x, M, C, _lib.ptr(x), _lib.ptr(pos), _lib.ptr(gamma),
              _lib.ptr(beta), float(eps), _lib.ptr(s) if pos is not None else None, _lib.ptr(y),
              _lib.ptr(mean), _lib.ptr(rstd))
    return s, y, mean, rstd


class PosGrad:
    """Gradient of a stack's position embedding.  The reference re-adds `pos` before EVERY block
    (PointCAE_transformer.py:174-177), so d pos = the sum over the blocks of each block's d(stream);
    the LayerNorm-backward kernel of each block adds its dx into one buffer (the last block of the
    stack, first in the backward pass, writes it; the first block returns it to autograd) instead
    of autograd running one elementwise add per block."""
    __slots__ = ('n', 'buf')

    def __init__(self, n):
        self.n, self.buf = n, None

    def mode(self, index, like):
        """-> (buffer or None, kernel mode, this block returns the buffer as its pos gradient)"""
        if self.n <= 1:
            return None, 0, False
        if index == self.n - 1:
            self.buf = torch.empty_like(like)
            return self.buf, 1, False
        return self.buf, 2, index == 0


def _ln_backward(dy, s, mean, rstd, gamma, dres, dacc=None, dacc_mode=0):
    """dy may be split-K slabs (S, M, C).  -> dx, dgamma, dbeta"""
    M, C = s.shape
    dx = torch.empty_like(s)
    gb, _ = arena.take(2 * C, s)
    dg, db = gb[:C], gb[C:]
    _lib.call('pdae_layernorm_backward', s, M, C, _lib.ptr(dy), _slabs(dy), _lib.ptr(s), _lib.ptr(mean),
              _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(dres), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), 1,
              _lib.ptr(dacc), dacc_mode)
    return dx, dg, db


def _res_ln_forward(a, bias, keep, res, pos, gamma, beta, eps, T):
    M, C = res.shape
    s, y = torch.empty_like(res), torch.empty_like(res)
    mean, rstd = _empty((M,), res), _empty((M,), res)
    _lib.call('pdae_residual_layernorm_forward', res, M, C, T, _lib.ptr(a), _slabs(a), _lib.ptr(bias),
              _lib.ptr(keep), _lib.ptr(res), _lib.ptr(pos), _lib.ptr(gamma), _lib.ptr(beta), float(eps),
              _lib.ptr(s), _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd))
    return s, y, mean, rstd


def _res_ln_backward(dy, s, mean, rstd, gamma, dres, keep, T, dacc=None, dacc_mode=0):
    """dy may be split-K slabs (S, M, C).  -> dx (w.r.t. the stream), da (w.r.t. the branch),
    dgamma, dbeta, dbias (column sums of da)"""
    M, C = s.shape
    dx = torch.empty_like(s)
    buf, _ = arena.take(3 * C, s)
    dg, db, dbias = buf[:C], buf[C:2 * C], buf[2 * C:]
    da = torch.empty_like(s) if keep is not None else dx
    _lib.call('pdae_residual_layernorm_backward', s, M, C, T, _lib.ptr(dy), _slabs(dy), _lib.ptr(s),
              _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(dres), _lib.ptr(keep), _lib.ptr(dx),
              _lib.ptr(da) if keep is not None else None, _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias), 1,
              _lib.ptr(dacc), dacc_mode)
    return dx, da, dg, db, dbias


class _ResidualLayerNorm(torch.autograd.Function):
    """(s, y) with s = res + keep[b] * (a + bias) (+ pos), y = LayerNorm(s): the tail of the
    previous sub-layer (Linear bias, DropPath, residual add) folded into the norm that
    consumes it, forward and backward (one launch each instead of two)."""

    @staticmethod
    def forward(ctx, a, bias, keep, res, pos, gamma, beta, eps, T):
        """`a` may be (S, M, C): split-K slabs of the branch, added up by the kernel."""
        a, res = a.contiguous(), res.contiguous()
        M, C = res.shape
        s, y, mean, rstd = _res_ln_forward(a, bias, keep, res, pos, gamma, beta, eps, T)
        ctx.save_for_backward(s, mean, rstd, gamma, keep)
        ctx.T, ctx.has_pos, ctx.has_bias, ctx.slabs = T, pos is not None, bias is not None, _slabs(a)
        ctx.mark_non_differentiable(mean, rstd)
        ctx.set_materialize_grads(False)
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        s, mean, rstd, gamma, keep = ctx.saved_tensors
        M, C = s.shape
        if dy is None:                        # only the sum was used: plain scale_residual backward
            ds = ds.contiguous()
            if keep is not None:
                da = torch.empty_like(ds)
                dbias = arena.take(C, ds)[0] if ctx.has_bias else None
                _lib.call('pdae_scale_colsum', ds, M, C, ctx.T, _lib.ptr(ds), _lib.ptr(keep), _lib.ptr(da),
                          _lib.ptr(dbias), 1)
            else:
                da, dbias = ds, (_colsum(ds) if ctx.has_bias else None)
            return _to_slabs(da, ctx.slabs), dbias, None, ds, (ds if ctx.has_pos else None), None, None, None, None
        dx, da, dg, db, dbias = _res_ln_backward(dy.contiguous(), s, mean, rstd, gamma, ds, keep, ctx.T)
        return (_to_slabs(da, ctx.slabs), (dbias if ctx.has_bias else None), None, dx, (dx if ctx.has_pos else None),
                dg, db, None, None)


class _ExpandToken(torch.autograd.Function):
    """token (1,1,C) -> (B, M, C) view; backward = column sums on the colsum kernel (ATen's strided
    reduction of the decoder's mask-token gradient took 46 us)."""

    @staticmethod
    def forward(ctx, token, B, M):
        return token.expand(B, M, -1)

    @staticmethod
    def backward(ctx, g):
        g2 = g.reshape(-1, g.shape[-1]).contiguous()
        return _colsum(g2).clone().reshape(1, 1, -1), None, None


def expand_token(token, B, M):
    if not token.is_cuda or token.shape[-1] % 4 != 0:
        return token.expand(B, M, -1)
    return _ExpandToken.apply(token, B, M)


class _AssembleTokens(torch.autograd.Function):
    """The decoder's input (PointCAE_transformer.py:700-703): per sample [the Tv visible tokens | M copies of the mask token],
    one launch (csrc/glue.hip assemble_tokens; was expand + cat); the backward splits the gradient into its two contiguous
    parts in one launch and column-sums the masked part (was two strided copies, the column sum and a clone)."""

    @staticmethod
    def forward(ctx, x_vis, token, B, Tv, M):
        C = x_vis.shape[-1]
        out = _empty((B, Tv + M, C), x_vis)
        _lib.call('pdae_assemble_tokens', x_vis, B, Tv + M, Tv, C, _lib.ptr(x_vis), _lib.ptr(token), _lib.ptr(out))
        ctx.dims = (B, Tv, M, C)
        return out

    @staticmethod
    def backward(ctx, g):
        B, Tv, M, C = ctx.dims
        g = g.contiguous()
        dvis, dmask = _empty((B, Tv, C), g), _empty((B * M, C), g)
        _lib.call('pdae_assemble_tokens_grad', g, B, Tv + M, Tv, C, _lib.ptr(g), _lib.ptr(dvis), _lib.ptr(dmask))
        return dvis, _colsum(dmask).reshape(1, 1, -1), None, None, None


ASSEMBLE = os.environ.get('PDAE_ASSEMBLE', os.environ.get('PDAE_GLUE', '1')) != '0'     # (A/B: 0 = torch.cat of the visible tokens and the expanded mask token)


def assemble_tokens(x_vis, token, B, Tv, M):
    """(B, Tv, C) visible tokens + the (1, 1, C) mask token -> (B, Tv + M, C)."""
    x_vis = x_vis.reshape(B, Tv, -1)
    if not ASSEMBLE or not x_vis.is_cuda or x_vis.shape[-1] % 4 != 0 or x_vis.dtype != torch.float32 or Tv == 0:
        return torch.cat([x_vis, expand_token(token, B, M)], dim=1)
    return _AssembleTokens.apply(x_vis.contiguous(), token.contiguous(), B, Tv, M)


class Pending:
    """A sub-layer's output that has not been added to the residual stream yet:
    stream = res + keep * (a + bias).  The norm that consumes it does the add."""
    __slots__ = ('a', 'bias', 'keep', 'res', 'T')

    def __init__(self, a, bias, keep, res, T):
        self.a, self.bias, self.keep, self.res, self.T = a, bias, keep, res, T

    def resolve(self):
        a = self.a.sum(0) if self.a.dim() == 3 else self.a      # split-K slabs of the branch
        return _ScaleResidual.apply(a, self.bias, self.keep, self.res, self.T)


def residual_layer_norm(x, pos, ln):
    """x: rows or a Pending -> (stream (+ pos), LN of it)."""
    if isinstance(x, Pending):
        return _ResidualLayerNorm.apply(x.a, x.bias, x.keep, x.res, pos, ln.weight, ln.bias, ln.eps, x.T)
    return _AddLayerNorm.apply(x, pos, ln.weight, ln.bias, ln.eps)


def add_layer_norm(x, pos, ln):
    """-> (x + pos, LN(x + pos)); the sum feeds the residual stream."""
    return _AddLayerNorm.apply(x, pos, ln.weight, ln.bias, ln.eps)


def layer_norm(x, ln):
    return residual_layer_norm(x, None, ln)[1]


class _Attention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, B, T, H, scale):
        qkv = qkv.contiguous()
        D = qkv.shape[1] // (3 * H)
        o = _empty((B * T, H * D), qkv)
        lse = _empty((B, H, T), qkv)
        _lib.call('pdae_attention_forward', qkv, B, T, H, D, float(scale), _lib.ptr(qkv), _lib.ptr(o),
                  _lib.ptr(lse))
        ctx.save_for_backward(qkv, o, lse)
        ctx.dims = (B, T, H, D, float(scale))
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, o, lse = ctx.saved_tensors
        B, T, H, D, scale = ctx.dims
        dqkv = torch.empty_like(qkv)
        _lib.call('pdae_attention_backward', qkv, B, T, H, D, scale, _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse),
                  _lib.ptr(do.contiguous()), _lib.ptr(dqkv))
        return dqkv, None, None, None, None


def attention_core(qkv, B, T, H, scale):
    """softmax(q k^T * scale) v per (sample, head) on qkv rows (B*T, 3*H*64)."""
    return _Attention.apply(qkv, B, T, H, scale)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z):
        z = z.contiguous()
        h = torch.empty_like(z)
        _lib.call('pdae_gelu_forward', z, z.numel(), _lib.ptr(z), _lib.ptr(h))
        ctx.save_for_backward(z)
        return h

    @staticmethod
    def backward(ctx, dh):
        (z,) = ctx.saved_tensors
        dz = torch.empty_like(z)
        _lib.call('pdae_gelu_backward', z, z.numel(), _lib.ptr(z), _lib.ptr(dh.contiguous()), _lib.ptr(dz))
        return dz


def gelu(z):
    return _Gelu.apply(z)


class _BiasGelu(torch.autograd.Function):
    """h = GELU(z + bias); backward also yields the bias gradient in the same pass."""

    @staticmethod
    def forward(ctx, z, bias):
        z = z.contiguous()
        M, C = z.shape
        h = torch.empty_like(z)
        _lib.call('pdae_bias_gelu_forward', z, M, C, _lib.ptr(z), _lib.ptr(bias), _lib.ptr(h))
        ctx.save_for_backward(z, bias)
        return h

    @staticmethod
    def backward(ctx, dh):
        z, bias = ctx.saved_tensors
        M, C = z.shape
        dz = torch.empty_like(z)
        db, _ = arena.take(C, z)
        _lib.call('pdae_bias_gelu_backward', z, M, C, _lib.ptr(z), _lib.ptr(bias), _lib.ptr(dh.contiguous()),
                  _lib.ptr(dz), _lib.ptr(db), 1)
        return dz, db


def bias_gelu(z, bias):
    return _BiasGelu.apply(z, bias)


class _ScaleResidual(torch.autograd.Function):
    """y = res + keep[b] * (a + bias): Linear bias + DropPath + residual add."""

    @staticmethod
    def forward(ctx, a, bias, keep, res, T):
        a = a.contiguous()
        M, C = a.shape
        y = torch.empty_like(a)
        _lib.call('pdae_scale_residual', a, M, C, T, _lib.ptr(a), _lib.ptr(bias), _lib.ptr(keep),
                  _lib.ptr(res.contiguous()), _lib.ptr(y))
        ctx.save_for_backward(keep)
        ctx.T = T
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        dy = dy.contiguous()
        M, C = dy.shape
        if keep is not None:
            da = torch.empty_like(dy)
            dbias = arena.take(C, dy)[0] if ctx.has_bias else None
            _lib.call('pdae_scale_colsum', dy, M, C, ctx.T, _lib.ptr(dy), _lib.ptr(keep), _lib.ptr(da),
                      _lib.ptr(dbias), 1)
        else:
            da = dy
            dbias = _colsum(da) if ctx.has_bias else None
        return da, dbias, None, dy, None


def drop_path_keep_buffer(drop_probs):
    """(2*depth, 1) keep probabilities, one row per stochastic-depth site."""
    return torch.tensor([1.0 - p for p in drop_probs for _ in range(2)], dtype=torch.float32).unsqueeze(1)


def draw_drop_path(B, drop_probs, training, keep):
    """timm 0.4.5 DropPath for a whole stack of blocks at once: every block has
    two stochastic-depth sites (after attention, after the MLP), each keeping a
    sample with probability 1-p and scaling it by 1/(1-p).  The reference draws
    torch.rand((B,1,1)) at each of the 2*depth sites; here all sites of a stack
    come from ONE torch.rand launch.  -> list of (keep_attn, keep_mlp) per block,
    None where p == 0.  `keep` is drop_path_keep_buffer(drop_probs) on the device."""
    if not training or all(p == 0. for p in drop_probs):
        return [(None, None)] * len(drop_probs)
    r = torch.rand((2 * len(drop_probs), B), dtype=keep.dtype, device=keep.device)
    if r.is_cuda:                 # floor(r + keep) / keep for every site in one launch (csrc/block.hip)
        _lib.call('pdae_drop_path_keep', r, r.shape[0], B, _lib.ptr(r), _lib.ptr(keep), _lib.ptr(r))
    else:
        r = (r + keep).floor_() / keep
    return [(None, None) if p == 0. else (r[2 * i], r[2 * i + 1]) for i, p in enumerate(drop_probs)]


PREDRAW = os.environ.get('PDAE_PREDRAW', os.environ.get('PDAE_GLUE', '1')) != '0'     # (A/B: 0 = one rand + one keep launch per stack)


def predraw_drop_path(B, stacks):
    """The stochastic-depth draws of several stacks of blocks (encoder + decoder) from ONE torch.rand and ONE keep launch: each
    stack's next stack_keeps() call takes its own rows.  Stacks: modules with .blocks, .dp_keep and .training."""
    live = [s for s in stacks if s.training and any(b.drop_prob != 0. for b in s.blocks)]
    if not PREDRAW or len(live) < 2 or not live[0].dp_keep.is_cuda:
        return
    owner, key = live[0].__dict__, tuple(id(s) for s in live)
    cached = owner.get('_dp_keep_all')
    if cached is None or cached[0] != key or cached[1].device != live[0].dp_keep.device:
        cached = owner['_dp_keep_all'] = (key, torch.cat([s.dp_keep for s in live], 0))
    keep = cached[1]
    r = torch.rand((keep.shape[0], B), dtype=keep.dtype, device=keep.device)
    _lib.call('pdae_drop_path_keep', r, r.shape[0], B, _lib.ptr(r), _lib.ptr(keep), _lib.ptr(r))
    o = 0
    for s in live:
        n = s.dp_keep.shape[0]
        s.__dict__['_keeps_next'] = (B, r[o:o + n])
        o += n


def stack_keeps(stack, B):
    """[(keep_attn, keep_mlp)] of one stack of blocks: its share of predraw_drop_path()'s launch, else its own draw."""
    probs = [blk.drop_prob for blk in stack.blocks]
    pre = stack.__dict__.pop('_keeps_next', None)
    if pre is not None and pre[0] == B and stack.training:
        r = pre[1]
        return [(None, None) if p == 0. else (r[2 * i], r[2 * i + 1]) for i, p in enumerate(probs)]
    return draw_drop_path(B, probs, stack.training, stack.dp_keep)


def rows_gemm(x, w, w_kn=False, bias=None, epi=0, z=None, may_split=False, big_cfg=None):
    """y = epi(x . op(w)) on the row-GEMM family (csrc/rows_gemm.hip, include/pdae.h).
    w_kn False: w is (N, K), torch's (out, in): a Linear's forward; True: w is (K, N): the same
    weight as the data-gradient operand.  epi 0 store (+bias) | 1 bias+ReLU | 2 GELU(z) -> y and
    GELU'(z) -> z | 3 y = acc * z.  may_split: the result may be (S, M, N) split-K slabs whose
    consumer adds them up (the LayerNorm kernels do)."""
    M, K = x.shape
    N = w.shape[1] if w_kn else w.shape[0]
    if M * max(K, N) >= 1 << 30:             # the kernels address an operand with 32-bit byte offsets: row chunks
        rows = ((1 << 30) // max(K, N) - 1) // 128 * 128
        y = _empty((M, N), x)
        for m0 in range(0, M, rows):
            m1 = min(M, m0 + rows)
            cfg, _, _ = _lib.rows_gemm_plan(m1 - m0, N, K, w_kn, False)
            if big_cfg is not None and cfg < 16 and BIG_TILES and m1 - m0 >= BIG_ROWS:
                cfg = big_cfg
            rows_c = m1 - m0
            probed_family('rows_gemm', 2.0 * rows_c * N * K,
                          lambda m0=m0, m1=m1, rows_c=rows_c, cfg=cfg: _lib.call(
                              'pdae_rows_gemm', x, rows_c, N, K, _lib.ptr(x[m0:m1]), _lib.ptr(w), int(w_kn), _lib.ptr(bias), epi,
                              _lib.ptr(z[m0:m1]) if z is not None else None, _lib.ptr(y[m0:m1]), cfg, 1, 0),
                          nbytes=4.0 * (rows_c * K + N * K + rows_c * N + (rows_c * N if z is not None else 0)))
        return y
    cfg, splits, sb = _lib.rows_gemm_plan(M, N, K, w_kn, may_split)
    if big_cfg is not None and cfg < 16 and M >= BIG_ROWS and splits == 1 and BIG_TILES:
        cfg = big_cfg         # a caller's measured fp32-input tile shape for a multi-millisecond product (that plan is calibrated
                              # on M <= 8192; the exact-split family's plan, cfg >= 16, prices rounds and stands)
    y = _empty((splits, M, N) if splits > 1 else (M, N), x)
    probed_family('rows_gemm', 2.0 * M * N * K,
                  lambda: _lib.call('pdae_rows_gemm', x, M, N, K, _lib.ptr(x), _lib.ptr(w), int(w_kn), _lib.ptr(bias),
                                    epi, _lib.ptr(z), _lib.ptr(y), cfg, splits, sb),
                  nbytes=4.0 * (M * K + N * K + y.numel() + (M * N if z is not None else 0)))
    return y


# Tile shapes for the FoldingNet stage's multi-millisecond products (tools/lab/rows_big.py, 524288 x 512 x 512: bias+ReLU
# forward 96x128 tiles 126.7 vs 121.9 TFLOP/s on 64x64; ReLU-masked data gradient 128x128 123.0 vs 115.7)
BIG_ROWS = 1 << 19                 # (the published variant's 190 k-row stages are faster on the planned 64x64 tiles: 17.55 vs 17.63 ms)
BIG_TILES = os.environ.get('PDAE_BIG_TILES', '1') != '0'


# Gradient sink.  FlatDataParallel tags every parameter it owns with (weakref to itself, index); a graphed step
# ARMS its own FlatDataParallel instance around its forward + backward (graph_step._phase1).  A Function whose
# weights all belong to one armed owner -- and still live in that owner's flat parameter buffer -- writes their
# gradients straight into the owner's flat gradient views, records the indices in owner.sink_written and returns
# None for them; the step's gather copy then skips those tensors (97 % of the 116 MB).  The state is per
# FlatDataParallel instance: a second model in the process, or an un-armed (eager) step, never sees it; a parameter
# moved out of the flat buffer (.to(), a re-allocation) no longer matches its slot and falls back to autograd.
def _sink_tags(weights):
    """forward-time: the (owner ref, index) tags of the parameters (saved tensors lose Python attributes)."""
    return [getattr(w, '_pdae_flat', None) for w in weights]


def _sink_views(tags, weights):
    """backward-time: (owner, indices, flat gradient views) when every weight sits in ONE armed owner, else None."""
    if not tags or any(t is None for t in tags):
        return None
    owner = tags[0][0]()
    if owner is None or not owner.sink_armed or any(t[0]() is not owner for t in tags):
        return None
    idx = [t[1] for t in tags]
    base, esz = owner.flat_param.data_ptr(), owner.flat_param.element_size()
    for i, w in zip(idx, weights):
        if w.data_ptr() != base + owner.offsets[i][0] * esz or owner.grad_views[i].shape != w.shape:
            return None
    return owner, idx, [owner.grad_views[i] for i in idx]


# lab switch (tools/lab/ab.sh): one grouped launch per block instead of one per stack (measured 0.3 ms slower)
WGRAD_PER_BLOCK = os.environ.get('PDAE_WGRAD_PER_BLOCK', '0') != '0'
# blocks per grouped weight-gradient launch (0 = a whole stack): a launch's operands are the activations / gradients of
# its blocks' backward passes, and a group small enough to still sit in the 256 MB Infinity Cache is read from there
WGRAD_GROUP = int(os.environ.get('PDAE_WGRAD_GROUP', '0'))


def flush_wgrad_queue(owner):
    """Issue the queued weight gradients of an armed FlatDataParallel (see _TransformerBlock.backward).
    With `owner.wgrad_stream` set (graph_step: PDAE_WGRAD_SIDE=1) the launch goes to that side stream behind an event
    on the current one -- a parallel branch of the captured graph: nothing downstream of the backward reads a weight
    gradient, so the stack's ~1 ms of dW tiles may run beside the rest of the backward (the other stack's / the
    embedder's chain of small launches); join_wgrad_stream() closes the branch before the gradients are gathered."""
    q = owner.wgrad_queue
    if q:
        flops = 2.0 * sum(dy.shape[0] * dy.shape[1] * x.shape[1] for dy, x, _, _ in q)
        side = getattr(owner, 'wgrad_stream', None)
        if side is not None:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _lib.rows_wgrad_multi(q)
            owner.wgrad_inflight.append(q)           # operands stay alive until the join
        else:
            probed_family('rows_wgrad', flops, lambda: _lib.rows_wgrad_multi(q),
                          nbytes=4.0 * sum(dy.shape[0] * (dy.shape[1] + x.shape[1]) + dy.shape[1] * x.shape[1] for dy, x, _, _ in q))
        owner.wgrad_queue = []


def join_wgrad_stream(owner):
    side = getattr(owner, 'wgrad_stream', None)
    if side is not None and owner.wgrad_inflight:
        torch.cuda.current_stream().wait_stream(side)
        owner.wgrad_inflight = []


def rows_wgrad(dys, xs, with_bias, outs=None, db_outs=None):
    """Weight (and bias) gradients of a group of Linear layers that share their rows, one grouped
    launch (+ the ordered slab reduction, complete when the call returns to the stream).  -> ([dW], [db or None]);
    outs / db_outs: preallocated outputs (db_outs: one per True in with_bias), e.g. the flat gradient views of an armed
    FlatDataParallel (_sink_views)."""
    M = dys[0].shape[0]
    Ns, Ks = [t.shape[1] for t in dys], [t.shape[1] for t in xs]
    ws = _empty((max(_lib.rows_wgrad_workspace(M, Ns, Ks), 1),), dys[0])
    dws = outs if outs is not None else [_empty((n, k), dys[0]) for n, k in zip(Ns, Ks)]
    it = iter(db_outs) if db_outs is not None else None
    dbs = [(next(it) if it is not None else _empty((n,), dys[0])) if f else None for n, f in zip(Ns, with_bias)]
    probed_family('rows_wgrad', 2.0 * M * sum(n * k for n, k in zip(Ns, Ks)),
                  lambda: _lib.rows_wgrad(dys[0], M, dys, xs, dws, dbs, ws),
                  nbytes=4.0 * sum(M * (n + k) + n * k for n, k in zip(Ns, Ks)))
    return dws, dbs


class _Linear(torch.autograd.Function):
    """y = act(x W^T + b) on rows; act None or 'relu' (the ReLU mask is recomputed from y)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x = x.contiguous()
        y = rows_gemm(x, w, False, b, 1 if relu else 0)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_bias, ctx.relu = b is not None, relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.relu:
            dy = dy * (y > 0)
        dx = rows_gemm(dy, w, True) if ctx.needs_input_grad[0] else None
        dws, dbs = rows_wgrad([dy], [x], [ctx.has_bias])
        return dx, dws[0], dbs[0], None


def _rows_gemm_few_rows(x, w, w_kn, bias, epi, z=None):
    """rows_gemm for a handful of rows against a long reduction: planned with up to 8 split-K slabs, which
    pdae_slab_sum_epi adds with the bias and the epilogue (one GEMM launch when the plan keeps one slab)."""
    M, K = x.shape
    N = w.shape[1] if w_kn else w.shape[0]
    cfg, splits, sb = _lib.rows_gemm_plan(M, N, K, w_kn, 8)
    if splits == 1:
        return rows_gemm(x, w, w_kn, bias, epi, z)
    slabs = _empty((splits, M, N), x)
    y = _empty((M, N), x)

    def both():                                     # the product is complete only behind the slab sum: one probed unit
        _lib.call('pdae_rows_gemm', x, M, N, K, _lib.ptr(x), _lib.ptr(w), int(w_kn), None, 0, None,
                  _lib.ptr(slabs), cfg, splits, sb)
        _lib.call('pdae_slab_sum_epi', x, splits, M, N, _lib.ptr(slabs), _lib.ptr(bias), epi, _lib.ptr(z), _lib.ptr(y))
    # bytes: operands + the slabs written by the GEMM, then the slabs read and the result written by the sum
    probed_family('rows_gemm', 2.0 * M * N * K, both, nbytes=4.0 * (M * K + N * K + 2 * slabs.numel() + M * N))
    return y


class _MLPChain(torch.autograd.Function):
    """Linear -> ReLU -> Linear -> ReLU -> ... -> Linear on rows as ONE node (the coarse heads: models/PointCAE_DGCNN.py
    recfc, PointCAE_pointnetv2.py folding1, PointCAE_transformer.py coarse_pred): forward = the layers' row GEMMs with
    bias (+ ReLU) in the epilogue; backward = each data gradient masked in ITS epilogue by the ReLU output it flows into
    (rows_gemm epi 4: no compare / multiply passes), and the weight + bias gradients of ALL layers as one grouped launch
    (they share their rows).  Inputs: x, then (w, b) per layer; every layer but the last is followed by a ReLU."""

    @staticmethod
    def forward(ctx, x, *params):
        ws, bs = params[0::2], params[1::2]
        acts = [x.contiguous()]
        for i, (w, b) in enumerate(zip(ws, bs)):
            acts.append(_rows_gemm_few_rows(acts[-1], w, False, b, 1 if i + 1 < len(ws) else 0))
        ctx.save_for_backward(*acts[:-1], *ws)
        ctx.n, ctx.has_bias = len(ws), [b is not None for b in bs]
        return acts[-1]

    @staticmethod
    def backward(ctx, dy):
        n = ctx.n
        acts, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        dys = [None] * n
        dys[n - 1] = dy.contiguous()
        for i in range(n - 1, 0, -1):                   # acts[i] = relu(layer i - 1): its sign masks the gradient
            dys[i - 1] = _rows_gemm_few_rows(dys[i], ws[i], True, None, 4, acts[i])
        dx = _rows_gemm_few_rows(dys[0], ws[0], True, None, 0) if ctx.needs_input_grad[0] else None
        dws, dbs = rows_wgrad(dys, list(acts), ctx.has_bias)
        out = [dx]
        for dw, db in zip(dws, dbs):
            out += [dw, db]
        return tuple(out)


def mlp_chain(x, layers):
    """layers: the nn.Linear modules of a Linear / ReLU / ... / Linear head, in order -> the last layer's output."""
    if not (x.dim() == 2 and x.is_cuda and x.dtype == torch.float32):
        raise RuntimeError('mlp_chain: rows must be a 2-D fp32 tensor on the GPU (there is no CPU / library path)')
    if any(l.weight.shape[0] % 4 or l.weight.shape[1] % 4 or l.bias is None for l in layers):
        for i, l in enumerate(layers):              # ragged widths: layer by layer (linear_any pads them)
            x = linear(x, l, 'relu' if i + 1 < len(layers) else None)
        return x
    params = []
    for l in layers:
        params += [l.weight, l.bias]
    return _MLPChain.apply(x, *params)


FOLD_SUMS = os.environ.get('PDAE_FOLD_SUMS', os.environ.get('PDAE_GLUE', '1')) != '0'     # (A/B: 0 = at::sum)


def _partials_sum(part):
    """part (P, ...) -> its sum over P in a fixed order (csrc/glue.hip partials_sum_t with one row: 16 strided lane sums per
    column, then the lanes; at::sum's reduce kernel took 11-24 us on these shapes)."""
    if not FOLD_SUMS or not part.is_cuda or part.dtype != torch.float32 or not part.is_contiguous() or part.shape[0] == 0:
        return part.sum(0)
    n = part[0].numel()
    out = _empty(tuple(part.shape[1:]), part)
    _lib.call('pdae_partials_sum_t', part, part.shape[0], 1, n, _lib.ptr(part), _lib.ptr(out))
    return out


class _FoldMLP(torch.autograd.Function):
    """The FoldingNet stage of Point_CAE_PointNetv2 (models/PointCAE_pointnetv2.py:157-167: folding2 =
    Conv1d(1029,512) ReLU Conv1d(512,512) ReLU Conv1d(512,3)) on rows, given the first conv's three
    hoisted terms a (clouds, C), p (clouds*coarse, C), gd (cells, C):
        h1 = relu((a + p) + gd)   one write pass (csrc/folding.hip)
        h2 = relu(h1 W2^T + b2)   bias + ReLU in the GEMM epilogue
        y  = h2 W3^T + b3         (W3 zero-padded to 4 outputs)
    Backward: each data-gradient GEMM masks its result with the ReLU output it flows into (rows_gemm
    epi 4) -- no compare / multiply / threshold passes over the 4.3 GB activations -- and one pass over
    the first layer's masked gradient yields dp and the partial sums of dgd."""

    @staticmethod
    def forward(ctx, a, p, gd, row, w2, b2, w3, b3, clouds, coarse, cells):
        p = p.contiguous()
        C = p.shape[1]
        rows = clouds * coarse * cells
        if row is None:
            a, gd = a.contiguous(), gd.contiguous()
            h1 = _empty((rows, C), p)
            _lib.call('pdae_fold_input', p, clouds, coarse, cells, C, _lib.ptr(a), _lib.ptr(p), _lib.ptr(gd), _lib.ptr(h1))
        else:      # a per-point term instead of the per-cloud / per-cell ones (the published variant's second stage)
            row = row.contiguous()
            h1 = _empty((rows, C), p)
            _lib.call('pdae_fold_input_rows', p, clouds * coarse, cells, C, _lib.ptr(row), _lib.ptr(p), _lib.ptr(h1))
        h2 = rows_gemm(h1, w2, False, b2, 1, big_cfg=5)
        y = rows_gemm(h2, w3, False, b3, 0)
        ctx.save_for_backward(h1, h2, w2, w3)
        ctx.dims = (clouds, coarse, cells, C)
        ctx.per_row = row is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        h1, h2, w2, w3 = ctx.saved_tensors
        clouds, coarse, cells, C = ctx.dims
        dy = dy.contiguous()
        if w3.shape[0] == 4:
            # the C -> 3(+1) layer backwards in one pass over h2 (csrc/folding.hip fold_out_backward)
            parts = _lib.lib().pdae_fold_out_backward_parts(dy.shape[0])
            d2, part = torch.empty_like(h2), _empty((parts, 4, C), dy)
            _lib.call('pdae_fold_out_backward', dy, dy.shape[0], C, _lib.ptr(dy), _lib.ptr(h2), _lib.ptr(w3.contiguous()),
                      _lib.ptr(d2), _lib.ptr(part))
            dw3, db3 = _partials_sum(part), dy.sum(0)
        else:
            d2 = rows_gemm(dy, w3, True, None, 4, h2)               # gradient of h2's pre-activation
            (dw3,), (db3,) = rows_wgrad([dy], [h2], [True])
        # (the large product of the stage: the [N,K] form of the kernel is 8 % faster than the [K,N] form at
        #  this size, and transposing the weight costs nothing)
        d1 = rows_gemm(d2, w2.t().contiguous(), False, None, 4, h1, big_cfg=0)  # gradient of h1's pre-activation
        (dw2,), (db2,) = rows_wgrad([d2], [h1], [True])
        del d2
        if ctx.per_row:        # dp = the pairs' sums over their cells (fold_input_grad's first output; the per-cell partials unused)
            pairs = clouds * coarse
            dp = _empty((pairs, C), dy)
            part = _empty((_lib.lib().pdae_fold_input_grad_parts(1, pairs), cells, C), dy)
            _lib.call('pdae_fold_input_grad', dy, 1, pairs, cells, C, _lib.ptr(d1), _lib.ptr(dp), _lib.ptr(part))
            return None, dp, None, d1, dw2, db2, dw3, db3, None, None, None
        parts = _lib.lib().pdae_fold_input_grad_parts(clouds, coarse)
        dp = _empty((clouds * coarse, C), dy)
        part = _empty((parts, cells, C), dy)
        _lib.call('pdae_fold_input_grad', dy, clouds, coarse, cells, C, _lib.ptr(d1), _lib.ptr(dp), _lib.ptr(part))
        da = dp.view(clouds, coarse, C).sum(1) if ctx.needs_input_grad[0] else None     # (a constant zero term has no gradient)
        return (da, dp, _partials_sum(part), None, dw2, db2, dw3, db3, None, None, None)


class _MaxPlusMean(torch.autograd.Function):
    """x (B, T, C) -> max over T + mean over T (B, C), one launch each way (csrc/glue.hip max_plus_mean)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, T, C = x.shape
        out, arg = _empty((B, C), x), _empty((B, C), x, torch.uint8)
        _lib.call('pdae_max_plus_mean', x, B, T, C, _lib.ptr(x), _lib.ptr(out), _lib.ptr(arg))
        ctx.save_for_backward(arg)
        ctx.T = T
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        g = g.contiguous()
        B, C = g.shape
        dx = _empty((B, ctx.T, C), g)
        _lib.call('pdae_max_plus_mean_grad', g, B, ctx.T, C, _lib.ptr(g), _lib.ptr(arg), _lib.ptr(dx))
        return dx


def max_plus_mean(x):
    """x.max(dim=1)[0] + x.mean(1) for x (B, T, C)."""
    if not PAD2D or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 3 or not 0 < x.shape[1] <= 255:
        return x.max(dim=1)[0] + x.mean(1)
    return _MaxPlusMean.apply(x)


class _SplitWeightCols(torch.autograd.Function):
    """w (R, C) -> its column blocks [b0, b1), ... as separate contiguous operands, each zero-padded to a multiple of 4 columns
    (one launch per block: pad2d reads the block through w's row stride); backward: the blocks' gradients side by side as dW
    in ONE launch (csrc/glue.hip hcat) -- autograd's own path is a zero fill + a copy per block and the adds between them."""

    @staticmethod
    def forward(ctx, w, *bounds):
        w = w.contiguous()
        R, C = w.shape
        outs = []
        for b0, b1 in zip(bounds[0::2], bounds[1::2]):
            n = b1 - b0
            o = _empty((R, n + (-n) % 4), w)
            _lib.call('pdae_pad2d', w, R, n, C, R, o.shape[1], w.data_ptr() + 4 * b0, _lib.ptr(o))
            outs.append(o)
        ctx.bounds, ctx.shape = bounds, (R, C)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        R, C = ctx.shape
        widths = [b1 - b0 for b0, b1 in zip(ctx.bounds[0::2], ctx.bounds[1::2])]
        gs = [g.contiguous() if g is not None else None for g in gs]
        like = next(g for g in gs if g is not None)
        srcs = [g if g is not None else torch.zeros((R, n + (-n) % 4), device=like.device) for g, n in zip(gs, widths)]
        k = len(srcs)
        dw = _empty((R, C), like)
        parr, iarr = ctypes.c_void_p * k, ctypes.c_int * k
        _lib.call('pdae_hcat', like, k, R, parr(*[_lib.ptr(t) for t in srcs]), iarr(*widths), iarr(*[t.shape[1] for t in srcs]),
                  _lib.ptr(dw))
        return (dw,) + (None,) * len(ctx.bounds)


def _hcat(pieces, R, like):
    """pieces: [(2-D tensor or None, cols)] -> (R, sum cols) with the pieces' leading columns side by side (None: zeros)."""
    k = len(pieces)
    out = _empty((R, sum(c for _, c in pieces)), like)
    parr, iarr = ctypes.c_void_p * k, ctypes.c_int * k
    _lib.call('pdae_hcat', like, k, R, parr(*[_lib.ptr(t) for t, _ in pieces]), iarr(*[c for _, c in pieces]),
              iarr(*[(t.stride(0) if t is not None else c) for t, c in pieces]), _lib.ptr(out))
    return out


class _InsertZeroCol(torch.autograd.Function):
    """w (R, C) -> (R, C + 1) with a zero column at `at` (the pad column of a set-abstraction level's first weight: the grouped
    rows are [xyz - centre | 0 | features]); one launch each way (was new_zeros + cat, and two slice gradients + their add)."""

    @staticmethod
    def forward(ctx, w, at):
        w = w.contiguous()
        R, C = w.shape
        ctx.at, ctx.shape = at, (R, C)
        pieces = [(w, at), (None, 1)] + ([(w[:, at:], C - at)] if C > at else [])
        return _hcat(pieces, R, w)

    @staticmethod
    def backward(ctx, g):
        R, C = ctx.shape
        g = g.contiguous()
        pieces = [(g, ctx.at)] + ([(g[:, ctx.at + 1:], C - ctx.at)] if C > ctx.at else [])
        return _hcat(pieces, R, g), None


def insert_zero_col(w, at):
    if not PAD2D or not w.is_cuda or w.dtype != torch.float32 or w.dim() != 2 or not 0 < at <= w.shape[1]:
        return torch.cat([w[:, :at], w.new_zeros(w.shape[0], 1), w[:, at:]], dim=1)
    return _InsertZeroCol.apply(w, at)


def split_weight_cols(w, bounds):
    """w (R, C), bounds = [(b0, b1), ...] covering 0..C in order, at most four -> the column blocks as contiguous (R, width padded
    to a multiple of 4) operands."""
    flat = [v for b in bounds for v in b]
    ok = (PAD2D and w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and 1 <= len(bounds) <= 4 and flat[0] == 0
          and flat[-1] == w.shape[1] and all(flat[2 * i + 1] == flat[2 * i + 2] for i in range(len(bounds) - 1))
          and all(b1 > b0 for b0, b1 in bounds))
    if not ok:
        return tuple(pad2d(w[:, b0:b1], 0, (-(b1 - b0)) % 4) for b0, b1 in bounds)
    return _SplitWeightCols.apply(w, *flat)


class _Pad2d(torch.autograd.Function):
    """x (R, C) [or (C,)] -> (R + pr, C + pc) with zeros, one launch (csrc/glue.hip pad2d; F.pad is a fill + a copy)."""

    @staticmethod
    def forward(ctx, x, pr, pc):
        one_d = x.dim() == 1
        x2 = x.reshape(1, -1) if one_d else x
        if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]):
            x2 = x2.contiguous()                        # (a column slice of a row-major matrix is read through its row stride)
        R, C = x2.shape
        ld = x2.stride(0) if R > 1 else C
        out = _empty((R + pr, C + pc), x2)
        _lib.call('pdae_pad2d', x2, R, C, ld, R + pr, C + pc, _lib.ptr(x2), _lib.ptr(out))
        ctx.dims = (R, C, one_d)
        return out.reshape(-1) if one_d else out

    @staticmethod
    def backward(ctx, g):
        R, C, one_d = ctx.dims
        return (g[:C] if one_d else g[:R, :C]), None, None


PAD2D = os.environ.get('PDAE_PAD2D', os.environ.get('PDAE_GLUE', '1')) != '0'


def pad2d(x, pr, pc):
    """zero rows below / zero columns right of a 1-D or 2-D fp32 device tensor (1-D: pc elements appended)."""
    if not PAD2D or not x.is_cuda or x.dtype != torch.float32 or x.dim() not in (1, 2):
        return F.pad(x, (0, pc, 0, pr)) if x.dim() == 2 else F.pad(x, (0, pc))
    return _Pad2d.apply(x, pr, pc)


def fold_mlp(a, p, gd, conv2, conv3, clouds, coarse, cells, row_term=None):
    """-> (clouds*coarse*cells, 3) offsets; conv2 / conv3 = the stage's second and third Conv1d (kernel 1).
    First-layer terms: a (clouds, C) per cloud, p (clouds*coarse, C) per coarse point / patch, gd (cells, C) per
    grid cell -- or, with row_term (rows, C), p plus a per-point term (a and gd unused)."""
    w3, b3 = conv3.weight.squeeze(-1), conv3.bias
    n = w3.shape[0]
    pn = (-n) % 4
    if pn:                                              # 3 output coordinates: a zero fourth row
        w3, b3 = pad2d(w3, pn, 0), pad2d(b3, 0, pn)
    y = _FoldMLP.apply(a, p, gd, row_term, conv2.weight.squeeze(-1), conv2.bias, w3, b3, clouds, coarse, cells)
    return y[:, :n] if pn else y


def _linear_rows(x, w, b, relu=False):
    return linear_any(x, w, b, relu)


def linear_any(x, w, b=None, relu=False):
    """x W^T (+ b) (+ ReLU) for any K, N on the row GEMMs: they reduce in multiples of 4 and write
    multiples of 4, so a ragged weight (K = 3 xyz columns, N = 3 output coordinates) is zero-padded
    (the weight is small; an activation is padded only when it is narrow -- wide ones should be built
    padded by the caller, as the set-abstraction grouping does)."""
    if not (x.dim() == 2 and x.is_cuda and x.dtype == torch.float32):
        raise RuntimeError('linear_any: rows must be a 2-D fp32 tensor on the GPU (there is no CPU / library path)')
    N, K = w.shape
    pk, pn = (-K) % 4, (-N) % 4
    if pk:
        if x.shape[1] == K:
            x = pad2d(x, 0, pk)
    if pk or pn:
        w = pad2d(w, pn, pk)                            # (both paddings of the weight in one launch)
    if pn:
        b = pad2d(b, 0, pn) if b is not None else None
    y = _Linear.apply(x, w, b, relu)
    return y[:, :N] if pn else y


def linear(x, lin, act=None):
    """nn.Linear on rows (bias, and a following ReLU, in the GEMM epilogue)."""
    y = _linear_rows(x, lin.weight, lin.bias, act == 'relu')
    if act == 'gelu':
        y = gelu(y)
    return y


def conv1x1(x_rows, conv):
    """nn.Conv1d(kernel 1) applied to (rows, Cin) -> (rows, Cout)."""
    return _linear_rows(x_rows, conv.weight.squeeze(-1), conv.bias)


class _PosEmbed(torch.autograd.Function):
    """Linear(3,128) -> GELU -> Linear(128,C) on centre rows (PointCAE_transformer.py:329-333).
    The first layer (K = 3) with its row gather, GELU and the factor GELU's backward needs is one small kernel
    (pdae_pos_embed_fc1: it also leaves the gathered rows zero-padded to 4 columns, the weight-gradient GEMM's
    operand); the second layer is a row GEMM.  Centres carry no gradient."""

    @staticmethod
    def forward(ctx, xyz, rows, w1, b1, w2, b2):
        M = rows.numel() if rows is not None else xyz.shape[0]
        H = w1.shape[0]
        h, gp, xp = _empty((M, H), xyz), _empty((M, H), xyz), _empty((M, 4), xyz)
        _lib.call('pdae_pos_embed_fc1', xyz, M, H, _lib.ptr(xyz), _lib.ptr(rows), _lib.ptr(w1.contiguous()), _lib.ptr(b1),
                  _lib.ptr(h), _lib.ptr(gp), _lib.ptr(xp))
        y = rows_gemm(h, w2, False, b2, 0)
        ctx.save_for_backward(xp, gp, h, w2)
        ctx.sink_tags, ctx.w1 = _sink_tags([w1]), w1
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, gp, h, w2 = ctx.saved_tensors
        dy = dy.contiguous()
        dz = rows_gemm(dy, w2, True, None, 3, gp)
        (dw2, dw1p), (db2, db1) = rows_wgrad([dy, dz], [h, xp], [True, True])
        sink = _sink_views(ctx.sink_tags, [ctx.w1])
        if sink:
            # graphed step: the (H, 3) gradient is the leading columns of the (H, 4) tile; returned as a view, autograd would
            # make it contiguous with a launch of its own -- the step's gather copy takes it as one more entry instead
            owner, idx, views = sink
            owner.sink_strided.append((views[0], dw1p, 3))
            owner.sink_written.update(idx)
            return None, None, None, db1, dw2, db2
        return None, None, dw1p[:, :3], db1, dw2, db2


def pos_embed(xyz, seq, rows=None):
    """Linear(3,128) -> GELU -> Linear(128,C) (PointCAE_transformer.py:329-333) of xyz (R,3) -- of its rows
    `rows` (int64 device indices) when given: the gather is part of the first layer's kernel."""
    if not xyz.is_cuda:
        raise RuntimeError('pos_embed: rows must be on the GPU (there is no CPU path)')
    if seq[0].weight.shape[0] % 4 != 0:
        raise NotImplementedError('pos_embed: hidden width must be a multiple of 4')
    return _PosEmbed.apply(xyz.contiguous(), rows, seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias)


class _TransformerBlock(torch.autograd.Function):
    """One pre-LN block on rows (Block.forward, PointCAE_transformer.py:155-158 with :174-177):

        stream = res + keep_in * (a_in + bias_in)   (the previous block's MLP branch; or stream = res)
        x1 = stream + pos;   x2 = x1 + keep1 * (proj(attn(qkv(LN1 x1))) + b_proj)
        -> (a2, x2) with the block's output = x2 + keep2 * (a2 + b_fc2), added by the NEXT norm.

    Forward: 4 GEMMs (fc1's epilogue yields GELU and GELU'), 2 LayerNorm launches that fold the
    bias / DropPath / residual adds, the attention core.  Backward: the data-gradient chain
    (dz straight out of the fc2-transposed GEMM's epilogue), then the four weight gradients +
    fc1's bias gradient as ONE grouped launch.  GEMMs with a long reduction and a narrow output
    (fc2, and the fc1 / qkv data gradients) may come back as split-K slabs; the LayerNorm kernel
    that consumes them adds the slabs."""

    @staticmethod
    def forward(ctx, a_in, bias_in, keep_in, res, pos, keep1, keep2, g1, b1, wqkv, wproj, bproj, g2, b2, w1,
                bf1, w2, B, T, H, scale, eps1, eps2, pos_grad, tail=0):
        res = res.contiguous()
        M, C = res.shape
        if a_in is not None:
            x1, n1, mean1, rstd1 = _res_ln_forward(a_in.contiguous(), bias_in, keep_in, res, pos, g1, b1, eps1, T)
        else:
            x1, n1, mean1, rstd1 = _add_ln_forward(res, pos, g1, b1, eps1)
        qkv = rows_gemm(n1, wqkv)
        D = wqkv.shape[0] // (3 * H)
        o = _empty((M, H * D), res)
        lse = _empty((B, H, T), res)
        _lib.call('pdae_attention_forward', qkv, B, T, H, D, float(scale), _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse))
        if tail:
            # only the last `tail` rows of every sample are read downstream (the decoder returns the
            # masked tokens, PointCAE_transformer.py:229-231): everything after the attention core is
            # row-wise, so it runs on those rows alone.  Same values on them, zero gradient elsewhere.
            Tt = tail
            o_t, x1_t = _empty((B * tail, H * D), res), _empty((B * tail, C), res)
            if H * D == C:                     # both slices in one launch (csrc/block.hip tail_rows_gather)
                _lib.call('pdae_tail_rows_gather', res, B, T, tail, C, _lib.ptr(o), _lib.ptr(x1), _lib.ptr(o_t), _lib.ptr(x1_t))
            else:
                _lib.call('pdae_tail_rows_gather', res, B, T, tail, H * D, _lib.ptr(o), None, _lib.ptr(o_t), None)
                _lib.call('pdae_tail_rows_gather', res, B, T, tail, C, _lib.ptr(x1), None, _lib.ptr(x1_t), None)
        else:
            Tt, o_t, x1_t = T, o, x1
        a1 = rows_gemm(o_t, wproj, may_split=True)
        x2, n2, mean2, rstd2 = _res_ln_forward(a1, bproj, keep1, x1_t, None, g2, b2, eps2, Tt)
        gp = _empty((x2.shape[0], w1.shape[0]), res)
        h = rows_gemm(n2, w1, False, bf1, 2, gp)
        a2 = rows_gemm(h, w2, may_split=True)
        ctx.save_for_backward(x1, n1, mean1, rstd1, qkv, o, lse, x2, n2, mean2, rstd2, gp, h, keep_in, keep1,
                              g1, wqkv, wproj, g2, w1, w2, bf1)
        ctx.dims = (B, T, H, D, float(scale))
        ctx.tail = tail
        ctx.has_in, ctx.has_pos, ctx.has_bias_in, ctx.in_slabs = a_in is not None, pos is not None, bias_in is not None, \
            (_slabs(a_in) if a_in is not None else 1)
        ctx.a2_slabs = _slabs(a2)
        ctx.pos_grad = pos_grad if pos is not None else None          # (PosGrad, index of this block in its stack)
        ctx.sink_tags = _sink_tags([wqkv, wproj, w1, w2, bf1])
        ctx.set_materialize_grads(False)
        return a2, x2

    @staticmethod
    def backward(ctx, da2, dx2):
        (x1, n1, mean1, rstd1, qkv, o, lse, x2, n2, mean2, rstd2, gp, h, keep_in, keep1,
         g1, wqkv, wproj, g2, w1, w2, bf1) = ctx.saved_tensors
        B, T, H, D, scale = ctx.dims
        da2 = _from_slabs(da2).contiguous()
        dz = rows_gemm(da2, w2, True, None, 3, gp)                        # (M, 4C): GELU' in the epilogue
        dn2 = rows_gemm(dz, w1, True, may_split=True)
        tail = ctx.tail
        dx1, da1, dg2, db2, dbproj = _res_ln_backward(dn2, x2, mean2, rstd2, g2,
                                                      dx2.contiguous() if dx2 is not None else None, keep1,
                                                      tail if tail else T)
        do = rows_gemm(da1, wproj, True)
        o_t = o
        if tail:                               # rows outside the tail: zero gradient from this block's second half
            C = x1.shape[1]
            o_t = _empty((B * tail, o.shape[1]), x1)
            _lib.call('pdae_tail_rows_gather', x1, B, T, tail, o.shape[1], _lib.ptr(o), None, _lib.ptr(o_t), None)
            full = _empty((2, B * T, C), x1)
            _lib.call('pdae_tail_rows_scatter', x1, B, T, tail, C, _lib.ptr(do), _lib.ptr(dx1), _lib.ptr(full[0]), _lib.ptr(full[1]))
            do, dx1 = full[0], full[1]
        dqkv = torch.empty_like(qkv)
        _lib.call('pdae_attention_backward', qkv, B, T, H, D, scale, _lib.ptr(qkv), _lib.ptr(o), _lib.ptr(lse),
                  _lib.ptr(do), _lib.ptr(dqkv))
        dn1 = rows_gemm(dqkv, wqkv, True, may_split=True)
        dacc, dmode, ret_acc = ctx.pos_grad[0].mode(ctx.pos_grad[1], x1) if ctx.pos_grad is not None else (None, 0, False)
        if ctx.has_in:
            dx0, da0, dg1, db1, dbias_in = _res_ln_backward(dn1, x1, mean1, rstd1, g1, dx1, keep_in, T, dacc, dmode)
            da0 = _to_slabs(da0, ctx.in_slabs)
        else:
            dx0, dg1, db1 = _ln_backward(dn1, x1, mean1, rstd1, g1, dx1, dacc, dmode)
            da0 = dbias_in = None
        # d pos: this block's dx0, or -- summed inside the kernels -- the stack's buffer from its first block
        dpos = None if not ctx.has_pos else (dx0 if dmode == 0 else (dacc if ret_acc else None))
        sink = _sink_views(ctx.sink_tags, [wqkv, wproj, w1, w2, bf1])   # graphed step: straight into the flat buffer
        if sink:
            # The weight gradients are not needed before the optimiser: the block only QUEUES them (operands kept
            # alive by the queue); the first block of the stack -- the last one backward reaches -- issues the whole
            # stack's weight gradients as ONE grouped launch (pdae_rows_wgrad_multi: every layer with its own row
            # count, (tile, 32-row chunk) units of all layers dealt in equal ranges to one residency of the chip).
            # 16 launches of 90-250 us with their ramps, tails and partial-tile traffic become two of ~1 ms.
            owner, _, views = sink
            owner.wgrad_queue += [(dqkv, n1, views[0], None), (da1, o_t, views[1], None), (dz, n2, views[2], views[4]),
                                  (da2, h, views[3], None)]
            owner.sink_written.update(sink[1])
            first_of_stack = ctx.pos_grad is None or ctx.pos_grad[1] == 0
            if first_of_stack or WGRAD_PER_BLOCK or (WGRAD_GROUP and len(owner.wgrad_queue) >= 4 * WGRAD_GROUP):
                flush_wgrad_queue(owner)
            dwqkv = dwproj = dw1 = dw2 = dbf1 = None
        elif tail:                             # two row counts: two groups
            (dwqkv,), _ = rows_wgrad([dqkv], [n1], [False])
            (dwproj, dw1, dw2), (_, dbf1, _) = rows_wgrad([da1, dz, da2], [o_t, n2, h], [False, True, False])
        else:
            (dwqkv, dwproj, dw1, dw2), (_, _, dbf1, _) = rows_wgrad([dqkv, da1, dz, da2], [n1, o, n2, h],
                                                                    [False, False, True, False])
        return (da0, dbias_in if ctx.has_bias_in else None, None, dx0, dpos, None, None,
                dg1, db1, dwqkv, dwproj, dbproj, dg2, db2, dw1, dbf1, dw2, None, None, None, None, None, None, None, None)


def transformer_block(x, pos, B, T, blk, keeps, pending=False, pos_grad=None, tail=0):
    """block(x + pos): x = x + dp(attn(ln1(x))); x = x + dp(mlp(ln2(x)))
    (PointCAE_transformer.py:155-158, :174-177) as one autograd Function.  `keeps` =
    (keep_attn, keep_mlp) per-sample DropPath factors from draw_drop_path.  tail > 0: the result
    holds only the last `tail` rows of every sample (B*tail rows; see _TransformerBlock.forward).
    x may be a Pending (the previous block's MLP branch, added here by norm1's
    kernel); with pending=True the result is one too (for the next norm).  pos_grad =
    (PosGrad of the stack, index of this block): d pos is summed inside the kernels."""
    attn, mlp = blk.attn, blk.mlp
    keep1, keep2 = keeps
    if isinstance(x, Pending):
        a_in, bias_in, keep_in, res = x.a, x.bias, x.keep, x.res
    else:
        a_in, bias_in, keep_in, res = None, None, None, x
    a2, x2 = _TransformerBlock.apply(a_in, bias_in, keep_in, res, pos, keep1, keep2, blk.norm1.weight, blk.norm1.bias,
                                     attn.qkv.weight, attn.proj.weight, attn.proj.bias, blk.norm2.weight,
                                     blk.norm2.bias, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, B, T,
                                     attn.num_heads, attn.scale, blk.norm1.eps, blk.norm2.eps, pos_grad, tail)
    out = Pending(a2, mlp.fc2.bias, keep2, x2, tail if tail else T)
    return out if pending else out.resolve()
