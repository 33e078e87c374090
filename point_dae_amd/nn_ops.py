"""Dense layers of the pretraining step on flat (rows, channels) activations.

The model code (point_cae_transformer.py, point_cae_pointnetv2.py) only calls
the functions below.  Plain GEMMs (nn.Linear without a fusable neighbour) go to
the BLAS library (hipBLASLt through torch.mm / addmm); everything around them --
the fused patch embedder (patch_embed.py), the attention core, LayerNorm with
the position / residual adds, GELU, bias + DropPath + residual -- runs on the
hand-written gfx950 kernels of csrc/{gemm,embed,attention,block}.hip through
autograd Functions whose backward calls the matching backward kernels.
Parameters are read from the reference-layout nn.Modules that own them.
"""
import torch
import torch.nn.functional as F

from . import _lib
from .arena import ZeroArena, arena, begin_step  # noqa: F401
from .patch_embed import patch_embed  # noqa: F401  (fused gfx950 embedder)
from .probe import Probe, set_probe  # noqa: F401


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
        M, C = x.shape
        y = torch.empty_like(x)
        mean, rstd = _empty((M,), x), _empty((M,), x)
        if pos is not None:
            pos = pos.contiguous()
            s = torch.empty_like(x)
        else:
            s = x
        _lib.call('pdae_add_layernorm_forward', x, M, C, _lib.ptr(x), _lib.ptr(pos), _lib.ptr(gamma),
                  _lib.ptr(beta), float(eps), _lib.ptr(s) if pos is not None else None, _lib.ptr(y),
                  _lib.ptr(mean), _lib.ptr(rstd))
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
        dx = torch.empty_like(s)
        gb, _ = arena.take(2 * C, s)
        dg, db = gb[:C], gb[C:]
        dres = ds.contiguous() if ds is not None else None
        _lib.call('pdae_layernorm_backward', s, M, C, _lib.ptr(dy.contiguous()), _lib.ptr(s), _lib.ptr(mean),
                  _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(dres), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), 1)
        return dx, (dx if ctx.has_pos else None), dg, db, None


class _ResidualLayerNorm(torch.autograd.Function):
    """(s, y) with s = res + keep[b] * (a + bias) (+ pos), y = LayerNorm(s): the tail of the
    previous sub-layer (Linear bias, DropPath, residual add) folded into the norm that
    consumes it, forward and backward (one launch each instead of two)."""

    @staticmethod
    def forward(ctx, a, bias, keep, res, pos, gamma, beta, eps, T):
        a, res = a.contiguous(), res.contiguous()
        M, C = a.shape
        s, y = torch.empty_like(a), torch.empty_like(a)
        mean, rstd = _empty((M,), a), _empty((M,), a)
        if pos is not None:
            pos = pos.contiguous()
        _lib.call('pdae_residual_layernorm_forward', a, M, C, T, _lib.ptr(a), _lib.ptr(bias), _lib.ptr(keep),
                  _lib.ptr(res), _lib.ptr(pos), _lib.ptr(gamma), _lib.ptr(beta), float(eps), _lib.ptr(s),
                  _lib.ptr(y), _lib.ptr(mean), _lib.ptr(rstd))
        ctx.save_for_backward(s, mean, rstd, gamma, keep)
        ctx.T, ctx.has_pos, ctx.has_bias = T, pos is not None, bias is not None
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
            return da, dbias, None, ds, (ds if ctx.has_pos else None), None, None, None, None
        dx = torch.empty_like(s)
        buf, _ = arena.take(3 * C, s)
        dg, db, dbias = buf[:C], buf[C:2 * C], buf[2 * C:]
        da = torch.empty_like(s) if keep is not None else dx
        dres = ds.contiguous() if ds is not None else None
        _lib.call('pdae_residual_layernorm_backward', s, M, C, ctx.T, _lib.ptr(dy.contiguous()), _lib.ptr(s),
                  _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(dres), _lib.ptr(keep), _lib.ptr(dx),
                  _lib.ptr(da) if keep is not None else None, _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias), 1)
        return da, (dbias if ctx.has_bias else None), None, dx, (dx if ctx.has_pos else None), dg, db, None, None


class Pending:
    """A sub-layer's output that has not been added to the residual stream yet:
    stream = res + keep * (a + bias).  The norm that consumes it does the add."""
    __slots__ = ('a', 'bias', 'keep', 'res', 'T')

    def __init__(self, a, bias, keep, res, T):
        self.a, self.bias, self.keep, self.res, self.T = a, bias, keep, res, T

    def resolve(self):
        return _ScaleResidual.apply(self.a, self.bias, self.keep, self.res, self.T)


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
    r = (torch.rand((2 * len(drop_probs), B), dtype=keep.dtype, device=keep.device) + keep).floor_() / keep
    return [(None, None) if p == 0. else (r[2 * i], r[2 * i + 1]) for i, p in enumerate(drop_probs)]


class _Linear(torch.autograd.Function):
    """y = x W^T + b on rows: the three GEMMs go to the BLAS library, the bias
    gradient to the column-sum kernel (autograd's own `grad.sum(0)` is a
    single-wave-per-column reduction: 20-55 us on the (B*G, C) activations here)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.mm(dy, w) if ctx.needs_input_grad[0] else None
        dw = torch.mm(dy.t(), x) if ctx.needs_input_grad[1] else None
        db = _colsum(dy) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


def _linear_rows(x, w, b):
    if x.dim() != 2 or not x.is_cuda or b is None or b.numel() % 4 != 0:
        return F.linear(x, w, b)
    return _Linear.apply(x, w, b)


def linear(x, lin, act=None):
    """Plain nn.Linear on rows -> BLAS library GEMM (bias in its epilogue, or fused
    with the GELU that follows)."""
    if act == 'gelu' and x.dim() == 2 and x.is_cuda and lin.bias is not None and lin.bias.numel() % 4 == 0:
        return bias_gelu(torch.mm(x, lin.weight.t()), lin.bias)
    y = _linear_rows(x, lin.weight, lin.bias)
    if act == 'gelu':
        y = gelu(y)
    elif act == 'relu':
        y = F.relu(y)
    return y


def conv1x1(x_rows, conv):
    """nn.Conv1d(kernel 1) applied to (rows, Cin) -> (rows, Cout)."""
    return _linear_rows(x_rows, conv.weight.squeeze(-1), conv.bias)


def pos_embed(xyz_rows, seq):
    """Linear(3,128) -> GELU -> Linear(128,C) (PointCAE_transformer.py:329-333)."""
    return linear(linear(xyz_rows, seq[0], 'gelu'), seq[2])


def transformer_block(x, pos, B, T, blk, keeps, pending=False):
    """block(x + pos): x = x + dp(attn(ln1(x))); x = x + dp(mlp(ln2(x)))
    (PointCAE_transformer.py:155-158, :174-177) in 9 launches.  `keeps` =
    (keep_attn, keep_mlp) per-sample DropPath factors from draw_drop_path.
    x may be a Pending (the previous block's MLP branch, added here by norm1's
    kernel); with pending=True the result is one too (for the next norm)."""
    attn = blk.attn
    x1, n1 = residual_layer_norm(x, pos, blk.norm1)
    qkv = F.linear(n1, attn.qkv.weight, attn.qkv.bias)
    o = attention_core(qkv, B, T, attn.num_heads, attn.scale)
    keep1, keep2 = keeps
    x2, n2 = residual_layer_norm(Pending(torch.mm(o, attn.proj.weight.t()), attn.proj.bias, keep1, x1, T), None,
                                 blk.norm2)
    h = bias_gelu(torch.mm(n2, blk.mlp.fc1.weight.t()), blk.mlp.fc1.bias)
    out = Pending(torch.mm(h, blk.mlp.fc2.weight.t()), blk.mlp.fc2.bias, keep2, x2, T)
    return out if pending else out.resolve()
