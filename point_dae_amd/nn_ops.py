"""Dense layers of the pretraining step on flat (rows, channels) activations.

The model code (point_cae_transformer.py, point_cae_pointnetv2.py) only calls
the functions below, so each can move from a PyTorch-ROCm library call to a
hand-written gfx950 kernel without touching the models.  Parameters are read
from the reference-layout nn.Modules that own them.
"""
import torch
import torch.nn.functional as F


from .patch_embed import patch_embed  # noqa: F401  (fused gfx950 embedder)
from .probe import Probe, set_probe  # noqa: F401


def linear(x, lin, act=None):
    y = F.linear(x, lin.weight, lin.bias)
    if act == 'gelu':
        y = F.gelu(y)
    elif act == 'relu':
        y = F.relu(y)
    return y


def conv1x1(x_rows, conv):
    """nn.Conv1d(kernel 1) applied to (rows, Cin) -> (rows, Cout)."""
    return F.linear(x_rows, conv.weight.squeeze(-1), conv.bias)


def layer_norm(x, ln):
    return F.layer_norm(x, (x.shape[-1],), ln.weight, ln.bias, ln.eps)


def pos_embed(xyz_rows, seq):
    """Linear(3,128) -> GELU -> Linear(128,C) (PointCAE_transformer.py:329-333)."""
    return linear(linear(xyz_rows, seq[0], 'gelu'), seq[2])


def drop_path(x, B, drop_prob, training):
    """timm 0.4.5 DropPath: per-sample keep mask, x / keep * floor(keep + U)."""
    if drop_prob == 0. or not training:
        return x
    keep = 1 - drop_prob
    r = keep + torch.rand((B, 1, 1), dtype=x.dtype, device=x.device)
    r.floor_()
    rows = x.shape[0] // B
    return (x.reshape(B, rows, -1).div(keep) * r).reshape(x.shape)


def attention(x, B, T, attn):
    """softmax(q k^T * scale) v with 6 heads of 64 (Attention.forward :125-137);
    x: (B*T, C) rows."""
    C = x.shape[-1]
    H = attn.num_heads
    qkv = F.linear(x, attn.qkv.weight, attn.qkv.bias).reshape(B, T, 3, H, C // H).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = ((q @ k.transpose(-2, -1)) * attn.scale).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B * T, C)
    return F.linear(o, attn.proj.weight, attn.proj.bias)


def transformer_block(x, pos, B, T, blk, training):
    """block(x + pos): x = x + dp(attn(ln1(x))); x = x + dp(mlp(ln2(x)))."""
    x = x + pos
    x = x + drop_path(attention(layer_norm(x, blk.norm1), B, T, blk.attn), B, blk.drop_prob, training)
    h = linear(layer_norm(x, blk.norm2), blk.mlp.fc1, 'gelu')
    return x + drop_path(linear(h, blk.mlp.fc2), B, blk.drop_prob, training)
