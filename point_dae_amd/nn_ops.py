"""Dense layers of the pretraining step on flat (rows, channels) activations.

The model code (point_cae_transformer.py, point_cae_pointnetv2.py) only calls
the functions below, so each can move from a PyTorch-ROCm library call to a
hand-written gfx950 kernel without touching the models.  Parameters are read
from the reference-layout nn.Modules that own them.
"""
import torch
import torch.nn.functional as F


class Probe:
    """HIP-event timer for ONE named kernel launch site, used by bench.py's
    roofline: events are recorded on the stream the kernel is launched on,
    around every launch inside the timed region, and read after it."""

    def __init__(self):
        self.name, self.flops, self.events = None, 0.0, []

    def record(self, name, flops, fn):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn()
        e.record()
        self.name, self.flops = name, flops
        self.events.append((s, e))
        return out

    def summary(self):
        if not self.events:
            return None
        ms = [s.elapsed_time(e) for s, e in self.events]
        return {'name': self.name, 'flops': self.flops, 'avg_ms': sum(ms) / len(ms), 'launches': len(ms)}


_probe = None


def set_probe(p):
    global _probe
    _probe = p


def _probed(name, flops, fn):
    return _probe.record(name, flops, fn) if _probe is not None else fn()


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


def patch_embed(points, first_conv, second_conv, training):
    """mini-PointNet of Encoder.forward (PointCAE_transformer.py:37-51) on
    points (BG, n, 3) -> (BG, C).

    The 512->512 layer consumes concat([global.expand(n), local]); its weight
    is split so the global half is multiplied once per GROUP instead of once
    per point (BG x 256 x 512 instead of BG*n x 256 x 512): the same sum,
    associated differently.
    """
    BG, n, _ = points.shape
    rows = points.reshape(BG * n, 3)
    f = conv1x1(rows, first_conv[0])
    f = F.relu(first_conv[1](f))                    # BatchNorm1d on (rows, C): batch statistics
    f = conv1x1(f, first_conv[3])                   # (BG*n, 256)
    g = f.reshape(BG, n, -1).max(dim=1)[0]          # (BG, 256)
    w = second_conv[0].weight.squeeze(-1)           # (512, 512) = [global | local]
    cg = g.shape[1]
    h = F.linear(f, w[:, cg:]).reshape(BG, n, -1) + F.linear(g, w[:, :cg], second_conv[0].bias).unsqueeze(1)
    h = F.relu(second_conv[1](h.reshape(BG * n, -1)))
    # the largest GEMM of the step (M = BG*n rows, K = 512, N = C): bench.py's roofline kernel
    cout = second_conv[3].weight.shape[0]
    h = _probed('patch_embed.second_conv.3 fwd GEMM %dx%dx%d' % (h.shape[0], h.shape[1], cout),
                2.0 * h.shape[0] * h.shape[1] * cout, lambda: conv1x1(h, second_conv[3]))
    return h.reshape(BG, n, -1).max(dim=1)[0]
