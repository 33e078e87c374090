"""Shared MLP + max-pool of a PointNet++ set-abstraction level as one autograd Function.

Reference: extensions/pointnet2/pointnet2_modules.py (PointnetSAModule.forward: SharedMLP of
Conv2d(1x1, bias=False) -> BatchNorm2d -> ReLU layers on (B, C, npoint, nsample), then
F.max_pool2d over nsample), used by models/pointnetv2_util.py:319-346 for Point_CAE_PointNetv2.

On rows = (cloud, centre, sample) a 1x1 conv is a GEMM and the activations of a level are
1-2 M rows x 64-256 channels (0.5-1 GB each), so every pass over one costs as much as the GEMM that
made it.  Layout of the fused version (csrc/gemm.hip conv_stats, csrc/set_abstraction.hip,
csrc/embed.hip), the patch embedder's recipe:
  forward   per layer ONE kernel: the previous layer's BatchNorm + ReLU is applied while the
            operand is staged, this layer's batch statistics come out of the epilogue, only the
            raw conv output y_l is stored; bn_finalize turns the sums into scale / shift and
            updates the running estimates; the last layer's BatchNorm + ReLU + max over nsample
            is one read of y_3.
  backward  max-pool scatter (dense), then per layer: ReLU + BatchNorm backward in two sweeps
            (bnrelu_backward: the sums that are dgamma / dbeta, then the in-place apply), the
            weight gradient with the normalised input recomputed in the GEMM's producer, the
            data gradient on the row GEMMs.
ATen ran ~7 passes per layer forward (stats, transform, clamp) and ~6 backward.
"""
import torch

from . import _lib
from .patch_embed import _bn_finalize, _empty, _gemm, _gemm_bnstats, _wgrad, _wgrad_listed


# Parity-test hook (None in production): called as ARG_HOOK(arg) with the (groups, C) uint8 winners of a level's
# max-pool right after the forward computed them; what it returns is what the backward routes the gradient through.
# The reference's own winners can be injected this way, so that a gradient comparison is not at the mercy of near-ties
# that 1 ulp of GEMM rounding resolves the other way (tests/test_gpu_model.py).
ARG_HOOK = None


class SharedMLPMaxFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ns, bns, *params):
        """x (R, K0) rows, R = groups * ns; params = (w, gamma, beta) per layer, w (C_l, C_{l-1})."""
        x = x.contiguous()
        R = x.shape[0]
        if R % ns != 0 or R % 32 != 0:
            raise RuntimeError('shared MLP: rows must be groups * nsample and a multiple of 32')
        ws = [params[i].contiguous() for i in range(0, len(params), 3)]
        gammas = [params[i] for i in range(1, len(params), 3)]
        ys, affs = [], []
        inp, sc, sh = x, None, None
        for w, bn in zip(ws, bns):
            N, K = w.shape
            y = _empty((R, N), x)
            stats = _empty((8, 2, N), x)
            _lib.call('pdae_conv_stats', x, R, N, K, _lib.ptr(inp), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(w),
                      _lib.ptr(y), _lib.ptr(stats))
            sc, sh, mean, invstd = _bn_finalize(bn, R, x, partials=stats)
            ys.append(y)
            affs.append((sc, sh, mean, invstd))
            inp = y
        G, C = R // ns, ws[-1].shape[0]
        out = _empty((G, C), x)
        arg = _empty((G, C), x, torch.uint8)
        _lib.call('pdae_bnrelu_group_max', x, G, ns, C, _lib.ptr(inp), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(out),
                  _lib.ptr(arg))
        if ARG_HOOK is not None:
            arg = ARG_HOOK(arg).contiguous()
        ctx.save_for_backward(x, arg, out, *ws, *gammas, *ys, *[t for a in affs for t in a])
        ctx.ns, ctx.nl = ns, len(ws)
        return out

    @staticmethod
    def backward(ctx, dout):
        t = ctx.saved_tensors
        nl, ns = ctx.nl, ctx.ns
        x, arg, out = t[0], t[1], t[2]
        t = t[1:]
        ws, gammas, ys = t[2:2 + nl], t[2 + nl:2 + 2 * nl], t[2 + 2 * nl:2 + 3 * nl]
        flat = t[2 + 3 * nl:]
        affs = [flat[4 * i:4 * i + 4] for i in range(nl)]
        R = x.shape[0]
        G, C = R // ns, ws[-1].shape[0]
        grads = [None] * (3 * nl)
        dx = None
        d = _empty((R, C), x)
        dout = dout.contiguous()
        fused_pool = 256 % (C // 4) == 0 and C <= 1024
        if not fused_pool:
            _lib.call('pdae_group_max_scatter_n', x, G, ns, C, _lib.ptr(dout), _lib.ptr(arg), _lib.ptr(d))
        S_pre = None                     # BatchNorm l's sums when the GEMM that produced d left them (rows_gemm_bnrelu_stats)
        for l in range(nl - 1, -1, -1):
            sc, sh, mean, invstd = affs[l]
            N = ws[l].shape[0]
            S = _empty((2, N), x) if S_pre is None else S_pre
            if S_pre is not None:
                _lib.call('pdae_bnrelu_backward_apply', x, R // 32, N, _lib.ptr(d), _lib.ptr(ys[l]), _lib.ptr(sc),
                          _lib.ptr(sh), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(gammas[l]), _lib.ptr(S), None,
                          R // 32, None, None, None)
            elif l == nl - 1 and fused_pool:
                # straight through the max-pool: the gradient is non-zero only at the arg-max rows
                wsp = _empty((max(_lib.lib().pdae_pool_bn_backward_workspace(G, C), 1),), x)
                _lib.call('pdae_pool_bn_backward', x, G, ns, C, _lib.ptr(dout), _lib.ptr(arg), _lib.ptr(out),
                          _lib.ptr(ys[l]), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(gammas[l]), _lib.ptr(S),
                          _lib.ptr(wsp), _lib.ptr(d))
            else:
                # d <- gradient of the raw conv output y_l (ReLU mask, BatchNorm backward), in place
                _lib.call('pdae_bnrelu_backward', x, R // 32, N, _lib.ptr(d), _lib.ptr(ys[l]), _lib.ptr(sc),
                          _lib.ptr(sh), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(gammas[l]), _lib.ptr(S), None,
                          R // 32, None, None, None)
            grads[3 * l + 1], grads[3 * l + 2] = S[1], S[0]                   # dgamma, dbeta
            if l > 0:
                psc, psh = affs[l - 1][0], affs[l - 1][1]
                K = ws[l].shape[1]
                # BatchNorm + ReLU of the previous layer recomputed while its raw output is staged (patch_embed._wgrad_listed)
                grads[3 * l] = _wgrad_listed(R, d, None, ys[l - 1], None, psc, psh)[0]
                # gradient of relu(bn(y_{l-1})), ReLU-masked, + that BatchNorm's sums out of the same launch
                d, S_pre = _gemm_bnstats(d, ws[l], ys[l - 1], None, psc, psh, affs[l - 1][2], affs[l - 1][3])
            else:
                grads[0] = _wgrad(d, x)
                if ctx.needs_input_grad[0]:
                    dx = _gemm(d, ws[0], True)
        return (dx, None, None) + tuple(grads)


def shared_mlp_max(x, layers, ns, pad_at=None):
    """x (groups*ns, K0) -> (groups, C_last).  layers: the level's _ConvBN modules (conv without bias,
    BatchNorm2d, ReLU); pad_at: x carries a zero column there, the first weight gets the matching one."""
    params, bns = [], []
    for i, layer in enumerate(layers):
        w = layer.conv.weight.reshape(layer.conv.weight.shape[0], -1)
        if i == 0 and pad_at is not None:
            from .nn_ops import insert_zero_col
            w = insert_zero_col(w, pad_at)
        bn = layer.bn.bn
        params += [w, bn.weight, bn.bias]
        bns.append(bn)
    return SharedMLPMaxFunction.apply(x, ns, bns, *params)
