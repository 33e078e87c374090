"""DGCNN denoising auto-encoder `Point_CAE_DGCNN_FCOnly` -- the non-Transformer model the released
checkpoints were trained with (rerun.sh:37-40) -- MI355X host side.

Reference: models/PointCAE_DGCNN.py:145-231; encoder models/dgcnn_util.py:87-136 with the
feature-space kNN and edge features of :7-34.  Parameter names equal the reference's (bn1..bn5 are
registered directly AND inside conv1..conv5, so its state_dict carries both key sets; so does this).

    model(corrupted_pts, pts) -> (loss_coarse, zeros(1));  model(.., pts, return_feat=True) -> (B,1024)

Data path on MI355X (csrc/dgcnn.hip has the derivations), activations as rows (points) x channels:
  * the graph is rebuilt before every EdgeConv in the space of that layer's input features: the Gram
    matrices X_b X_b^T of all clouds are ONE batched launch of the row GEMM (pdae_rows_gemm_batched),
    pdae_gram_topk selects the 20 largest of the reference's -|xi|^2 + 2 xi.xj - |xj|^2 per row;
  * an EdgeConv is conv([x_j - x_i, x_i]) = W1 x_j + (W2 - W1) x_i = p[j] + q[i]: two products PER POINT
    (one row GEMM on the stacked weight [W1; W2 - W1]) instead of one per edge -- 20x fewer FLOPs
    and no (B,2C,N,20) tensor; BatchNorm statistics, the winning edge (max or min of e by the sign of
    gamma: lrelu(bn(.)) is monotone) and sum_j p[j] in ONE pass over the gathered rows
    (pdae_edge_gather_stats), BatchNorm + LeakyReLU on the winners per point (pdae_bn_lrelu_rows);
    backward as a gather over the reverse graph (pdae_knn_reverse, pdae_edge_backward): no atomics,
    no per-edge tensor in either direction;
  * conv5 on the row GEMM, its BatchNorm1d + LeakyReLU + max over the points as pdae_cloud_pool_stats /
    pdae_cloud_pool_backward; recfc on the row GEMMs, Chamfer on the gfx950 kernel.
The whole encoder is ONE autograd node (_Encoder): forward and backward are explicit kernel sequences.
"""
import torch
import torch.nn as nn

from . import _lib, nn_ops
from .chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
from .corrupt_util_tensor import IN_FORWARD, corrupt_in_forward
from .patch_embed import _bn_finalize
from .registry import MODELS

K_GRAPH = 20


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


def feature_knn(x_rows, B, N, k=K_GRAPH, out=None, xyz=False, pd_out=None):
    """x_rows (B*N, C), C % 4 == 0 -> (B, N, k) int32 neighbour ids within the cloud (dgcnn_util.knn :7-12):
    Gram matrices on the batched row GEMM, the reference's distance expression and its top-k on one wave per row.
    out: a (B, N, k) int32 tensor to fill (the encoder keeps its four graphs in one allocation).  xyz: x_rows are the
    points themselves (3 coordinates, a zero 4th column): no Gram matrix (pdae_xyz_topk); pd_out (tests): (B, N, N) that
    receives the -pd values that kernel selected from."""
    x = x_rows.detach()
    C = x.shape[1]
    xx = _empty((B * N,), x)
    _lib.call('pdae_rows_sqnorm', x, B * N, C, _lib.ptr(x), _lib.ptr(xx))
    idx = _empty((B, N, k), x, torch.int32) if out is None else out
    if xyz:
        # the first layer's features are the points (3 coordinates + a zero column): distances straight from the rows
        if C != 4:
            raise RuntimeError('feature_knn(xyz=True): rows of 3 coordinates padded to 4 columns')
        _lib.call('pdae_xyz_topk', x, B, N, k, _lib.ptr(x), _lib.ptr(xx), _lib.ptr(idx), _lib.ptr(pd_out))
        return idx
    gram = _empty((B, N, N), x)
    _lib.call('pdae_rows_gemm_batched', x, B, N, N, C, _lib.ptr(x), N * C, _lib.ptr(x), N * C, _lib.ptr(gram), N * N)
    _lib.call('pdae_gram_topk', x, B, N, k, _lib.ptr(gram), _lib.ptr(xx), _lib.ptr(idx))
    return idx


def _parts(like, width):
    return _empty((_lib.lib().pdae_edge_parts(), 2 * width), like, torch.float64), _empty((2 * width,), like, torch.float64)


def _eval_affine(bn):
    """scale / shift / mean / invstd of an eval-mode BatchNorm (running estimates)."""
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    scale = bn.weight * invstd
    return scale, bn.bias - bn.running_mean * scale, bn.running_mean, invstd


class _Encoder(torch.autograd.Function):
    """dgcnn_encoder.forward (dgcnn_util.py:117-136) on rows.  Inputs: pts (B*N, 3) xyz rows; per EdgeConv the Conv2d
    weight (Co, 2 Cin, 1, 1) = [W1 | W2], gamma, beta; conv5's weight (1024, 512, 1), gamma, beta.  The BatchNorm
    modules ride along for their running estimates (updated in place in training mode)."""

    @staticmethod
    def forward(ctx, pts, B, N, training, bns, *params):
        R = B * N
        convs, gammas, betas = params[0:15:3], params[1:15:3], params[2:15:3]
        k = min(K_GRAPH, N)
        cin = pts.shape[1]
        x = _empty((R, cin + (-cin) % 4), pts)
        _lib.call('pdae_rows_pad', pts, R, cin, x.shape[1], _lib.ptr(pts.contiguous()), _lib.ptr(x))
        cat = _empty((R, sum(w.shape[0] for w in convs[:4])), pts)
        saved, off = [], 0
        graphs = _empty((4, B, N, k), pts, torch.int32)            # the four layers' graphs: ONE reverse-graph launch backward
        # the stacked weights [W1; W2 - W1] of the four layers (K padded like the layer's input rows) in one launch
        cos = [convs[li].shape[0] for li in range(4)]
        cins = [cin] + cos[:3]
        kps = [x.shape[1]] + cos[:3]
        stacked = [_empty((2 * co, kp), x) for co, kp in zip(cos, kps)]
        _lib.edge_weights_multi('pdae_edge_weight_stack_multi', x, cos, cins, kps, [c.contiguous() for c in convs[:4]], stacked)
        for li in range(4):
            gamma, bn = gammas[li], bns[li]
            co, kp = convs[li].shape[0], x.shape[1]
            assert kp == kps[li]
            w = stacked[li]
            idx = feature_knn(x, B, N, k, out=graphs[li], xyz=(li == 0 and cin == 3 and N <= 6144))
            pq = nn_ops.rows_gemm(x, w)
            esel, psum = _empty((R, co), x), _empty((R, co), x)
            sel = _empty((R, co), x, torch.int16)
            part, sums = _parts(x, co)
            _lib.call('pdae_edge_gather_stats', x, B, N, k, co, _lib.ptr(pq), _lib.ptr(idx), _lib.ptr(gamma), _lib.ptr(esel),
                      _lib.ptr(sel), _lib.ptr(psum), _lib.ptr(part), _lib.ptr(sums))
            if training:
                scale, shift, mean, invstd = _bn_finalize(bn, R * k, x, stats64=sums)
            else:
                scale, shift, mean, invstd = (t.contiguous() for t in _eval_affine(bn))
            out = _empty((R, co), x)
            _lib.call('pdae_bn_lrelu_rows', x, R, co, _lib.ptr(esel), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(out),
                      cat.data_ptr() + 4 * off, cat.shape[1])
            saved.append((x, w, idx, pq, esel, sel, psum, scale, shift, mean, invstd, cin))
            x, off, cin = out, off + co, co
        w5, g5, bn5 = convs[4].flatten(1).contiguous(), gammas[4], bns[4]
        C5 = w5.shape[0]
        y5 = nn_ops.rows_gemm(cat, w5)
        ysel, arow = _empty((B, C5), x), _empty((B, C5), x, torch.int32)
        rs = _lib.lib().pdae_cloud_pool_splits(B, N)
        pv, pr = _empty((B, rs, C5), x), _empty((B, rs, C5), x, torch.int32)
        part, sums = _empty((B * rs, 2 * C5), x, torch.float64), _empty((2 * C5,), x, torch.float64)
        _lib.call('pdae_cloud_pool_stats', x, B, N, C5, _lib.ptr(y5), _lib.ptr(g5), _lib.ptr(ysel), _lib.ptr(arow),
                  _lib.ptr(pv), _lib.ptr(pr), _lib.ptr(part), _lib.ptr(sums))
        if training:
            sc5, sh5, mean5, is5 = _bn_finalize(bn5, R, x, stats64=sums)
        else:
            sc5, sh5, mean5, is5 = (t.contiguous() for t in _eval_affine(bn5))
        feat = _empty((B, C5), x)
        _lib.call('pdae_bn_lrelu_rows', x, B, C5, _lib.ptr(ysel), _lib.ptr(sc5), _lib.ptr(sh5), _lib.ptr(feat), None, 0)
        ctx.layers, ctx.top, ctx.graphs = saved, (cat, w5, y5, ysel, arow, sc5, sh5, mean5, is5), graphs
        ctx.dims = (B, N, k)
        ctx.training = bool(training)
        ctx.shapes = [c.shape for c in convs]
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        B, N, k = ctx.dims
        R = B * N
        cat, w5, y5, ysel, arow, sc5, sh5, mean5, is5 = ctx.top
        C5 = w5.shape[0]
        dfeat = dfeat.contiguous()
        grads = [None] * 15
        g5 = _empty((B, C5), dfeat)
        part, sums = _parts(dfeat, C5)
        dgamma, dbeta = _empty((C5,), dfeat), _empty((C5,), dfeat)
        _lib.call('pdae_bn_lrelu_backward_reduce', dfeat, B, C5, _lib.ptr(dfeat), None, 0, _lib.ptr(ysel), _lib.ptr(sc5),
                  _lib.ptr(sh5), _lib.ptr(mean5), _lib.ptr(is5), _lib.ptr(g5), _lib.ptr(part), _lib.ptr(sums),
                  _lib.ptr(dgamma), _lib.ptr(dbeta))
        if not ctx.training:
            # eval-mode BatchNorm (running estimates): y = scale x + shift has no batch-statistic terms, so the c1 / c2
            # corrections of the backward kernels (fed by `sums`) vanish: d x = scale dy at the winners
            sums.zero_()
        dy5 = _empty((R, C5), dfeat)
        _lib.call('pdae_cloud_pool_backward', dfeat, B, N, C5, _lib.ptr(y5), _lib.ptr(g5), _lib.ptr(arow), _lib.ptr(sc5),
                  _lib.ptr(mean5), _lib.ptr(is5), _lib.ptr(sums), _lib.ptr(dy5))
        dcat = nn_ops.rows_gemm(dy5, w5, True)
        grads[12], grads[13], grads[14] = nn_ops.rows_wgrad([dy5], [cat], [False])[0][0].view(ctx.shapes[4]), dgamma, dbeta
        del dy5
        dx, off = None, cat.shape[1]
        # the reverse graphs (for every point the points that list it) of all four layers in one launch of 4 B blocks
        rev_start, rev_src = _empty((4, B, N + 1), dfeat, torch.int32), _empty((4, B, N * k), dfeat, torch.int32)
        _lib.call('pdae_knn_reverse', dfeat, 4 * B, N, k, _lib.ptr(ctx.graphs), _lib.ptr(rev_start), _lib.ptr(rev_src))
        queue = []
        for li in (3, 2, 1, 0):
            x, w, idx, pq, esel, sel, psum, scale, shift, mean, invstd, cin = ctx.layers[li]
            co = w.shape[0] // 2
            off -= co
            g = _empty((R, co), dfeat)
            part, sums = _parts(dfeat, co)
            dgamma, dbeta = _empty((co,), dfeat), _empty((co,), dfeat)
            _lib.call('pdae_bn_lrelu_backward_reduce', dfeat, R, co, _lib.ptr(dx), dcat.data_ptr() + 4 * off, dcat.shape[1],
                      _lib.ptr(esel), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(mean), _lib.ptr(invstd), _lib.ptr(g),
                      _lib.ptr(part), _lib.ptr(sums), _lib.ptr(dgamma), _lib.ptr(dbeta))
            if not ctx.training:
                sums.zero_()
            dpq = _empty((R, 2 * co), dfeat)
            _lib.call('pdae_edge_backward', dfeat, B, N, k, co, _lib.ptr(g), _lib.ptr(pq), _lib.ptr(sel), _lib.ptr(psum),
                      _lib.ptr(rev_start[li]), _lib.ptr(rev_src[li]), _lib.ptr(scale), _lib.ptr(mean), _lib.ptr(invstd),
                      _lib.ptr(sums), _lib.ptr(dpq))
            queue.append((li, dpq, x, co, cin))
            grads[3 * li + 1], grads[3 * li + 2] = dgamma, dbeta
            dx = nn_ops.rows_gemm(dpq, w, True) if li > 0 else None
        # the four stacked-weight gradients [dW1; d(W2 - W1)] = dpq^T x share their rows: ONE grouped launch (+ one ordered
        # reduction) instead of four of each
        dws = nn_ops.rows_wgrad([q[1] for q in queue], [q[2] for q in queue], [False] * len(queue))[0]
        dconvs = [_empty(ctx.shapes[q[0]], dfeat) for q in queue]
        _lib.edge_weights_multi('pdae_edge_weight_unstack_multi', dfeat, [q[3] for q in queue], [q[4] for q in queue],
                                [q[2].shape[1] for q in queue], dws, dconvs)
        for q, dconv in zip(queue, dconvs):
            grads[3 * q[0]] = dconv
        return (None, None, None, None, None) + tuple(grads)


class dgcnn_encoder(nn.Module):
    def __init__(self, channel=3):
        super().__init__()
        self.bn1, self.bn2 = nn.BatchNorm2d(64), nn.BatchNorm2d(64)
        self.bn3, self.bn4, self.bn5 = nn.BatchNorm2d(128), nn.BatchNorm2d(256), nn.BatchNorm1d(1024)
        act = lambda: nn.LeakyReLU(negative_slope=0.2)
        self.conv1 = nn.Sequential(nn.Conv2d(channel * 2, 64, kernel_size=1, bias=False), self.bn1, act())
        self.conv2 = nn.Sequential(nn.Conv2d(64 * 2, 64, kernel_size=1, bias=False), self.bn2, act())
        self.conv3 = nn.Sequential(nn.Conv2d(64 * 2, 128, kernel_size=1, bias=False), self.bn3, act())
        self.conv4 = nn.Sequential(nn.Conv2d(128 * 2, 256, kernel_size=1, bias=False), self.bn4, act())
        self.conv5 = nn.Sequential(nn.Conv1d(256 * 2, 1024, kernel_size=1, bias=False), self.bn5, act())

    def forward(self, x):
        """x (B,3,N) as the reference -> (B,1024)."""
        return self.forward_rows(x.transpose(1, 2))

    def forward_rows(self, pts):
        """pts (B,N,3) point rows (what the auto-encoder holds before the reference transposes them) -> (B,1024)."""
        B, N, C = pts.shape
        params, bns = [], []
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5):
            params += [conv[0].weight, conv[1].weight, conv[1].bias]
            bns.append(conv[1])
        return _Encoder.apply(pts.reshape(B * N, C), B, N, self.training, bns, *params)


@MODELS.register_module()
class Point_CAE_DGCNN_FCOnly(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.corrupt_type = config.corrupt_type
        self.num_coarse = 1024
        self.dgcnn_encoder = dgcnn_encoder(channel=3)
        self.recfc = nn.Sequential(nn.Linear(1024, 1024), nn.ReLU(), nn.Linear(1024, 1024), nn.ReLU(),
                                   nn.Linear(1024, self.num_coarse * 3))
        self.loss = config.loss
        self.build_loss_func(self.loss)

    @property
    def draws_in_forward(self):
        """True when forward() draws random numbers on the host (the in-forward dropouts): such a step must not be
        captured into a hipGraph (graph_step.GraphedStaticStep refuses it)."""
        return any(item in IN_FORWARD for item in self.corrupt_type)

    def build_loss_func(self, loss_type):
        if loss_type == 'cdl1':
            self.loss_func = ChamferDistanceL1()
        elif loss_type == 'cdl2':
            self.loss_func = ChamferDistanceL2()
        else:
            raise NotImplementedError(loss_type)

    def forward(self, corrupted_pts, pts, vis=False, return_feat=False, capture=None, **kwargs):
        nn_ops.begin_step(pts.device)
        if return_feat:
            return self.dgcnn_encoder.forward_rows(pts[:, :, :3])
        corrupted_pts, pts = corrupted_pts[:, :, :3].contiguous(), pts[:, :, :3].contiguous()
        corrupted_pts = corrupt_in_forward(corrupted_pts, self.corrupt_type)      # (:198-221: the CUDA-side dropouts)
        feature = self.dgcnn_encoder.forward_rows(corrupted_pts)
        r = self.recfc
        coarse = nn_ops.mlp_chain(feature, [r[0], r[2], r[4]])
        coarse = coarse.view(-1, self.num_coarse, 3)
        if capture is not None:
            capture.update(feature=feature, coarse=coarse)
        loss = self.loss_func(coarse, pts)
        return loss, torch.zeros(1, device=loss.device)
