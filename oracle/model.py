"""Plain-PyTorch fp32 restatement of the reference's pretraining models, on CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Geometry ops come from the
C oracle (oracle/ops.py); everything else is ordinary torch.nn so that it can
be checked directly against the reference's own Python classes imported in the
dev container (tests/golden/make_fixtures.py) -- parameter names equal the
reference's state_dict keys, so one state_dict drives both.  It is also the
model half of bench.py's cpu_baseline.

Restates: models/PointCAE_transformer.py (Encoder :20-51, Group :54-86,
Mlp :94-110, Attention :113-137, Block :140-158, TransformerEncoder :161-177,
TransformerDecoder :200-232, MaskTransformer :304-469, PointCAE_transformer
:616-742), datasets/corrupt_util_tensor.py (:59-348, :706-727) and
models/PointCAE_pointnetv2.py :61-173 + models/pointnetv2_util.py :319-346.
"""
import math
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops as O


def _np(t):
    return t.detach().cpu().numpy()


# ---------------------------------------------------------------- native ops
class _Chamfer(torch.autograd.Function):
    """extensions/chamfer_dist/__init__.py:14-26"""

    @staticmethod
    def forward(ctx, xyz1, xyz2):
        d1, d2, i1, i2 = (torch.from_numpy(a) for a in O.chamfer_forward(_np(xyz1), _np(xyz2)))
        ctx.save_for_backward(xyz1, xyz2, i1, i2)
        ctx.mark_non_differentiable(i1, i2)
        return d1, d2, i1, i2

    @staticmethod
    def backward(ctx, g1, g2, *_):
        xyz1, xyz2, i1, i2 = ctx.saved_tensors
        a, b = O.chamfer_backward(_np(xyz1), _np(xyz2), _np(i1), _np(i2), _np(g1), _np(g2))
        return torch.from_numpy(a), torch.from_numpy(b)


def chamfer_l2(a, b):
    d1, d2, _, _ = _Chamfer.apply(a.contiguous(), b.contiguous())
    return d1.mean() + d2.mean()


def chamfer_l1(a, b):
    d1, d2, _, _ = _Chamfer.apply(a.contiguous(), b.contiguous())
    return (d1.sqrt().mean() + d2.sqrt().mean()) / 2


class _Grouping(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.N = features.shape[2]
        return torch.from_numpy(O.grouping_operation(_np(features), _np(idx)))

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return torch.from_numpy(O.grouping_operation_grad(_np(g.contiguous()), _np(idx), ctx.N)), None


def group_divider(xyz, num_group, group_size):
    """Group.forward (PointCAE_transformer.py:61-86): FPS centres, kNN, gather,
    subtract centre."""
    x = _np(xyz)
    _, center = O.furthest_point_sample(x, num_group, return_centres=True)
    _, _, nbr = O.knn(x, center, group_size, return_nbr=True)
    return torch.from_numpy(nbr), torch.from_numpy(center)


# ------------------------------------------------------ in-forward corruption
AFFINE = ['translate', 'scale_nonorm', 'rotate', 'reflection', 'shear']


def _eye(B):
    return torch.eye(3).expand((B, 3, 3)).clone().float()


def draw_affine_step(name, B):
    """One level-4 affine corruption as ('mul', (B,3)) or ('mat', (B,3,3)); the
    host RNG calls are the reference's, in its order
    (corrupt_util_tensor.py:59-116, :139-191, :258-292, :307-342)."""
    if name == 'scale_nonorm':
        s = 2.0
        return 'mul', torch.FloatTensor(B, 1, 1, 3).uniform_(1. / s, s).reshape(B, 3)
    if name == 'translate':        # multiplies (sic), :88,113
        s = 0.5
        return 'mul', torch.FloatTensor(B, 1, 1, 3).uniform_(-s, s).reshape(B, 3)
    if name == 'rotate':
        clip = math.pi / 5 * (4 + 1)
        ang = torch.FloatTensor(B, 3).uniform_(-clip, clip)
        Rx, Ry, Rz = _eye(B), _eye(B), _eye(B)
        Rx[:, 1, 1] = torch.cos(ang[:, 0]); Rx[:, 1, 2] = -torch.sin(ang[:, 0])
        Rx[:, 2, 1] = torch.sin(ang[:, 0]); Rx[:, 2, 2] = torch.cos(ang[:, 0])
        Ry[:, 0, 0] = torch.cos(ang[:, 1]); Ry[:, 0, 2] = torch.sin(ang[:, 1])
        Ry[:, 2, 0] = -torch.sin(ang[:, 1]); Ry[:, 2, 2] = torch.cos(ang[:, 1])
        Rz[:, 0, 0] = torch.cos(ang[:, 2]); Rz[:, 0, 1] = -torch.sin(ang[:, 2])
        Rz[:, 1, 0] = torch.sin(ang[:, 2]); Rz[:, 1, 1] = torch.cos(ang[:, 2])
        return 'mat', torch.matmul(Rz, torch.matmul(Ry, Rx))
    if name == 'reflection':
        r = torch.from_numpy(np.random.choice(np.array([1, -1]), size=(B, 3)))
        Rx, Ry, Rz = _eye(B), _eye(B), _eye(B)
        Rx[:, 0, 0] = r[:, 0]
        Ry[:, 1, 1] = r[:, 1]
        Rz[:, 0, 0] = r[:, 2]      # z-flip lands on x (sic), :280
        return 'mat', torch.matmul(Rz, torch.matmul(Ry, Rx))
    if name == 'shear':
        sh = torch.from_numpy(np.random.uniform(-0.5, 0.5, size=(B, 6)))
        R = _eye(B)
        R[:, 0, 1] = sh[:, 0]; R[:, 0, 2] = sh[:, 1]; R[:, 1, 0] = sh[:, 2]
        R[:, 1, 2] = sh[:, 3]; R[:, 2, 0] = sh[:, 4]; R[:, 2, 1] = sh[:, 5]
        return 'mat', R
    raise KeyError(name)


def draw_corruption(corrupt_type, B):
    """corrupt_data (corrupt_util_tensor.py:706-727) up to the RNG draws: the
    list of affine steps for this batch."""
    steps = []
    for item in corrupt_type:
        if item in ('clean', 'Drop-Patch'):
            continue
        if item == 'affine_r3':
            number = random.choice([1, 2, 3])
            for name in random.sample(AFFINE, number):
                steps.append(draw_affine_step(name, B))
        else:
            raise NotImplementedError(item)
    return steps


def apply_corruption(neighborhood, center, steps):
    for kind, p in steps:
        if kind == 'mul':
            neighborhood = neighborhood * p[:, None, None, :]
            center = center * p[:, None, :]
        else:
            neighborhood = torch.matmul(neighborhood, p[:, None])
            center = torch.matmul(center, p)
    return neighborhood, center


def draw_mask(B, G, mask_ratio, rand_ratio):
    """MaskTransformer._mask_center_rand (PointCAE_transformer.py:395-422):
    one ratio per batch, int(ratio*G) masked per sample, numpy shuffle."""
    if rand_ratio == 'True':
        mask_ratio = torch.FloatTensor(1).uniform_(0.5, 0.8).item()
    num_mask = int(mask_ratio * G)
    overall = np.zeros([B, G])
    for i in range(B):
        m = np.hstack([np.zeros(G - num_mask), np.ones(num_mask)])
        np.random.shuffle(m)
        overall[i, :] = m
    return torch.from_numpy(overall).to(torch.bool)


# ------------------------------------------------------------------- layers
class DropPath(nn.Module):
    """timm 0.4.5 drop_path (per-sample stochastic depth)."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0. or not self.training:
            return x
        keep = 1 - self.p
        r = keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
        r.floor_()
        return x.div(keep) * r


class Encoder(nn.Module):
    def __init__(self, encoder_channel):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                        nn.Conv1d(128, 256, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, encoder_channel, 1))

    def forward(self, point_groups):
        bs, g, n, _ = point_groups.shape
        f = self.first_conv(point_groups.reshape(bs * g, n, 3).transpose(2, 1))
        fg = f.max(dim=2, keepdim=True)[0]
        f = self.second_conv(torch.cat([fg.expand(-1, -1, n), f], dim=1))
        return f.max(dim=2)[0].reshape(bs, g, self.encoder_channel)


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * self.scale).softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class Block(nn.Module):
    def __init__(self, dim, num_heads, drop_path):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * 4.))
        self.attn = Attention(dim, num_heads)

    def forward(self, x):
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class TransformerEncoder(nn.Module):
    def __init__(self, embed_dim, depth, num_heads, drop_path_rate):
        super().__init__()
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, drop_path_rate[i]) for i in range(depth)])

    def forward(self, x, pos):
        for blk in self.blocks:
            x = blk(x + pos)
        return x


class TransformerDecoder(nn.Module):
    def __init__(self, embed_dim, depth, num_heads, drop_path_rate):
        super().__init__()
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, drop_path_rate[i]) for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Identity()
        for m in self.modules():      # :216-223
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def forward(self, x, pos, return_token_num=-1):
        for blk in self.blocks:
            x = blk(x + pos)
        if return_token_num == -1:
            return self.norm(x)
        return self.norm(x[:, -return_token_num:])


def _pos_embed(dim):
    return nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, dim))


class MaskTransformer(nn.Module):
    def __init__(self, config):
        super().__init__()
        tc = config.transformer_config
        self.mask_ratio, self.rand_ratio = tc.mask_ratio, tc.rand_ratio
        self.trans_dim, self.depth = tc.trans_dim, tc.depth
        self.encoder = Encoder(tc.encoder_dims)
        self.pos_embed = _pos_embed(self.trans_dim)
        dpr = [x.item() for x in torch.linspace(0, tc.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(self.trans_dim, self.depth, tc.num_heads, dpr)
        self.norm = nn.LayerNorm(self.trans_dim)
        for m in self.modules():      # :353-364
            if isinstance(m, (nn.Linear, nn.Conv1d)):
                nn.init.trunc_normal_(m.weight, std=.02, a=-2., b=2.)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)

    def forward(self, neighborhood, center, mask=None):
        tokens = self.encoder(neighborhood)
        B, G, C = tokens.shape
        if mask is None:
            mask = draw_mask(B, G, self.mask_ratio, self.rand_ratio)
        x_vis = tokens[~mask].reshape(B, -1, C)
        pos = self.pos_embed(center[~mask].reshape(B, -1, 3))
        return self.norm(self.blocks(x_vis, pos)), mask


class PointCAE_transformer(nn.Module):
    """models/PointCAE_transformer.py:616-742: the 'Drop-Patch' branch with all_patch 'False' or
    'True', and the branch without masking (:717-739; its NormalTransformer :473-541 is a
    MaskTransformer that keeps every token and draws nothing)."""

    def __init__(self, config):
        super().__init__()
        tc = config.transformer_config
        self.trans_dim = tc.trans_dim
        self.masked = 'Drop-Patch' in config.corrupt_type
        self.MAE_encoder = MaskTransformer(config)
        self.group_size, self.num_group = config.group_size, config.num_group
        self.corrupt_type, self.all_patch = config.corrupt_type, config.all_patch
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.decoder_pos_embed = _pos_embed(self.trans_dim)
        dpr = [x.item() for x in torch.linspace(0, tc.drop_path_rate, tc.decoder_depth)]
        self.MAE_decoder = TransformerDecoder(self.trans_dim, tc.decoder_depth, tc.decoder_num_heads, dpr)
        self.increase_dim = nn.Sequential(nn.Conv1d(self.trans_dim, 3 * self.group_size, 1))
        nn.init.trunc_normal_(self.mask_token, std=.02, a=-2., b=2.)
        self.loss_func = {'cdl1': chamfer_l1, 'cdl2': chamfer_l2}[config.loss]

    def forward(self, corrupted_pts, pts, mask=None, steps=None, capture=None):
        pts = pts[:, :, :3].contiguous()
        neighborhood, center = group_divider(pts, self.num_group, self.group_size)
        neighborhood = neighborhood + center.unsqueeze(2)
        if steps is None:
            steps = draw_corruption(self.corrupt_type, pts.shape[0])
        t_nb, t_c = apply_corruption(neighborhood, center, steps)
        neighborhood = neighborhood - center.unsqueeze(2)
        t_nb = t_nb - t_c.unsqueeze(2)
        if not self.masked:                         # :717-739: every patch encoded, decoded and reconstructed
            B, G = center.shape[:2]
            x_vis, _ = self.MAE_encoder(t_nb, t_c, torch.zeros(B, G, dtype=torch.bool))
            C = x_vis.shape[-1]
            x_rec = self.MAE_decoder(x_vis, self.decoder_pos_embed(center).reshape(B, -1, C))
            rebuild = self.increase_dim(x_rec.transpose(1, 2)).transpose(1, 2).reshape(B * G, -1, 3)
            gt = neighborhood.reshape(B * G, -1, 3)
            loss1 = self.loss_func(rebuild, gt)
            if capture is not None:
                capture.update(center=center, neighborhood=neighborhood, t_nb=t_nb, t_c=t_c, x_vis=x_vis,
                               x_rec=x_rec, rebuild=rebuild, gt=gt)
            return loss1, torch.zeros(1)
        x_vis, mask = self.MAE_encoder(t_nb, t_c, mask)
        B, _, C = x_vis.shape
        pos_vis = self.decoder_pos_embed(center[~mask]).reshape(B, -1, C)
        pos_mask = self.decoder_pos_embed(center[mask]).reshape(B, -1, C)
        N = pos_mask.shape[1]
        x_full = torch.cat([x_vis, self.mask_token.expand(B, N, -1)], dim=1)
        pos_full = torch.cat([pos_vis, pos_mask], dim=1)
        if self.all_patch == 'True':
            x_rec = self.MAE_decoder(x_full, pos_full)
            gt = torch.cat((neighborhood[~mask].reshape(B, -1, self.group_size, 3),
                            neighborhood[mask].reshape(B, -1, self.group_size, 3)), dim=1)
        else:
            x_rec = self.MAE_decoder(x_full, pos_full, N)
            gt = neighborhood[mask]
        B, M, C = x_rec.shape
        rebuild = self.increase_dim(x_rec.transpose(1, 2)).transpose(1, 2).reshape(B * M, -1, 3)
        gt = gt.reshape(B * M, -1, 3)
        loss1 = self.loss_func(rebuild, gt)
        if capture is not None:
            capture.update(center=center, neighborhood=neighborhood, t_nb=t_nb, t_c=t_c, mask=mask,
                           x_vis=x_vis, x_rec=x_rec, rebuild=rebuild, gt=gt)
        return loss1, torch.zeros(1)


class PointCAE_transformer_fc_global_folding_local(nn.Module):
    """models/PointCAE_transformer.py:919-1088 ('Drop-Patch' branch): FoldingNet
    patch head + FC global head, the variant of the released checkpoints."""

    def __init__(self, config):
        super().__init__()
        tc = config.transformer_config
        self.trans_dim = C = tc.trans_dim
        assert 'Drop-Patch' in config.corrupt_type
        self.MAE_encoder = MaskTransformer(config)
        self.group_size, self.num_group = config.group_size, config.num_group
        self.corrupt_type, self.all_patch = config.corrupt_type, config.all_patch
        self.mask_token = nn.Parameter(torch.zeros(1, 1, C))
        self.decoder_pos_embed = _pos_embed(C)
        dpr = [x.item() for x in torch.linspace(0, tc.drop_path_rate, tc.decoder_depth)]
        self.MAE_decoder = TransformerDecoder(C, tc.decoder_depth, tc.decoder_num_heads, dpr)
        self.coarse_pred = nn.Sequential(nn.Linear(C, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 1024),
                                         nn.ReLU(inplace=True), nn.Linear(1024, 3 * 64))
        self.folding1 = nn.Sequential(nn.Conv1d(C + 2, C, 1), nn.ReLU(), nn.Conv1d(C, C, 1), nn.ReLU(),
                                      nn.Conv1d(C, 3, 1))
        self.folding2 = nn.Sequential(nn.Conv1d(C + 3, C, 1), nn.ReLU(), nn.Conv1d(C, C, 1), nn.ReLU(),
                                      nn.Conv1d(C, 3, 1))
        nn.init.trunc_normal_(self.mask_token, std=.02, a=-2., b=2.)
        self.loss_func = {'cdl1': chamfer_l1, 'cdl2': chamfer_l2}[config.loss]

    def build_grid(self, n):
        import itertools
        x = np.linspace(-0.3, 0.3, 6)
        pts = np.array(list(itertools.product(x, x)))
        return torch.tensor(np.repeat(pts[np.newaxis, ...], n, axis=0)).float()

    def forward(self, corrupted_pts, pts, mask=None, steps=None, capture=None, return_feat=False):
        pts = pts[:, :, :3].contiguous()
        neighborhood, center = group_divider(pts, self.num_group, self.group_size)
        neighborhood = neighborhood + center.unsqueeze(2)
        if steps is None:
            steps = draw_corruption(self.corrupt_type, pts.shape[0])
        t_nb, t_c = apply_corruption(neighborhood, center, steps)
        neighborhood = neighborhood - center.unsqueeze(2)
        t_nb = t_nb - t_c.unsqueeze(2)
        x_vis, mask = self.MAE_encoder(t_nb, t_c, mask)
        B, _, C = x_vis.shape
        global_feature = torch.max(x_vis.transpose(1, 2), dim=-1)[0] + x_vis.mean(1)
        if return_feat:
            return global_feature
        coarse = self.coarse_pred(global_feature).reshape(B, -1, 3)
        pos_vis = self.decoder_pos_embed(center[~mask]).reshape(B, -1, C)
        pos_mask = self.decoder_pos_embed(center[mask]).reshape(B, -1, C)
        N = pos_mask.shape[1]
        x_full = torch.cat([x_vis, self.mask_token.expand(B, N, -1)], dim=1)
        pos_full = torch.cat([pos_vis, pos_mask], dim=1)
        x_rec = self.MAE_decoder(x_full, pos_full) if self.all_patch == 'True' else self.MAE_decoder(x_full, pos_full, N)
        B, M, C = x_rec.shape
        tok = x_rec.reshape(B * M, C).unsqueeze(-1).repeat(1, 1, 36)
        grid = self.build_grid(B * M).transpose(1, 2)
        f1 = self.folding1(torch.cat((tok, grid), dim=1))
        f2 = self.folding2(torch.cat((tok, f1), dim=1)).transpose(1, 2)
        if self.all_patch == 'True':
            gt = torch.cat((neighborhood[~mask].reshape(B, -1, self.group_size, 3),
                            neighborhood[mask].reshape(B, -1, self.group_size, 3)), dim=1).reshape(B * M, -1, 3)
        else:
            gt = neighborhood[mask].reshape(B * M, -1, 3)
        loss1 = self.loss_func(f2, gt)
        loss2 = self.loss_func(coarse, center)
        if capture is not None:
            capture.update(center=center, mask=mask, x_vis=x_vis, x_rec=x_rec, coarse=coarse, fold=f2,
                           global_feature=global_feature)
        return loss1, loss2


# ======================================================================
# Point_CAE_PointNetv2 (BASELINE configs 1-2): models/PointCAE_pointnetv2.py
# :61-173 with the PointNet++ encoder of models/pointnetv2_util.py:319-346.
# The set-abstraction module is third-party pointnet2_ops; its vendored twin
# (extensions/pointnet2/pointnet2_modules.py:31-72,124-158, pytorch_utils.py
# SharedMLP) gives the parameter names used here.
# ======================================================================
def _fps_gather(xyz, npoint):
    idx, ctr = O.furthest_point_sample(_np(xyz), npoint, return_centres=True)
    return torch.from_numpy(ctr)


def _ball_query(radius, nsample, xyz, new_xyz):
    return torch.from_numpy(O.ball_query(radius, nsample, _np(xyz), _np(new_xyz)))


class _ConvBN(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__()
        self.add_module('conv', nn.Conv2d(cin, cout, kernel_size=(1, 1), bias=False))
        bn = nn.Sequential()
        bn.add_module('bn', nn.BatchNorm2d(cout))
        self.add_module('bn', bn)
        self.add_module('activation', nn.ReLU(inplace=True))
        nn.init.kaiming_normal_(self.conv.weight)


class _SharedMLP(nn.Sequential):
    def __init__(self, spec):
        super().__init__()
        for i in range(len(spec) - 1):
            self.add_module('layer{}'.format(i), _ConvBN(spec[i], spec[i + 1]))


class SAModule(nn.Module):
    def __init__(self, mlp, npoint=None, radius=None, nsample=None):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        spec = list(mlp)
        spec[0] += 3                                   # use_xyz=True
        self.mlps = nn.ModuleList([_SharedMLP(spec)])

    def forward(self, xyz, features=None):
        if self.npoint is not None:
            new_xyz = _fps_gather(xyz, self.npoint)
            idx = _ball_query(self.radius, self.nsample, xyz, new_xyz)
            grouped = _Grouping.apply(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
            if features is not None:
                grouped = torch.cat([grouped, _Grouping.apply(features, idx)], dim=1)
        else:
            new_xyz = None
            grouped = xyz.transpose(1, 2).unsqueeze(2)
            if features is not None:
                grouped = torch.cat([grouped, features.unsqueeze(2)], dim=1)
        f = self.mlps[0](grouped)
        f = F.max_pool2d(f, kernel_size=[1, f.size(3)]).squeeze(-1)
        return new_xyz, f


class PointNetv2_encoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.sa1 = SAModule(npoint=512, radius=0.2, nsample=32, mlp=[0, 64, 64, 128])
        self.sa2 = SAModule(npoint=128, radius=0.4, nsample=64, mlp=[128, 128, 128, 256])
        self.sa3 = SAModule(mlp=[256, 256, 512, 1024])

    def forward(self, xyz):
        l1_xyz, l1 = self.sa1(xyz, None)
        l2_xyz, l2 = self.sa2(l1_xyz, l1)
        _, l3 = self.sa3(l2_xyz, l2)
        return l3.view(xyz.shape[0], 1024)


def dropout_global_random(pointcloud, drop_rate=0.5):
    """datasets/corrupt_util.py:572-588: a random half of every cloud (one CPU torch.rand per batch, argsort)."""
    num_samples, num_points = pointcloud.size(0), pointcloud.size(1)
    inx = torch.rand(num_samples, num_points, 1).argsort(1).to(pointcloud.device)
    pointcloud = torch.take_along_dim(pointcloud, inx, dim=1)
    return pointcloud[:, :int(num_points * (1 - drop_rate)), :].contiguous()


def dropout_patch_random(pc_tensor, level=None):
    """datasets/corrupt_util.py:900-924: 64 FPS centres, their 32 nearest points, a random subset of the patches kept
    (python random.random for the level, one CPU torch.rand(64) for the mask; at least one patch)."""
    import random
    if level is None:
        level = random.random() * 4
    prob = level / 10.0 + 0.5
    batch_size, num_points, _ = pc_tensor.shape
    x = _np(pc_tensor[:, :, :3].contiguous())
    _, center = O.furthest_point_sample(x, 64, return_centres=True)
    _, idx = O.knn(x, center, 32)
    idx = torch.from_numpy(idx) + torch.arange(0, batch_size).view(-1, 1, 1) * num_points
    neighborhood = pc_tensor[:, :, :3].reshape(batch_size * num_points, -1)[idx.view(-1), :].view(batch_size, 64, 32, 3)
    group_mask = torch.rand(64) > prob
    if group_mask.sum().item() == 0:
        group_mask[0] = True
    return neighborhood[:, group_mask].reshape(batch_size, -1, 3).contiguous()


def corrupt_in_forward(corrupted_pts, corrupt_type, items):
    """The forward-side dispatch of models/PointCAE_pointnetv2.py:143-149 and models/PointCAE_DGCNN.py:198-221."""
    import random
    for item in corrupt_type:
        if item not in items:
            continue
        if item == 'dropout_patch_pointmae':
            corrupted_pts = dropout_patch_random(corrupted_pts)
        elif item == 'dropout_global':
            corrupted_pts = dropout_global_random(corrupted_pts)
        elif item.startswith('dropout_global_p'):
            corrupted_pts = dropout_global_random(corrupted_pts, drop_rate=int(item[-1]) / 10.0)
        elif item == 'random_dropout':
            if random.random() > 0.5:
                corrupted_pts = dropout_patch_random(corrupted_pts)
            else:
                corrupted_pts = dropout_global_random(corrupted_pts)
    return corrupted_pts


IN_FORWARD = ('dropout_patch_pointmae', 'dropout_global', 'dropout_global_p1', 'dropout_global_p3', 'dropout_global_p5',
              'dropout_global_p7', 'dropout_global_p9', 'random_dropout')


class Point_CAE_PointNetv2(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.corrupt_type = config.corrupt_type
        self.grid_size, self.grid_scale, self.num_coarse = 4, 0.05, 1024
        self.num_fine = self.grid_size ** 2 * self.num_coarse
        self.pointnetv2_encoder = PointNetv2_encoder()
        self.folding1 = nn.Sequential(nn.Linear(1024, 1024), nn.ReLU(), nn.Linear(1024, 1024), nn.ReLU(),
                                      nn.Linear(1024, self.num_coarse * 3))
        self.folding2 = nn.Sequential(nn.Conv1d(1024 + 2 + 3, 512, 1), nn.ReLU(), nn.Conv1d(512, 512, 1), nn.ReLU(),
                                      nn.Conv1d(512, 3, 1))
        self.loss_func = {'cdl1': chamfer_l1, 'cdl2': chamfer_l2}[config.loss]

    def build_grid(self, batch_size):
        import itertools
        x = np.linspace(-self.grid_scale, self.grid_scale, self.grid_size)
        pts = np.array(list(itertools.product(x, x)))
        return torch.tensor(np.repeat(pts[np.newaxis, ...], batch_size, axis=0)).float()

    def forward(self, corrupted_pts, pts, capture=None, **kwargs):
        corrupted_pts = corrupted_pts[:, :, :3].contiguous()
        pts = pts[:, :, :3].contiguous()
        # only the CUDA-side dropouts act here (:144-149)
        corrupted_pts = corrupt_in_forward(corrupted_pts, self.corrupt_type, ('dropout_patch_pointmae', 'dropout_global'))
        feature = self.pointnetv2_encoder(corrupted_pts)
        B = pts.shape[0]
        coarse = self.folding1(feature).view(-1, self.num_coarse, 3)
        g2 = self.grid_size ** 2
        grid_feat = self.build_grid(B).repeat(1, self.num_coarse, 1)
        point_feat = coarse.repeat_interleave(g2, dim=1)
        global_feat = feature.unsqueeze(1).expand(-1, self.num_fine, -1)
        feat = torch.cat([grid_feat, point_feat, global_feat], dim=2)
        fine = self.folding2(feat.transpose(2, 1)).transpose(2, 1) + point_feat
        if capture is not None:
            capture.update(feature=feature, coarse=coarse, fine=fine)
        return self.loss_func(coarse, pts), self.loss_func(fine, pts)


# ----------------------------------------------------------------------------------------------
# Point_CAE_DGCNN_FCOnly (models/PointCAE_DGCNN.py:145-231, encoder models/dgcnn_util.py:7-34,87-136):
# four EdgeConv layers on a k = 20 graph rebuilt in FEATURE space before every layer, concat,
# conv5, global max -> 1024-d feature; three Linear layers -> 1024 coarse points; Chamfer loss.

def dgcnn_knn(x, k):
    """x (B,C,N) -> idx (B,N,k): the k largest of -|xi|^2 + 2 xi.xj - |xj|^2 (dgcnn_util.py:7-12)."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    return (-xx - inner - xx.transpose(2, 1)).topk(k=k, dim=-1)[1]


def dgcnn_graph_feature(x, k=20):
    """(B,C,N) -> (B,2C,N,k) = [x_j - x_i, x_i] over the k neighbours j of i (dgcnn_util.py:15-34)."""
    B, C, N = x.shape
    idx = (dgcnn_knn(x, k) + torch.arange(B, device=x.device).view(-1, 1, 1) * N).view(-1)
    xt = x.transpose(2, 1).contiguous()
    f = xt.view(B * N, C)[idx].view(B, N, k, C)
    c = xt.view(B, N, 1, C).repeat(1, 1, k, 1)
    return torch.cat((f - c, c), dim=3).permute(0, 3, 1, 2)


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
        B = x.shape[0]
        feats = []
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4):
            x = conv(dgcnn_graph_feature(x, k=20)).max(dim=-1)[0]
            feats.append(x)
        x = self.conv5(torch.cat(feats, dim=1))
        return F.adaptive_max_pool1d(x, 1).view(B, -1)


class Point_CAE_DGCNN_FCOnly(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.corrupt_type = config.corrupt_type
        self.num_coarse = 1024
        self.dgcnn_encoder = dgcnn_encoder(channel=3)
        self.recfc = nn.Sequential(nn.Linear(1024, 1024), nn.ReLU(), nn.Linear(1024, 1024), nn.ReLU(),
                                   nn.Linear(1024, self.num_coarse * 3))
        self.loss_func = chamfer_l1 if config.loss == 'cdl1' else chamfer_l2

    def forward(self, corrupted_pts, pts, vis=False, return_feat=False, **kwargs):
        if return_feat:
            return self.dgcnn_encoder(pts[:, :, :3].transpose(1, 2).contiguous())
        corrupted_pts, pts = corrupted_pts[:, :, :3].contiguous(), pts[:, :, :3].contiguous()
        corrupted_pts = corrupt_in_forward(corrupted_pts, self.corrupt_type, IN_FORWARD)        # (:198-221)
        feature = self.dgcnn_encoder(corrupted_pts.transpose(1, 2).contiguous())
        coarse = self.recfc(feature).view(-1, self.num_coarse, 3)
        loss = self.loss_func(coarse, pts)
        return loss, torch.zeros(1)
