"""Transformer denoising auto-encoder of the pretraining step, MI355X host side.

Keeps the reference's model API and state_dict layout
(models/PointCAE_transformer.py: Encoder :20-51, Group :54-86, Block :140-158,
MaskTransformer :304-469, PointCAE_transformer :616-742):

    model = MODELS.build(cfg.model)
    loss_xyz, loss_normal = model(corrupted_pts, pts)

Parameter names equal the reference's, so its checkpoints load unchanged.  The
data path is not the reference's: grouping is FPS+centre gather in one launch
and kNN+neighbourhood gather in one launch, the in-forward corruption is one
kernel, tokens are kept flat as (B*T, C) rows for the GEMMs, the visible /
masked token shuffles are two index_selects on precomputed row ids (the
reference boolean-indexes five times), and the loss runs on the packed Chamfer
kernels.  There is no CPU path: the geometry operators raise off-GPU.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nn_ops
from .chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
from .corrupt_util_tensor import corrupt_patches, draw_corruption
from .knn_cuda import knn
from .pointnet2_utils import furthest_point_sample_with_centres
from .registry import MODELS


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    """timm.models.layers.trunc_normal_ (absolute cut-offs a, b)."""
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class Group(nn.Module):
    """FPS centres + kNN patches, centre-subtracted (Group.forward :61-86)."""

    def __init__(self, num_group, group_size):
        super().__init__()
        self.num_group, self.group_size = num_group, group_size

    @torch.no_grad()
    def forward(self, xyz):
        _, center = furthest_point_sample_with_centres(xyz, self.num_group)
        _, _, neighborhood = knn(xyz, center, self.group_size, with_neighbourhood=True)
        return neighborhood, center


class Encoder(nn.Module):
    """mini-PointNet patch embedder (:20-51); parameters live in the reference's
    Sequential layout, the forward is nn_ops.patch_embed."""

    def __init__(self, encoder_channel):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                        nn.Conv1d(128, 256, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, encoder_channel, 1))

    def forward(self, point_groups, groups=None, masked=None):
        """(B,G,n,3) -> tokens (B,G,C); with `groups` (int32 flat group ids) only their
        tokens, as rows (len(groups), C); `masked` = the complementary list."""
        bs, g, n, _ = point_groups.shape
        tok = nn_ops.patch_embed(point_groups.reshape(bs * g, n, 3), self.first_conv, self.second_conv,
                                 self.training, groups, masked)
        return tok if groups is not None else tok.reshape(bs, g, self.encoder_channel)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden_features, in_features)


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    """pre-LN block (:140-158): x += dp(attn(ln1 x)); x += dp(mlp(ln2 x))."""

    def __init__(self, dim, num_heads, drop_path=0.):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * 4.))
        self.attn = Attention(dim, num_heads)
        self.drop_prob = float(drop_path)

    def forward(self, x, pos, B, T, keeps=(None, None), pending=False, pos_grad=None, tail=0):
        """x, pos: (B*T, C) rows; computes block(x + pos).  x may be, and with
        pending=True the result is, an nn_ops.Pending (a branch not yet added to the
        residual stream: the next norm's kernel adds it).  tail: keep only the last `tail` rows of
        every sample after the attention core (nn_ops.transformer_block)."""
        return nn_ops.transformer_block(x, pos, B, T, self, keeps, pending, pos_grad, tail)


def _stack_keeps(stack, B):
    return nn_ops.stack_keeps(stack, B)


class TransformerEncoder(nn.Module):
    def __init__(self, embed_dim, depth, num_heads, drop_path_rate):
        super().__init__()
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, drop_path_rate[i]) for i in range(depth)])
        self.register_buffer('dp_keep', nn_ops.drop_path_keep_buffer(drop_path_rate[:depth]), persistent=False)

    def forward(self, x, pos, B, T):
        keeps = _stack_keeps(self, B)
        pg = nn_ops.PosGrad(len(self.blocks))
        for i, (blk, k) in enumerate(zip(self.blocks, keeps)):   # position re-added before EVERY block (:174-177)
            x = blk(x, pos, B, T, k, pending=True, pos_grad=(pg, i))
        return x                                 # nn_ops.Pending: the caller's norm adds the last branch


class TransformerDecoder(nn.Module):
    def __init__(self, embed_dim, depth, num_heads, drop_path_rate):
        super().__init__()
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, drop_path_rate[i]) for i in range(depth)])
        self.register_buffer('dp_keep', nn_ops.drop_path_keep_buffer(drop_path_rate[:depth]), persistent=False)
        self.norm = nn.LayerNorm(embed_dim)
        self.head = nn.Identity()
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):              # :216-223
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward(self, x, pos, B, T, return_token_num=-1):
        keeps = _stack_keeps(self, B)
        pg = nn_ops.PosGrad(len(self.blocks))
        # Only the last return_token_num tokens of every sample leave the decoder (:229-231, the masked
        # ones), and after its attention core the last block is row-wise: its projection, MLP and the final
        # norm run on those rows only (identical values there; the dropped rows had no reader).
        n = return_token_num if 0 < return_token_num < T else 0
        last = len(self.blocks) - 1
        for i, (blk, k) in enumerate(zip(self.blocks, keeps)):
            x = blk(x, pos, B, T, k, pending=True, pos_grad=(pg, i), tail=n if i == last else 0)
        # LayerNorm is row-wise: norm(x[:, -n:]) == norm(x)[:, -n:]; the norm's kernel also does the last
        # block's bias + DropPath + residual add
        return nn_ops.layer_norm(x, self.norm)


def _pos_embed(dim):
    return nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, dim))


def draw_mask(B, G, mask_ratio, rand_ratio):
    """MaskTransformer._mask_center_rand (:395-422) with the reference's host
    RNG calls: one ratio ~ U(0.5, 0.8) per BATCH when rand_ratio == 'True',
    int(ratio * G) masked groups per sample, numpy shuffle per sample.
    -> (bool mask (B,G) on the host, ratio actually used)."""
    if rand_ratio == 'True':
        mask_ratio = torch.FloatTensor(1).uniform_(0.5, 0.8).item()
    num_mask = int(mask_ratio * G)
    overall = np.zeros([B, G])
    overall[:, G - num_mask:] = 1.0
    for i in range(B):
        np.random.shuffle(overall[i])        # one global-generator shuffle per sample, as the reference
    return torch.from_numpy(overall.astype(np.bool_)), mask_ratio


def mask_row_ids(mask):
    """Flat row ids (into B*G) of the visible and of the masked groups, each in
    ascending group order per sample -- what tokens[~mask] / tokens[mask] select."""
    flat = mask.numpy().reshape(-1)          # numpy: no OpenMP fork/join on the per-step host path
    return torch.from_numpy(np.flatnonzero(~flat)), torch.from_numpy(np.flatnonzero(flat))


class MaskTransformer(nn.Module):
    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        tc = config.transformer_config
        self.mask_ratio, self.rand_ratio = tc.mask_ratio, tc.rand_ratio
        self.trans_dim, self.depth = tc.trans_dim, tc.depth
        self.drop_path_rate, self.num_heads = tc.drop_path_rate, tc.num_heads
        self.encoder_dims = tc.encoder_dims
        self.num_group, self.group_size = config.num_group, config.group_size
        if tc.get('enc_arch', 'PointViT') != 'PointViT':
            raise NotImplementedError("enc_arch '3detr' is outside the pretraining hot path")
        self.mask_type = tc.mask_type
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.pos_embed = _pos_embed(self.trans_dim)
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(self.trans_dim, self.depth, self.num_heads, dpr)
        self.norm = nn.LayerNorm(self.trans_dim)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):              # :353-364
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _mask_center_rand(self, center, noaug=False):
        B, G, _ = center.shape
        if noaug or self.mask_ratio == 0:
            return torch.zeros(B, G, dtype=torch.bool)
        if self.mask_type != 'rand':
            raise NotImplementedError
        mask, self.mask_ratio = draw_mask(B, G, self.mask_ratio, self.rand_ratio)   # stateful, :406-408
        self.num_mask = int(self.mask_ratio * G)
        return mask

    def forward(self, neighborhood, center, noaug=False, mask=None, rows=None):
        """-> x_vis rows (B*Tvis, C), host bool mask (B,G), (vis_rows, mask_rows) on device.
        `rows` = (vis_rows, mask_rows) already on the device skips the host-side
        mask draw (hipGraph replay: nothing on this path may touch the host)."""
        B, G = center.shape[:2]
        vis32 = msk32 = None
        if rows is not None:
            vis_rows, mask_rows = rows[:2]
            if len(rows) > 2:                     # the graphed step uploads the int32 lists too (no cast launches)
                vis32, msk32 = rows[2], rows[3]
        else:
            if mask is None:
                mask = self._mask_center_rand(center, noaug=noaug)
            vis_rows, mask_rows = (r.to(center.device) for r in mask_row_ids(mask.cpu()))
        if vis32 is None:
            vis32, msk32 = vis_rows.to(torch.int32), mask_rows.to(torch.int32)
        Tvis = vis_rows.numel() // B
        # every group goes through the embedder up to its last BatchNorm (:437); the
        # final conv + max-pool, whose output the reference computes for all groups and
        # then drops for the masked ones (:449), runs for the visible groups only
        x_vis = self.encoder(neighborhood, groups=vis32, masked=msk32)
        cut = getattr(self, 'grad_cut', None)
        if cut is not None and x_vis.requires_grad:
            # two-phase backward (graph_step.py): the Transformer's backward stops at a leaf copy of the
            # tokens; the embedder's backward runs later from cut['tokens'].backward(cut['leaf'].grad)
            cut['tokens'] = x_vis
            x_vis = x_vis.detach().requires_grad_()
            cut['leaf'] = x_vis
        pos = nn_ops.pos_embed(center.reshape(B * G, 3), self.pos_embed, rows=vis_rows)
        x_vis = self.blocks(x_vis, pos, B, Tvis)
        return nn_ops.layer_norm(x_vis, self.norm), mask, (vis_rows, mask_rows)


@MODELS.register_module()
class PointCAE_transformer(nn.Module):
    """models/PointCAE_transformer.py:616-742 ('Drop-Patch' configurations)."""

    # parameters whose gradients backward produces last (FlatDataParallel lays them at the ends of the
    # flat buffer; the rest is reduced while their backward still runs)
    late_grad_prefixes = ('MAE_encoder.encoder.',)

    def __init__(self, config):
        super().__init__()
        self.config = config
        tc = config.transformer_config
        self.trans_dim = tc.trans_dim
        # without 'Drop-Patch' the reference builds a NormalTransformer (:473-541): the same modules under the
        # same names, every token kept, no mask drawn -- here the MaskTransformer with an all-visible row list
        self.masked = 'Drop-Patch' in config.corrupt_type
        self.MAE_encoder = MaskTransformer(config)
        self.group_size, self.num_group = config.group_size, config.num_group
        self.corrupt_type, self.all_patch = config.corrupt_type, config.all_patch
        self.drop_path_rate = tc.drop_path_rate
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.trans_dim))
        self.register_buffer('zero_loss', torch.zeros(1), persistent=False)      # the second loss this model returns
        self.decoder_pos_embed = _pos_embed(self.trans_dim)
        self.decoder_depth, self.decoder_num_heads = tc.decoder_depth, tc.decoder_num_heads
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.decoder_depth)]
        self.MAE_decoder = TransformerDecoder(self.trans_dim, self.decoder_depth, self.decoder_num_heads, dpr)
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.increase_dim = nn.Sequential(nn.Conv1d(self.trans_dim, 3 * self.group_size, 1))
        trunc_normal_(self.mask_token, std=.02)
        self.loss = config.loss
        self.build_loss_func(self.loss)

    def build_loss_func(self, loss_type):
        if loss_type == 'cdl1':
            self.loss_func = ChamferDistanceL1()
        elif loss_type == 'cdl2':
            self.loss_func = ChamferDistanceL2()
        else:
            raise NotImplementedError(loss_type)

    def trunk(self, pts, mask=None, steps=None, rows=None, encoder_only=False):
        """Group -> corrupt -> masked encoder -> decoder, shared by the reconstruction
        heads.  -> dict with x_rec rows (B*R, C), the ground-truth patch rows they
        are compared with, and the intermediates the heads / tests need."""
        pts = pts[:, :, :3].contiguous()
        B = pts.shape[0]
        nn_ops.begin_step(pts.device)
        neighborhood, center = self.group_divider(pts)
        if steps is None:
            steps = draw_corruption(self.corrupt_type, B)
        gt_nb, t_nb, t_c = corrupt_patches(neighborhood, center, steps)

        if not self.masked and rows is None:        # (:717-739) nothing masked, nothing drawn
            G0 = self.num_group
            rows = (torch.arange(B * G0, device=pts.device), torch.zeros(0, dtype=torch.int64, device=pts.device))
            mask = torch.zeros(B, G0, dtype=torch.bool)
        if not encoder_only:
            nn_ops.predraw_drop_path(B, [self.MAE_encoder.blocks, self.MAE_decoder])       # one draw for both stacks
        x_vis, mask, (vis_rows, mask_rows) = self.MAE_encoder(t_nb, t_c, mask=mask, rows=rows)
        C = x_vis.shape[-1]
        G = self.num_group
        Tvis = vis_rows.numel() // B
        M = G - Tvis
        out = dict(B=B, C=C, Tvis=Tvis, center=center, gt_nb=gt_nb, t_nb=t_nb, t_c=t_c, mask=mask,
                   x_vis=x_vis.reshape(B, Tvis, C))
        if encoder_only:
            return out
        # decoder positions come from the UN-transformed centres (:695-696)
        ctr = center.reshape(B * G, 3)
        if rows is not None and len(rows) > 4:       # uploaded with the row lists by the graphed step
            order = rows[4]
        else:
            order = torch.cat([vis_rows.reshape(B, Tvis), mask_rows.reshape(B, M)], dim=1).reshape(-1)
        pos_full = nn_ops.pos_embed(ctr, self.decoder_pos_embed, rows=order)
        if M:
            x_full = nn_ops.assemble_tokens(x_vis, self.mask_token, B, Tvis, M)
        else:
            x_full = x_vis
        x_full = x_full.reshape(B * G, C)
        if self.all_patch == 'True' or not self.masked:
            x_rec = self.MAE_decoder(x_full, pos_full, B, G)
            gt_rows, R = order, G
        else:
            x_rec = self.MAE_decoder(x_full, pos_full, B, G, M)
            gt_rows, R = mask_rows, M
        out.update(x_rec=x_rec, R=R,
                   gt_points=gt_nb.reshape(B * G, self.group_size, 3).index_select(0, gt_rows))
        return out

    def forward(self, corrupted_pts, pts, vis=False, mask=None, steps=None, capture=None, rows=None, **kwargs):
        """`corrupted_pts` is ignored on this path, as in the reference (:676).
        `mask` (B,G) bool and `steps` (nsteps,B,10) inject the random draws
        (parity tests); by default they come from the host RNGs like the
        reference's."""
        t = self.trunk(pts, mask=mask, steps=steps, rows=rows)
        B, R = t['B'], t['R']
        rebuild = nn_ops.conv1x1(t['x_rec'], self.increase_dim[0]).reshape(B * R, self.group_size, 3)
        loss1 = self.loss_func(rebuild, t['gt_points'])
        if capture is not None:
            capture.update(center=t['center'], neighborhood=t['gt_nb'], t_nb=t['t_nb'], t_c=t['t_c'], mask=t['mask'],
                           x_vis=t['x_vis'], x_rec=t['x_rec'].reshape(B, R, t['C']), rebuild=rebuild,
                           gt=t['gt_points'])
        return loss1, self.zero_loss           # (the reference returns a fresh zeros(1), :742; a constant costs no launch)


@MODELS.register_module()
class PointCAE_transformer_fc_global_folding_local(PointCAE_transformer):
    """The variant the released Transformer checkpoints were trained with
    (models/PointCAE_transformer.py:919-1088, `--model_name` in rerun2.sh:38-41):
    masked patches are rebuilt by two FoldingNet stages on a 6x6 grid (36 points
    per patch, loss1 = CD(fold, patch)) and a global branch predicts the 64 patch
    centres from max+mean pooled visible tokens (loss2 = CD(coarse, centres)).
    `return_feat=True` returns the pooled global feature (:1025-1026).

    The reference tiles each token 36x and concatenates the grid / first fold
    before the 1x1 convs; here the first conv of each stage is split by column
    block so the token part is multiplied once per patch, not once per point."""

    def __init__(self, config):
        super().__init__(config)
        del self.increase_dim
        C = self.trans_dim
        self.coarse_pred = nn.Sequential(nn.Linear(C, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 1024),
                                         nn.ReLU(inplace=True), nn.Linear(1024, 3 * 64))
        self.folding1 = nn.Sequential(nn.Conv1d(C + 2, C, 1), nn.ReLU(), nn.Conv1d(C, C, 1), nn.ReLU(),
                                      nn.Conv1d(C, 3, 1))
        self.folding2 = nn.Sequential(nn.Conv1d(C + 3, C, 1), nn.ReLU(), nn.Conv1d(C, C, 1), nn.ReLU(),
                                      nn.Conv1d(C, 3, 1))
        x = np.linspace(-0.3, 0.3, 6)
        import itertools
        grid = torch.tensor(np.array(list(itertools.product(x, x)))).float()      # (36, 2), build_grid :990-996
        self.register_buffer('fold_grid', grid, persistent=False)

    @staticmethod
    def _fold(stage, tok, extra_w_cols, extra):
        """stage(cat([tok tiled, extra])) with the first conv split: tok (P, C) once per
        patch, extra (P, 36, e) or (36, e) per point.  -> (P*36, 3)."""
        C = tok.shape[1]
        w = stage[0].weight.squeeze(-1)
        # the conv's two column blocks as separate operands (the narrow one zero-padded to 4 columns), their gradients put back
        # side by side in one launch (nn_ops.split_weight_cols)
        wa, we = nn_ops.split_weight_cols(w, [(0, C), (C, C + extra_w_cols)])
        a = nn_ops.linear_any(tok, wa, stage[0].bias)                                     # (P, C) once per patch
        xe = extra.reshape(-1, extra.shape[-1])
        e = nn_ops.linear_any(nn_ops.pad2d(xe, 0, we.shape[1] - xe.shape[1]) if we.shape[1] != xe.shape[1] else xe, we)
        P, cells = tok.shape[0], extra.shape[-2]
        if extra.dim() == 2:            # one term per grid cell: the fused first layer (csrc/folding.hip)
            # (no per-cloud term here: a row of the step's pre-zeroed arena, no fill launch)
            return nn_ops.fold_mlp(nn_ops.arena.take(C, a)[0].view(1, C), a, e, stage[2], stage[4], 1, P, cells)
        return nn_ops.fold_mlp(None, a, None, stage[2], stage[4], 1, P, cells, row_term=e)   # one term per point

    def forward(self, corrupted_pts, pts, vis=False, return_feat=False, mask=None, steps=None, capture=None,
                rows=None, **kwargs):
        t = self.trunk(pts, mask=mask, steps=steps, rows=rows, encoder_only=return_feat)
        x_vis = t['x_vis']
        global_feature = nn_ops.max_plus_mean(x_vis)                               # (B, C), :1024: x.max(dim=1)[0] + x.mean(1)
        if return_feat:
            return global_feature
        B, R, C = t['B'], t['R'], t['C']
        coarse = nn_ops.mlp_chain(global_feature, [self.coarse_pred[0], self.coarse_pred[2], self.coarse_pred[4]]).reshape(B, -1, 3)
        tok = t['x_rec']                                                           # (B*R, C)
        f1 = self._fold(self.folding1, tok, 2, self.fold_grid).reshape(B * R, 36, 3)
        f2 = self._fold(self.folding2, tok, 3, f1).reshape(B * R, 36, 3)
        loss1 = self.loss_func(f2, t['gt_points'])
        loss2 = self.loss_func(coarse, t['center'])
        if capture is not None:
            capture.update(center=t['center'], mask=t['mask'], x_vis=x_vis, x_rec=tok.reshape(B, R, C),
                           coarse=coarse, fold=f2, global_feature=global_feature)
        return loss1, loss2
