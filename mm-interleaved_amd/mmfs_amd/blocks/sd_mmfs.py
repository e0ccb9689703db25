"""UNet-side feature synchronizer blocks.

``MMFSBlock`` and ``MMFSNet`` mirror mm_interleaved/models/decoders/sd_mmfs.py:44-272:
LayerNorm(query) + bicubically resized sin-cos position embedding, LayerNorm(features),
MMFS with one reference point per UNet pixel, zero-initialised 1x1 convolution; the net
applies one block to each of the 12 down-path residuals and one to the mid sample,
additively.  State-dict keys are the reference's (``query_norm.*``, ``feat_norm.*``,
``mmfs.*``, ``pos_embed``, ``conv.*``; ``mmfs_down_blocks.{i}.*``, ``mmfs_mid_block.*``).

The position-embedding helpers restate mm_interleaved/models/utils/pos_embed.py:16-97
(constant tables).
"""
import math
from functools import partial

import torch
import torch.nn.functional as F
import torch.utils.checkpoint as cp
from torch import nn

from ..levels import make_level_tables
from ..modules.mmfs import MMFS


# ------------------------------------------------------------------ constant tables
def sincos_1d(dim, pos):
    """[M] positions -> [M, dim]: sin | cos halves, frequencies 1/10000^(2i/dim) (pos_embed.py:78-97)."""
    assert dim % 2 == 0
    omega = 1.0 / (10000.0 ** (torch.arange(dim // 2, dtype=torch.float32) / (dim / 2.0)))
    ang = pos.reshape(-1).to(torch.float32)[:, None] * omega[None, :]
    return torch.cat((ang.sin(), ang.cos()), 1)


def sincos_2d(dim, grid_size):
    """[grid*grid, dim] table, row-major over (y, x): first half encodes y, second half x
    (pos_embed.py:47-75)."""
    assert dim % 2 == 0
    ys, xs = torch.meshgrid(torch.arange(grid_size, dtype=torch.float32),
                            torch.arange(grid_size, dtype=torch.float32), indexing="ij")
    return torch.cat((sincos_1d(dim // 2, ys), sincos_1d(dim // 2, xs)), 1)


def resize_pos_embed(table, n_tokens):
    """[g*g, C] -> [n_tokens, C] by bicubic interpolation when the square sizes differ
    (pos_embed.py:16-40, without its cls-token branch: MMFS tables have none)."""
    src = int(math.sqrt(table.size(0)))
    tgt = int(math.sqrt(n_tokens))
    if src == tgt:
        return table
    out = F.interpolate(table.float().reshape(1, src, src, -1).permute(0, 3, 1, 2), size=(tgt, tgt),
                        mode="bicubic", align_corners=False)
    return out.permute(0, 2, 3, 1).flatten(0, 2).to(table.dtype)


_ref_cache = {}


def pixel_reference_points(h, w, device):
    """[1, h*w, 1, 2] pixel centres (x, y) of an h x w map, normalised (sd_mmfs.py:15-28)."""
    key = (h, w, str(device))
    hit = _ref_cache.get(key)
    if hit is None:
        ys = (torch.arange(h, dtype=torch.float32, device=device) + 0.5) / h
        xs = (torch.arange(w, dtype=torch.float32, device=device) + 0.5) / w
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        hit = torch.stack((xx.reshape(-1), yy.reshape(-1)), -1)[None, :, None, :]
        _ref_cache[key] = hit
    return hit


def deform_inputs(sample, spatial_shapes=((8, 8),), n_images=1):
    """(reference_points, spatial_shapes, level_start_index) for a [B, C, h, w] UNet sample
    (sd_mmfs.py:31-41); ``spatial_shapes`` are one image's levels."""
    _, _, h, w = sample.shape
    shapes, start, _ = make_level_tables(spatial_shapes, n_images, sample.device)
    return pixel_reference_points(h, w, sample.device), shapes, start


# ------------------------------------------------------------------ blocks
class MMFSBlock(nn.Module):
    def __init__(self, attn_dim=1024, query_dim=320, feat_dim=1024, num_heads=16, n_points=8,
                 n_levels=1, deform_ratio=1.0, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                 gradient_checkpointing=False, grid_size=64, offset_init_magnitude=1,
                 max_num_image_per_seq=10, spatial_shapes=[16], base_spatial_shape=8, layer_idx=0):
        super().__init__()
        self.gradient_checkpointing = gradient_checkpointing
        self.query_norm = norm_layer(query_dim)
        self.feat_norm = norm_layer(feat_dim)
        self.mmfs = MMFS(d_model=attn_dim, d_query=query_dim, d_value=feat_dim, d_out=query_dim,
                         n_levels=n_levels, n_heads=num_heads, n_points=n_points, ratio=deform_ratio,
                         offset_init_magnitude=offset_init_magnitude, spatial_shapes=spatial_shapes,
                         base_spatial_shape=base_spatial_shape,
                         max_num_image_per_seq=max_num_image_per_seq, layer_idx=layer_idx)
        self.pos_embed = nn.Parameter(sincos_2d(query_dim, grid_size), requires_grad=False)
        self.conv = nn.Conv2d(query_dim, query_dim, kernel_size=1, stride=1)
        nn.init.zeros_(self.conv.weight)          # zero_module (sd_mmfs.py:88-94, 148-151)
        nn.init.zeros_(self.conv.bias)

    def _reset_parameters(self):
        self.mmfs._reset_parameters()

    def _inner(self, sample, ms_feat, ms_feat_mask, spatial_shapes):
        B, C, H, W = sample.shape
        n_images = ms_feat_mask.shape[-1]
        ref, shapes, start = deform_inputs(sample, spatial_shapes, n_images)
        query = self.query_norm(sample.flatten(2).transpose(1, 2))            # b (h w) c
        query = query + resize_pos_embed(self.pos_embed, H * W)
        out = self.mmfs(query, ref, self.feat_norm(ms_feat), shapes, start,
                        input_padding_mask=None, attention_mask=ms_feat_mask)
        return self.conv(out.transpose(1, 2).reshape(B, C, H, W))

    def forward(self, sample, ms_feat, ms_feat_mask, spatial_shapes):
        """sample [B, C_q, H, W]; ms_feat [B, n, sum_l H_l*W_l, C_v]; ms_feat_mask [B, n];
        spatial_shapes: the levels of ONE image, list of (H_l, W_l)  ->  [B, C_q, H, W]."""
        spatial_shapes = [tuple(int(v) for v in s) for s in spatial_shapes]
        if self.gradient_checkpointing and self.training:
            # the op is stateless and re-entrant: the forward is simply re-run in backward
            return cp.checkpoint(self._inner, sample, ms_feat, ms_feat_mask, spatial_shapes,
                                 use_reentrant=False)
        return self._inner(sample, ms_feat, ms_feat_mask, spatial_shapes)


class MMFSNet(nn.Module):
    def __init__(self, input_channel, block_out_channels, layers_per_block, downsample_factor=1,
                 n_levels=4, n_points=8, gradient_checkpointing=True, spatial_shapes=[64, 32, 16, 8]):
        super().__init__()
        self.downsample_factor = downsample_factor
        sd_shapes = [s // downsample_factor for s in spatial_shapes]

        def block(query_dim, index):
            # three UNet residuals share a resolution: residual i is at sd_shapes[i // 3]
            return MMFSBlock(query_dim=query_dim, feat_dim=input_channel, n_points=n_points,
                             n_levels=n_levels, gradient_checkpointing=gradient_checkpointing,
                             grid_size=64 // downsample_factor, spatial_shapes=spatial_shapes,
                             base_spatial_shape=sd_shapes[index // 3], layer_idx=index)

        # the UNet's down path: conv_in residual, then per stage `layers_per_block` resnet
        # residuals plus one downsampler residual (not after the last stage) (sd_mmfs.py:185-211)
        channels = [block_out_channels[0]]
        for i, ch in enumerate(block_out_channels):
            channels += [ch] * layers_per_block
            if i != len(block_out_channels) - 1:
                channels.append(ch)
        self.mmfs_down_blocks = nn.ModuleList(block(ch, i) for i, ch in enumerate(channels))
        mid = MMFSBlock(query_dim=block_out_channels[-1], feat_dim=input_channel, n_points=n_points,
                        n_levels=n_levels, gradient_checkpointing=gradient_checkpointing,
                        grid_size=64 // downsample_factor, spatial_shapes=spatial_shapes,
                        base_spatial_shape=sd_shapes[-1], layer_idx=len(channels))
        self.mmfs_mid_block = mid
        self._reset_parameters()

    def _reset_parameters(self):
        for blk in self.mmfs_down_blocks:
            blk._reset_parameters()
        self.mmfs_mid_block._reset_parameters()

    def forward(self, sample, down_block_res_samples, mmfs_features, mmfs_mask):
        """sample: mid-block input; down_block_res_samples: the UNet's down residuals;
        mmfs_features: per level [B, n, C, h_l, w_l]; mmfs_mask [B, n]
        -> (sample', tuple of residuals')   (sd_mmfs.py:230-272)."""
        assert len(down_block_res_samples) == len(self.mmfs_down_blocks)
        shapes = [(f.shape[-2], f.shape[-1]) for f in mmfs_features]
        bank = torch.cat([f.flatten(3).transpose(2, 3) for f in mmfs_features], dim=2)   # b n (h w) c
        new_res = tuple(r + blk(r, bank, mmfs_mask, shapes)
                        for r, blk in zip(down_block_res_samples, self.mmfs_down_blocks))
        sample = sample + self.mmfs_mid_block(sample, bank, mmfs_mask, shapes)
        return sample, new_res
