"""MSDeformAttn -- the single-image multi-scale deformable attention module the
ViT-Adapter encoder uses (second user of the same native op).

Interface, parameter names and initialisation follow
mm_interleaved/models/encoders/vit_adapter/ops/modules/ms_deform_attn.py:28-131
(d_model=1024, H=16, P=4, ratio=0.5 -> D=32 in the adapter; L=3 injector, L=1 extractor).
The op runs in the gfx950 kernels; projections are ``nn.Linear`` (hipBLASLt / MFMA).
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from ..functions import MSDeformAttnFunction


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, ratio=1.0):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError(f"d_model must be divisible by n_heads, but got {d_model} and {n_heads}")
        per_head = d_model // n_heads
        if per_head & (per_head - 1):
            warnings.warn("MSDeformAttn: a power-of-two head width takes the vector kernels; "
                          f"{per_head} falls back to the scalar path")
        self.im2col_step = 1
        self.d_model, self.n_levels, self.n_heads, self.n_points, self.ratio = \
            d_model, n_levels, n_heads, n_points, ratio
        d_inner = int(d_model * ratio)
        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_inner)
        self.output_proj = nn.Linear(d_inner, d_model)
        self._reset_parameters()

    def _reset_parameters(self):
        """Directional offset prior of Deformable DETR (ms_deform_attn.py:64-81): head h
        looks along angle 2*pi*h/H, point i at distance i+1."""
        H, L, P = self.n_heads, self.n_levels, self.n_points
        ang = torch.arange(H, dtype=torch.float32) * (2.0 * math.pi / H)
        dirs = torch.stack((ang.cos(), ang.sin()), -1)
        dirs = dirs / dirs.abs().max(-1, keepdim=True).values
        steps = torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, P, 1)
        bias = (dirs.view(H, 1, 1, 2) * steps).expand(H, L, P, 2)
        with torch.no_grad():
            self.sampling_offsets.weight.zero_()
            self.sampling_offsets.bias.copy_(bias.reshape(-1))
            self.attention_weights.weight.zero_()
            self.attention_weights.bias.zero_()
            nn.init.xavier_uniform_(self.value_proj.weight)
            self.value_proj.bias.zero_()
            nn.init.xavier_uniform_(self.output_proj.weight)
            self.output_proj.bias.zero_()

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes,
                input_level_start_index, input_padding_mask=None):
        """query [N, Lq, C]; reference_points [N, Lq, L, 2|4]; input_flatten [N, S, C];
        input_spatial_shapes [L, 2]; input_level_start_index [L]; padding mask [N, S] -> [N, Lq, C]."""
        N, Lq, _ = query.shape
        S = input_flatten.shape[1]
        H, L, P = self.n_heads, self.n_levels, self.n_points
        value = self.value_proj(input_flatten)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None], 0.0)
        value = value.view(N, S, H, -1)
        offsets = self.sampling_offsets(query).view(N, Lq, H, L, P, 2)
        attn = F.softmax(self.attention_weights(query).view(N, Lq, H, L * P), -1).view(N, Lq, H, L, P)
        if reference_points.shape[-1] == 2:
            wh = torch.stack((input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]), -1)
            loc = reference_points[:, :, None, :, None, :] + offsets / wh[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            loc = reference_points[:, :, None, :, None, :2] + \
                offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(
                f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")
        out = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                         loc.to(value.dtype).contiguous(), attn.contiguous(), self.im2col_step)
        return self.output_proj(out)
