"""Autograd binding of the multi-scale deformable attention op.

Mirrors the reference's ``MSDeformAttnFunction``
(mm_interleaved/models/utils/ops/functions/ms_deform_attn_func.py:24-44, twin at
mm_interleaved/models/encoders/vit_adapter/ops/functions/ms_deform_attn_func.py:25-49):
same ``apply`` signature, same saved tensors, ``grad_output`` made contiguous, gradients
returned for (value, sampling_locations, attention_weights) only, not twice
differentiable, autocast disabled inside (custom_fwd / custom_bwd without cast_inputs).

The work is done by the gfx950 kernels behind ``MultiScaleDeformableAttention`` (the
drop-in shim next to this package).  There is no fallback implementation here.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import MultiScaleDeformableAttention as MSDA   # mm-interleaved_amd/MultiScaleDeformableAttention.py

_fwd = torch.amp.custom_fwd(device_type="cuda")
_bwd = torch.amp.custom_bwd(device_type="cuda")


class MSDeformAttnFunction(Function):
    @staticmethod
    @_fwd
    def forward(ctx, value, value_spatial_shapes, value_level_start_index,
                sampling_locations, attention_weights, im2col_step, lazy_zero_attn=False):
        # ``lazy_zero_attn`` is an addition to the reference's six arguments (see
        # MSDA.ms_deform_attn_backward): set by MMFS, whose softmax backward never looks at the
        # gradient of a weight that is exactly zero
        ctx.im2col_step = im2col_step
        ctx.lazy_zero_attn = bool(lazy_zero_attn)
        output = MSDA.ms_deform_attn_forward(
            value, value_spatial_shapes, value_level_start_index,
            sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index,
                              sampling_locations, attention_weights)
        return output

    @staticmethod
    @once_differentiable
    @_bwd
    def backward(ctx, grad_output):
        grad_output = grad_output.contiguous()
        value, shapes, start, loc, attn = ctx.saved_tensors
        grad_value, grad_loc, grad_attn = MSDA.ms_deform_attn_backward(
            value, shapes, start, loc, attn, grad_output, ctx.im2col_step, ctx.lazy_zero_attn)
        return grad_value, None, None, grad_loc, grad_attn, None, None


def ms_deform_attn_core_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Same call shape as the reference's debug helper
    (ops/functions/ms_deform_attn_func.py:47-67): ``value_spatial_shapes`` is an
    iterable of (H, W) or an [L, 2] tensor, no level_start_index argument.

    In this package it is NOT a second implementation: it derives level_start_index
    and runs the same HIP op (differentiably).  The independent CPU restatements live
    under ``oracle/`` and are test infrastructure only; CPU tensors raise here.
    """
    shapes = torch.as_tensor(value_spatial_shapes, dtype=torch.long, device=value.device)
    shapes = shapes.reshape(-1, 2).contiguous()
    start = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1])).contiguous()
    return MSDeformAttnFunction.apply(value.contiguous(), shapes, start,
                                      sampling_locations.contiguous(),
                                      attention_weights.contiguous(), 1)
