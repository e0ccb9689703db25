"""oracle/msda_torch.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restatement of the reference's only CPU-runnable statement of the op,
``ms_deform_attn_core_pytorch``
(mm_interleaved/models/utils/ops/functions/ms_deform_attn_func.py:47-67):
per-level ``F.grid_sample(bilinear, zeros, align_corners=False)`` on
``2*loc - 1`` followed by the attention-weighted sum, with the backward left
to autograd.  BASELINE.md section 2 names this function as "the reference's CPU
path", so ``bench.py``'s ``cpu_baseline`` leg times this restatement (kind
"port"); tests use it as a second, independent check of the C oracle.

It differs from the CUDA kernel only on measure-zero inputs (NaN locations and
samples exactly on the -1 / W border, SURVEY.md section 8a "Edge semantics");
the C oracle follows the kernel there.
"""
import torch
import torch.nn.functional as F


def msda_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """value [B,S,H,D]; spatial_shapes iterable of (Hl, Wl); loc [B,Nq,H,L,P,2];
    attn [B,Nq,H,L,P]  ->  [B, Nq, H*D]."""
    B, S, H, D = value.shape
    Nq, L, P = sampling_locations.shape[1], sampling_locations.shape[3], sampling_locations.shape[4]
    sizes = [int(h) * int(w) for h, w in spatial_shapes]
    assert sum(sizes) == S and len(sizes) == L
    # one image per (b, h): channels-first maps for grid_sample
    maps = value.permute(0, 2, 3, 1).reshape(B * H, D, S)          # [B*H, D, S]
    grid = (sampling_locations * 2 - 1).permute(0, 2, 1, 3, 4, 5)  # [B, H, Nq, L, P, 2]
    grid = grid.reshape(B * H, Nq, L, P, 2)
    w = attention_weights.permute(0, 2, 1, 3, 4).reshape(B * H, 1, Nq, L, P)
    acc = value.new_zeros(B * H, D, Nq)
    begin = 0
    for lvl, (hl, wl) in enumerate(spatial_shapes):
        hl, wl = int(hl), int(wl)
        level_map = maps[:, :, begin:begin + hl * wl].reshape(B * H, D, hl, wl)
        begin += hl * wl
        taps = F.grid_sample(level_map, grid[:, :, lvl], mode="bilinear",
                             padding_mode="zeros", align_corners=False)    # [B*H, D, Nq, P]
        acc = acc + (taps * w[:, :, :, lvl]).sum(-1)
    return acc.reshape(B, H * D, Nq).transpose(1, 2).contiguous()


def fwd_bwd(value, spatial_shapes, sampling_locations, attention_weights, grad_output=None):
    """One forward + backward; returns (out, grad_value, grad_loc, grad_attn)."""
    v = value.detach().clone().requires_grad_(True)
    l = sampling_locations.detach().clone().requires_grad_(True)
    a = attention_weights.detach().clone().requires_grad_(True)
    out = msda_grid_sample(v, spatial_shapes, l, a)
    if grad_output is None:
        out.sum().backward()
    else:
        out.backward(grad_output.reshape(out.shape))
    return out.detach(), v.grad, l.grad, a.grad
