"""The two layout changes around the image decoder's synchronizer block as one gfx950 kernel each
(csrc/mmfs_query.hip; C ABI ``mmfs_query_prep`` / ``mmfs_tokens_add``, include/mmfs_msda.h):

``query_prep(sample, norm, pos)``   = ``norm(rearrange(sample, "b c h w -> b (h w) c")) + pos``
                                      (mm_interleaved/models/decoders/sd_mmfs.py:124-131)
``tokens_add(tokens, residual)``    = ``residual + rearrange(tokens, "b (h w) c -> b c h w")``
                                      (sd_mmfs.py:146 and the caller's add, :262-270)

``query_prep`` / ``tokens_add`` are the plain calls (no autograd graph: sampling, 30 denoising steps per image);
``QueryPrepFunction`` / ``TokensAddFunction`` run the same kernels in a training step's forward (twice under gradient
checkpointing) and evaluate the backward with the framework's own LayerNorm backward on the statistics the kernel
returned.  Anything the kernels do not take (fp32 storage, CPU tensors: what the CPU tests run) stays on the framework's
kernels in ``MMFSBlock``.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import MultiScaleDeformableAttention as MSDA

_lib = MSDA._lib
_i64, _vp, _int = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
_lib.mmfs_query_prep_supported.restype = _int
_lib.mmfs_query_prep_supported.argtypes = [_int, _i64, _i64]
_lib.mmfs_query_prep.restype = _int
_lib.mmfs_query_prep.argtypes = [_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_float, _vp]
_lib.mmfs_tokens_add.restype = _int
_lib.mmfs_tokens_add.argtypes = [_int, _vp, _vp, _vp, _i64, _i64, _i64, _vp]
_CODE = {torch.float16: 1, torch.bfloat16: 2}
_ok = {}


def layout_supported(sample):
    """A [B, C, H, W] device tensor of a 16-bit type whose shape the kernels take."""
    if not (sample.is_cuda and sample.dim() == 4 and sample.dtype in _CODE and sample.shape[0] <= 65535):
        return False
    key = (sample.dtype, sample.shape[1], sample.shape[2] * sample.shape[3])
    ok = _ok.get(key)
    if ok is None:
        ok = _ok[key] = bool(_lib.mmfs_query_prep_supported(_CODE[sample.dtype], key[1], key[2]))
    return ok


def query_prep(sample, weight, bias, eps, pos=None, stats=False):
    """sample [B, C, H, W]; weight, bias [C]; pos [H*W, C] or None -> [B, H*W, C] (no autograd);
    ``stats``: also the rows' (mean, rstd) [B, H*W, 1] fp32, what a LayerNorm backward needs."""
    B, C, H, W = sample.shape
    x = MSDA._aligned(sample.contiguous())
    g, b = MSDA._aligned(weight.contiguous()), MSDA._aligned(bias.contiguous())
    p = MSDA._aligned(pos.contiguous()) if pos is not None else None
    assert g.dtype == x.dtype and b.dtype == x.dtype and (p is None or (p.dtype == x.dtype and p.shape == (H * W, C)))
    q = torch.empty((B, H * W, C), dtype=x.dtype, device=x.device)
    mean = torch.empty((B, H * W, 1), dtype=torch.float32, device=x.device) if stats else None
    rstd = torch.empty_like(mean) if stats else None
    with MSDA._on_device(x.device):
        rc = MSDA._launch("mmfs_query_prep", x.device, _lib.mmfs_query_prep, _CODE[x.dtype], x.data_ptr(), g.data_ptr(),
                          b.data_ptr(), p.data_ptr() if p is not None else None, q.data_ptr(),
                          mean.data_ptr() if stats else None, rstd.data_ptr() if stats else None,
                          B, C, H * W, float(eps), MSDA._stream(x.device))
    MSDA._check(rc, "mmfs_query_prep")
    return (q, mean, rstd) if stats else q


class QueryPrepFunction(Function):
    """(sample [B, C, H, W], weight [C], bias [C], eps, pos [H*W, C] | None) -> LayerNorm_C(sample as tokens) + pos.
    Forward: ``mmfs_query_prep``; backward: the framework's LayerNorm backward on the token view of the sample with the
    kernel's statistics (gradients for sample, weight, bias; for ``pos`` the sum over the batch)."""

    @staticmethod
    def forward(ctx, sample, weight, bias, eps, pos):
        q, mean, rstd = query_prep(sample, weight, bias, eps, pos, stats=True)
        ctx.save_for_backward(sample, weight, bias, mean, rstd)
        ctx.pos_grad = pos is not None and pos.requires_grad
        return q

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_q):
        sample, weight, bias, mean, rstd = ctx.saved_tensors
        B, C, H, W = sample.shape
        tok = sample.flatten(2).transpose(1, 2)
        mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]]
        gq = grad_q.to(sample.dtype)
        gx, gw, gb = torch.ops.aten.native_layer_norm_backward(gq, tok, [C], mean, rstd, weight, bias, mask)
        gs = gx.transpose(1, 2).reshape(B, C, H, W) if mask[0] else None
        return gs, gw, gb, None, (gq.sum(0) if ctx.pos_grad else None)


def tokens_add(tokens, residual):
    """tokens [B, H*W, C] + residual [B, C, H, W] -> [B, C, H, W] (no autograd)."""
    B, C, H, W = residual.shape
    assert tokens.shape == (B, H * W, C) and tokens.dtype == residual.dtype
    t = MSDA._aligned(tokens.contiguous())
    r = MSDA._aligned(residual.contiguous())
    y = torch.empty_like(r)
    with MSDA._on_device(r.device):
        rc = MSDA._launch("mmfs_tokens_add", r.device, _lib.mmfs_tokens_add, _CODE[r.dtype], t.data_ptr(), r.data_ptr(),
                          y.data_ptr(), B, C, H * W, MSDA._stream(r.device))
    MSDA._check(rc, "mmfs_tokens_add")
    return y


class TokensAddFunction(Function):
    """(tokens [B, H*W, C], residual [B, C, H, W]) -> residual + tokens as a [B, C, H, W] map (``mmfs_tokens_add``);
    the gradient is the incoming one for the residual and its token view for the tokens."""

    @staticmethod
    def forward(ctx, tokens, residual):
        return tokens_add(tokens, residual)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        gt = grad_y.flatten(2).transpose(1, 2) if ctx.needs_input_grad[0] else None
        return gt, (grad_y if ctx.needs_input_grad[1] else None)
