"""RMS normalisation of the LLM-side block as one gfx950 kernel each way (csrc/mmfs_norm.hip; C ABI
``mmfs_rmsnorm_forward`` / ``mmfs_rmsnorm_backward``, include/mmfs_msda.h) -- the reference's ``LlamaRMSNorm``
(mm_interleaved/models/decoders/modeling_llama_mmfs.py:53-70) is seven framework kernels per call, and on the
LLM path those "other" kernels were the largest cost (profiles/r02_module_bench_cfg3_cfg4.jsonl).

``rmsnorm_supported`` tells ``MMFSRMSNorm`` whether the kernel applies (device tensor, the input and the gain of
one storage type out of fp32 / fp16 / bf16, a row of whole 16-byte vectors, at most 8192 16-bit channels);
otherwise the module evaluates the same mathematics with framework ops, which is what the CPU tests run.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import MultiScaleDeformableAttention as MSDA

_lib = MSDA._lib
_i64, _vp, _int = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
_lib.mmfs_rmsnorm_supported.restype = _int
_lib.mmfs_rmsnorm_supported.argtypes = [_int, _i64]
_lib.mmfs_rmsnorm_forward.restype = _int
_lib.mmfs_rmsnorm_forward.argtypes = [_int, _vp, _vp, _vp, _vp, _i64, _i64, ctypes.c_float, _vp]
_lib.mmfs_rmsnorm_backward.restype = _int
_lib.mmfs_rmsnorm_backward.argtypes = [_int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]
_lib.mmfs_rmsnorm_backward_partials.restype = _int
_lib.mmfs_rmsnorm_backward_partials.argtypes = [_int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]
_lib.mmfs_rmsnorm_backward_partials_rows.restype = _int
_lib.mmfs_rmsnorm_backward_partials_rows.argtypes = [_i64]
_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_ok = {}


def rmsnorm_supported(x, weight):
    if not (x.is_cuda and x.dtype == weight.dtype and x.dtype in _CODE and x.dim() >= 1 and weight.dim() == 1
            and x.shape[-1] == weight.shape[0]):
        return False
    key = (x.dtype, x.shape[-1])
    ok = _ok.get(key)
    if ok is None:
        ok = _ok[key] = bool(_lib.mmfs_rmsnorm_supported(_CODE[x.dtype], x.shape[-1]))
    return ok


def _rmsnorm_launch(x, weight, eps, keep):
    C = x.shape[-1]
    xc = MSDA._aligned(x.contiguous())
    wc = MSDA._aligned(weight.contiguous())
    rows = xc.numel() // C if C else 0
    y = torch.empty_like(xc)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if keep else None
    with MSDA._on_device(x.device):
        rc = MSDA._launch("mmfs_rmsnorm_fwd", x.device, _lib.mmfs_rmsnorm_forward, _CODE[x.dtype], xc.data_ptr(),
                          wc.data_ptr(), y.data_ptr(), rstd.data_ptr() if keep else None, rows, C, float(eps),
                          MSDA._stream(x.device))
    MSDA._check(rc, "mmfs_rmsnorm_forward")
    return y, xc, wc, rstd


def rmsnorm(x, weight, eps):
    """The kernel behind ``RMSNormFunction``.  Whether a backward will ever ask for the statistics is decided HERE:
    inside ``Function.forward`` ``needs_input_grad`` says "yes" for a Parameter even under ``no_grad``, and
    every decode / sampling step then allocated, wrote and saved ``rstd`` for nothing."""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        return RMSNormFunction.apply(x, weight, eps)
    return _rmsnorm_launch(x, weight, eps, False)[0]


class RMSNormFunction(Function):
    """(x [..., C], weight [C], eps) -> weight * round(x * rsqrt(mean(x^2, -1) + eps)), statistics in fp32."""

    @staticmethod
    def forward(ctx, x, weight, eps):
        keep = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        y, xc, wc, rstd = _rmsnorm_launch(x, weight, eps, keep)
        if keep:
            ctx.save_for_backward(xc, wc, rstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_y):
        xc, wc, rstd = ctx.saved_tensors
        C = xc.shape[-1]
        rows = xc.numel() // C
        gy = MSDA._aligned(grad_y.to(xc.dtype).contiguous())
        gx = torch.empty_like(xc)
        # every workgroup leaves its share of the gain gradient in a row of its own (no atomics), added up here
        parts = torch.empty((int(_lib.mmfs_rmsnorm_backward_partials_rows(rows)), C), dtype=torch.float32, device=xc.device)
        with MSDA._on_device(xc.device):
            rc = MSDA._launch("mmfs_rmsnorm_bwd", xc.device, _lib.mmfs_rmsnorm_backward_partials, _CODE[xc.dtype],
                              gy.data_ptr(), xc.data_ptr(), wc.data_ptr(), rstd.data_ptr(), gx.data_ptr(), parts.data_ptr(),
                              rows, C, MSDA._stream(xc.device))
        MSDA._check(rc, "mmfs_rmsnorm_backward_partials")
        return gx, parts.sum(0).to(wc.dtype), None
