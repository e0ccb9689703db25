"""``small_linear(x, weight, bias)``: ``F.linear`` for a handful of token rows (a decode step) as one weight-streaming
gfx950 kernel (csrc/mmfs_linear.hip; C ABI ``mmfs_linear_small``, include/mmfs_msda.h) -- what the Linear layers of an
MMFS layer (mm_interleaved/models/utils/ops/modules/mmfs.py:174-176, 274) are at 4 tokens.  Taken when no gradient is
wanted, for device tensors of one 16-bit storage type with at most 8 rows; anything else is ``F.linear`` (which is
what the CPU tests run).  fp32 accumulation, bias added in fp32, one rounding: the roundings of the library call."""
import ctypes

import torch
import torch.nn.functional as F

import MultiScaleDeformableAttention as MSDA

_lib = MSDA._lib
_i64, _vp, _int = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
_lib.mmfs_linear_small_supported.restype = _int
_lib.mmfs_linear_small_supported.argtypes = [_int, _i64, _i64, _i64]
_lib.mmfs_linear_small.restype = _int
_lib.mmfs_linear_small.argtypes = [_int, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]
_lib.mmfs_linear_small_add.restype = _int
_lib.mmfs_linear_small_add.argtypes = [_int, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]
_CODE = {torch.float16: 1, torch.bfloat16: 2}
_ok = {}
enabled = True          # (measurements: False keeps every call on F.linear)


def small_linear(x, weight, bias=None, residual=None):
    """x [..., K], weight [N, K], bias [N] | None -> [..., N]; ``residual`` [..., N] | None is added to the (rounded)
    result -- in the kernel's store where the kernel applies, as a framework add otherwise: the same bits."""
    K = x.shape[-1]
    M = x.numel() // K if K else 0
    def plain():
        y = F.linear(x, weight, bias)
        return y if residual is None else residual + y
    if not (enabled and not torch.is_grad_enabled() and x.is_cuda and x.dtype in _CODE and weight.dtype == x.dtype
            and (bias is None or bias.dtype == x.dtype) and 1 <= M <= 8 and weight.dim() == 2 and weight.shape[1] == K
            and weight.is_contiguous() and x.is_contiguous()):
        return plain()
    N = weight.shape[0]
    if residual is not None and not (residual.dtype == x.dtype and residual.is_contiguous() and residual.is_cuda
                                     and residual.shape == x.shape[:-1] + (N,)):
        return plain()
    key = (x.dtype, M, N, K)
    ok = _ok.get(key)
    if ok is None:
        ok = _ok[key] = bool(_lib.mmfs_linear_small_supported(_CODE[x.dtype], M, N, K))
    if not ok or x.data_ptr() % 16 or weight.data_ptr() % 16:
        return plain()
    y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    b = bias.contiguous() if bias is not None else None
    with MSDA._on_device(x.device):
        rc = MSDA._launch("mmfs_linear_small", x.device, _lib.mmfs_linear_small_add, _CODE[x.dtype], x.data_ptr(),
                          weight.data_ptr(), b.data_ptr() if b is not None else None,
                          residual.data_ptr() if residual is not None else None, y.data_ptr(), M, N, K, K, N, N,
                          MSDA._stream(x.device))
    MSDA._check(rc, "mmfs_linear_small")
    return y
