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


# ---------------------------------------------------------------- many tokens, few features: the weight gradient
def _split_k(T, N, K):
    """Chunks of the token axis for dW = g^T x (``TokenLinearFunction``): the BLAS call gives every output tile ONE
    workgroup that walks all T tokens -- 4 tiles for a 320 x 320 weight and 32768 tokens (the image decoder's first
    blocks), 117 us at 0.06 PFLOP/s; as a batch of S shorter products + a sum: 33 us (tools/gemm_splitk.py, r04zj).
    Largest power of two <= min(T / 1024, 256 / tiles); 1 = the plain call (few tokens, or enough tiles anyway)."""
    if T < 4096:
        return 1
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    cap = min(T // 1024, 256 // tiles)
    s = 1
    while 2 * s <= cap and T % (2 * s) == 0:
        s *= 2
    return s


class TokenLinearFunction(torch.autograd.Function):
    """``F.linear(x, weight, bias)`` for [..., K] activations whose weight gradient sums over MANY token rows: the same
    products as autograd's backward of ``F.linear``, the token sum of dW cut into ``_split_k`` chunks (a batched GEMM +
    one reduction; every partial product accumulated in fp32, rounded to the storage type, summed in fp32)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        N, K = weight.shape
        g2, x2 = grad.reshape(-1, N), x.reshape(-1, K)
        T = g2.shape[0]
        dx = (g2 @ weight).reshape(x.shape) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            S = _split_k(T, N, K)
            if S > 1 and g2.is_contiguous() and x2.is_contiguous():
                dw = torch.bmm(g2.view(S, T // S, N).transpose(1, 2), x2.view(S, T // S, K)).sum(0)
            else:
                dw = g2.t() @ x2
        db = g2.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return dx, dw, db


def token_linear(x, layer_or_weight, bias=None):
    """A Linear layer on token rows.  ``layer_or_weight``: an ``nn.Linear`` (called as the module it is when it is not a
    plain one or a hook could observe it) or a weight tensor (+ ``bias``).  With gradients, on the device, for 16-bit
    storage and enough tokens for ``_split_k`` to cut: ``TokenLinearFunction``; else ``F.linear``."""
    if isinstance(layer_or_weight, torch.nn.Module):
        layer = layer_or_weight
        from ..levels import hook_free
        if type(layer) is not torch.nn.Linear or not hook_free(layer):
            return layer(x)
        weight, bias = layer.weight, layer.bias
    else:
        weight = layer_or_weight
    if (torch.is_grad_enabled() and x.is_cuda and x.dtype in _CODE and weight.dtype == x.dtype and weight.requires_grad
            and not torch.is_autocast_enabled() and weight.dim() == 2
            and _split_k(x.numel() // max(1, x.shape[-1]), weight.shape[0], weight.shape[1]) > 1):
        return TokenLinearFunction.apply(x, weight, bias)
    return F.linear(x, weight, bias)
