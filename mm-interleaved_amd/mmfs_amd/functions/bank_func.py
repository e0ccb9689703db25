"""Multi-image feature bank on the device: per-level feature maps + one image
index per bank slot -> the token-major bank MMFS reads, in ONE pass (csrc/mmfs_bank.hip;
C ABI ``mmfs_bank_gather`` / ``mmfs_bank_scatter`` in include/mmfs_msda.h).  Replaces the Python
loops + rearrange + concatenation of mm_interleaved/models/mm_interleaved.py:223-250 and
decoders/sd_mmfs.py:241-245.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import MultiScaleDeformableAttention as MSDA

_lib = MSDA._lib
_i64, _vp, _int = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
for _f in (_lib.mmfs_bank_gather, _lib.mmfs_bank_scatter):
    _f.restype = _int
    _f.argtypes = [_int, _int, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]
_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
MAX_LEVELS, MAX_SLOTS = 8, 65535


def bank_gather_supported(levels, n_slots):
    f = levels[0]
    return (f.is_cuda and f.dtype in _CODE and 1 <= len(levels) <= MAX_LEVELS and 0 < n_slots <= MAX_SLOTS
            and 0 < f.shape[0] <= MAX_SLOTS and all(g.dtype == f.dtype and g.shape[:2] == f.shape[:2] for g in levels))


def _tables(levels):
    ptrs = (ctypes.c_void_p * len(levels))(*[f.data_ptr() for f in levels])
    hw = (ctypes.c_int64 * len(levels))(*[f.shape[2] * f.shape[3] for f in levels])
    return ptrs, hw


class BankGatherFunction(Function):
    """(src_index [n_slots] long, level_0 [N_img, C, h_0, w_0], level_1, ...) -> bank
    [n_slots, sum_l h_l*w_l, C]; a slot whose index is negative (or past the last image) is zero."""

    @staticmethod
    def forward(ctx, src_index, *levels):
        levels = [f.contiguous() for f in levels]
        f0 = levels[0]
        n_img, C = f0.shape[0], f0.shape[1]
        src_index = src_index.to(device=f0.device, dtype=torch.long).contiguous()
        n_slots = src_index.numel()
        S = sum(f.shape[2] * f.shape[3] for f in levels)
        bank = torch.empty((n_slots, S, C), dtype=f0.dtype, device=f0.device)
        ptrs, hw = _tables(levels)
        with torch.cuda.device(f0.device):
            rc = MSDA._launch("mmfs_bank_gather", f0.device, _lib.mmfs_bank_gather, _CODE[f0.dtype], len(levels),
                              ctypes.cast(ptrs, _vp), ctypes.cast(hw, _vp), src_index.data_ptr(), bank.data_ptr(),
                              n_img, C, n_slots, MSDA._stream(f0.device))
        MSDA._check(rc, "mmfs_bank_gather")
        ctx.save_for_backward(src_index)
        ctx.level_shapes = [tuple(f.shape) for f in levels]
        return bank

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_bank):
        (src_index,) = ctx.saved_tensors
        grad_bank = grad_bank.contiguous()
        grads = [torch.empty(s, dtype=grad_bank.dtype, device=grad_bank.device) for s in ctx.level_shapes]
        n_img, C = ctx.level_shapes[0][:2]
        ptrs, hw = _tables(grads)
        with torch.cuda.device(grad_bank.device):
            rc = MSDA._launch("mmfs_bank_scatter", grad_bank.device, _lib.mmfs_bank_scatter, _CODE[grad_bank.dtype],
                              len(grads), ctypes.cast(ptrs, _vp), ctypes.cast(hw, _vp), src_index.data_ptr(),
                              grad_bank.data_ptr(), n_img, C, src_index.numel(), MSDA._stream(grad_bank.device))
        MSDA._check(rc, "mmfs_bank_scatter")
        return (None, *grads)
