"""oracle/msda_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of ``libmsda_oracle.so`` (oracle/msda_ref.c): the scalar CPU
restatement of the reference's multi-scale deformable attention op
(mm_interleaved/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:240-406).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  Pinned against the reference by
``tests/test_oracle.py`` + ``tests/golden/`` (see DESIGN.md "Oracle").

All functions take/return CPU tensors (or numpy arrays) in fp64 or fp32; 16-bit
inputs are checked by rounding them first and running the fp64 path.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmsda_oracle.so")
_lib = None


def build(force=False):
    """Compile the C restatement with gcc (called by __graft_entry__.build())."""
    src_newer = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("msda_ref.c", "msda_ref_body.inc", "Makefile"))
    if force or src_newer:
        subprocess.run(["make", "-C", _HERE, "-B", "libmsda_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        lib = ctypes.CDLL(_LIB_PATH)
        i64, vp = ctypes.c_int64, ctypes.c_void_p
        for sfx in ("f64", "f32"):
            f = getattr(lib, "msda_ref_forward_" + sfx)
            f.restype = None
            f.argtypes = [vp] * 5 + [i64] * 7 + [vp]
            g = getattr(lib, "msda_ref_backward_" + sfx)
            g.restype = None
            g.argtypes = [vp] * 6 + [i64] * 7 + [vp] * 3
        _lib = lib
    return _lib


def _np(x, dtype):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu()
        if x.dtype in (torch.float16, torch.bfloat16):
            x = x.to(torch.float64)
        x = x.numpy()
    return np.ascontiguousarray(x, dtype=dtype)


def _dims(value, shapes, loc):
    B, S, H, D = value.shape
    L = shapes.shape[0]
    Nq, P = loc.shape[1], loc.shape[4]
    assert loc.shape == (B, Nq, H, L, P, 2), (loc.shape, (B, Nq, H, L, P, 2))
    return B, S, H, D, L, Nq, P


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
            dtype=np.float64):
    """-> numpy [B, Nq, H*D] in ``dtype`` (float64 or float32 arithmetic)."""
    lib = _load()
    v, l, a = _np(value, dtype), _np(sampling_locations, dtype), _np(attention_weights, dtype)
    sh, st = _np(spatial_shapes, np.int64), _np(level_start_index, np.int64)
    B, S, H, D, L, Nq, P = _dims(v, sh, l)
    out = np.empty((B, Nq, H * D), dtype=dtype)
    fn = lib.msda_ref_forward_f64 if dtype == np.float64 else lib.msda_ref_forward_f32
    fn(_ptr(v), _ptr(sh), _ptr(st), _ptr(l), _ptr(a), B, S, H, D, L, Nq, P, _ptr(out))
    return out


def backward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
             grad_output, dtype=np.float64):
    """-> (grad_value, grad_loc, grad_attn) numpy arrays shaped like their inputs."""
    lib = _load()
    v, l, a = _np(value, dtype), _np(sampling_locations, dtype), _np(attention_weights, dtype)
    g = _np(grad_output, dtype)
    sh, st = _np(spatial_shapes, np.int64), _np(level_start_index, np.int64)
    B, S, H, D, L, Nq, P = _dims(v, sh, l)
    assert g.size == B * Nq * H * D
    gv, gl, ga = np.empty_like(v), np.empty_like(l), np.empty_like(a)
    fn = lib.msda_ref_backward_f64 if dtype == np.float64 else lib.msda_ref_backward_f32
    fn(_ptr(v), _ptr(sh), _ptr(st), _ptr(l), _ptr(a), _ptr(g), B, S, H, D, L, Nq, P,
       _ptr(gv), _ptr(gl), _ptr(ga))
    return gv, gl, ga


class OracleMSDAFunction(torch.autograd.Function):
    """Autograd wrapper over the C oracle with the reference's 6-argument call
    shape (ops/functions/ms_deform_attn_func.py:24-44).  Lets tests run the
    module-level code (MMFS and the blocks) on CPU tensors."""

    @staticmethod
    def forward(ctx, value, shapes, start, loc, attn, im2col_step, lazy_zero_attn=False):   # (hint ignored: everything is computed)
        ctx.save_for_backward(value, shapes, start, loc, attn)
        npdt = np.float64 if value.dtype == torch.float64 else np.float32
        out = forward(value, shapes, start, loc, attn, dtype=npdt)
        return torch.from_numpy(out).to(value.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        value, shapes, start, loc, attn = ctx.saved_tensors
        npdt = np.float64 if value.dtype == torch.float64 else np.float32
        gv, gl, ga = backward(value, shapes, start, loc, attn, grad_out.contiguous(), dtype=npdt)
        cast = lambda x: torch.from_numpy(x).to(value.dtype)
        return cast(gv), None, None, cast(gl), cast(ga), None, None
