"""Drop-in for the reference's native extension module ``MultiScaleDeformableAttention``.

The reference imports it as ``import MultiScaleDeformableAttention as MSDA`` in
  mm_interleaved/models/utils/ops/functions/ms_deform_attn_func.py:18-21
  mm_interleaved/models/encoders/vit_adapter/ops/functions/ms_deform_attn_func.py:19-22
and calls exactly two functions, exported by its pybind module
(mm_interleaved/models/utils/ops/src/vision.cpp:13-16):

    ms_deform_attn_forward(value, spatial_shapes, level_start_index,
                           sampling_loc, attn_weight, im2col_step) -> Tensor[B, Nq, H*D]
    ms_deform_attn_backward(value, spatial_shapes, level_start_index,
                            sampling_loc, attn_weight, grad_output, im2col_step)
        -> [grad_value, grad_sampling_loc, grad_attn_weight]

Putting this directory on ``sys.path`` makes that import resolve here; the two
functions keep the reference's argument order, preconditions and error
behaviour (ms_deform_attn_cuda.cu:29-53, 94-118) and run the hand-written gfx950
kernels of ``libmmfs_msda.so`` through its C ABI (include/mmfs_msda.h) with ctypes.

There is NO CPU or PyTorch fallback: CPU tensors raise, exactly as the reference
does (src/ms_deform_attn.h:29-38: "Not implemented on the CPU"), and a missing
shared library raises at import time.
"""
import ctypes
import os

import numpy as np
import torch

__all__ = ["ms_deform_attn_forward", "ms_deform_attn_backward", "register_level_tables", "library_path",
           "build_info", "check_level_table_status"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("MMFS_MSDA_LIB", os.path.join(_HERE, "libmmfs_msda.so"))

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        f"MultiScaleDeformableAttention: {_LIB_PATH} not found. Build it with "
        f"`make -C {os.path.join(_HERE, 'csrc')}` (hipcc, gfx950); there is no fallback path.")

_lib = ctypes.CDLL(_LIB_PATH)

_ABI_VERSION = 13
_i64, _vp, _int = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int

_lib.mmfs_msda_abi_version.restype = _int
_lib.mmfs_msda_abi_version.argtypes = []
_lib.mmfs_msda_build_info.restype = ctypes.c_char_p
_lib.mmfs_msda_build_info.argtypes = []
_lib.mmfs_msda_status_string.restype = ctypes.c_char_p
_lib.mmfs_msda_status_string.argtypes = [_int]
_lib.mmfs_msda_forward.restype = _int
_lib.mmfs_msda_forward.argtypes = [_int] + [_vp] * 6 + [_i64] * 7 + [_vp]
_lib.mmfs_msda_forward_flags.restype = _int
_lib.mmfs_msda_forward_flags.argtypes = [_int] + [_vp] * 6 + [_i64] * 7 + [ctypes.c_uint, _vp]
_lib.mmfs_msda_backward.restype = _int
_lib.mmfs_msda_backward.argtypes = [_int] + [_vp] * 10 + [_i64] * 8 + [ctypes.c_uint, _vp]
_lib.mmfs_msda_backward_checked.restype = _int
_lib.mmfs_msda_backward_checked.argtypes = [_int] + [_vp] * 10 + [_i64] * 8 + [ctypes.c_uint, _vp, _vp]
_lib.mmfs_msda_backward_taps_fused.restype = _int
_lib.mmfs_msda_backward_taps_fused.argtypes = [_int] + [_i64] * 7 + [ctypes.c_uint]
_lib.mmfs_msda_backward_workspace_bytes.restype = _i64
_lib.mmfs_msda_backward_workspace_bytes.argtypes = [_int] + [_i64] * 7 + [ctypes.c_uint]
_lib.mmfs_msda_backward_taps.restype = _int
_lib.mmfs_msda_backward_taps.argtypes = [_int] + [_vp] * 8 + [_i64] * 7 + [_vp]
_lib.mmfs_msda_backward_value.restype = _int
_lib.mmfs_msda_backward_value.argtypes = [_int] + [_vp] * 7 + [_i64] * 8 + [_vp]
_lib.mmfs_msda_backward_value_prepare.restype = _int
_lib.mmfs_msda_backward_value_prepare.argtypes = [_int] + [_vp] * 3 + [_i64] * 8 + [_vp]
_lib.mmfs_msda_backward_value_run.restype = _int
_lib.mmfs_msda_backward_value_run.argtypes = [_int] + [_vp] * 5 + [_i64] * 8 + [_vp]
_lib.mmfs_msda_backward_value_sort.restype = _int
_lib.mmfs_msda_backward_value_sort.argtypes = [_int] + [_vp] * 3 + [_i64] * 8 + [_vp]
_lib.mmfs_msda_backward_value_reduce.restype = _int
_lib.mmfs_msda_backward_value_reduce.argtypes = [_int] + [_vp] * 3 + [_i64] * 8 + [_vp]
_lib.mmfs_msda_backward_hybrid_workspace_bytes.restype = _i64
_lib.mmfs_msda_backward_hybrid_workspace_bytes.argtypes = [_int, _vp, _vp] + [_i64] * 7 + [ctypes.c_uint]
_lib.mmfs_msda_backward_hybrid.restype = _int
_lib.mmfs_msda_backward_hybrid.argtypes = [_int] + [_vp] * 12 + [_i64] * 8 + [ctypes.c_uint, ctypes.c_uint, _vp]
_lib.mmfs_msda_backward_sorted_workspace_bytes.restype = _i64
_lib.mmfs_msda_backward_sorted_workspace_bytes.argtypes = [_int] + [_i64] * 7 + [ctypes.c_uint]
_lib.mmfs_msda_backward_sorted.restype = _int
_lib.mmfs_msda_backward_sorted.argtypes = [_int] + [_vp] * 10 + [_i64] * 9 + [ctypes.c_uint, ctypes.c_uint, _vp]
_lib.mmfs_msda_cast_from_f32.restype = _int
_lib.mmfs_msda_cast_from_f32.argtypes = [_int, _vp, _vp, _i64, _vp]

_lib.mmfs_env_reload.restype = None
_lib.mmfs_env_reload.argtypes = []
_lib.mmfs_env_knob.restype = _int
_lib.mmfs_env_knob.argtypes = [_int] + [ctypes.POINTER(ctypes.c_char_p)] * 3

if _lib.mmfs_msda_abi_version() != _ABI_VERSION:
    raise ImportError(f"{_LIB_PATH}: ABI version {_lib.mmfs_msda_abi_version()} != {_ABI_VERSION}")

# enum mmfs_dtype (include/mmfs_msda.h)
_DTYPE_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2, torch.float64: 3}


def library_path():
    return _LIB_PATH


def reload_env():
    """The library reads its ``MMFS_*`` tuning / test knobs ONCE (csrc/msda_env.h); a process that changes one afterwards
    (the tests do, to hold every formulation to the oracle) says so here.  Also drops what this module derived from them."""
    _lib.mmfs_env_reload()
    _ws_cache.clear()


def env_knobs():
    """[(name, documentation, value or None)] -- the library's whole table of environment knobs."""
    out, i = [], 0
    name, doc, val = ctypes.c_char_p(), ctypes.c_char_p(), ctypes.c_char_p()
    while _lib.mmfs_env_knob(i, ctypes.byref(name), ctypes.byref(doc), ctypes.byref(val)) == 0:
        out.append((name.value.decode(), doc.value.decode(), None if val.value is None else val.value.decode()))
        i += 1
    return out


def build_info():
    return _lib.mmfs_msda_build_info().decode()


def _check(status, what):
    if status != 0:
        raise RuntimeError(f"{what}: {_lib.mmfs_msda_status_string(status).decode()} (status {status})")


def _require(cond, msg):
    # AT_ASSERTM in the reference raises a RuntimeError
    if not cond:
        raise RuntimeError(msg)


def _validate(named, value):
    # (messages are only built when a check fails: this runs on every call of a ~100 us step)
    dev = value.device
    for name, t in named:
        if not isinstance(t, torch.Tensor):
            raise RuntimeError(f"{name} must be a tensor")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
        # PyTorch-ROCm exposes HIP devices under the "cuda" device type
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")   # reference wording, .cu:35-39
        if t.device != dev:
            raise RuntimeError(f"{name} must be on the same device as value")


def _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight):
    if value.dim() != 4:
        raise RuntimeError("value must be [B, S, H, D]")
    B, S, H, D = value.shape
    if spatial_shapes.dim() != 2 or spatial_shapes.shape[1] != 2:
        raise RuntimeError("spatial_shapes must be [L, 2]")
    L = spatial_shapes.shape[0]
    if sampling_loc.dim() != 6:
        raise RuntimeError("sampling_loc must be [B, Nq, H, L, P, 2]")
    Nq, P = sampling_loc.shape[1], sampling_loc.shape[4]
    if sampling_loc.shape != (B, Nq, H, L, P, 2):
        raise RuntimeError(f"sampling_loc shape {tuple(sampling_loc.shape)} != {(B, Nq, H, L, P, 2)}")
    if attn_weight.shape != (B, Nq, H, L, P):
        raise RuntimeError(f"attn_weight shape {tuple(attn_weight.shape)} != {(B, Nq, H, L, P)}")
    if level_start_index.numel() != L:
        raise RuntimeError("level_start_index must have one entry per level")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise RuntimeError("spatial_shapes and level_start_index must be int64 (reference reads data<int64_t>)")
    return B, S, H, D, L, Nq, P


def _same_dtype(value, sampling_loc, attn_weight):
    """The reference reads all three through one scalar_t (.cu:67-73).  softmax under
    autocast may hand an fp32 attn_weight next to fp16 values; cast instead of reading
    garbage (SURVEY.md section 8a "Dtype flow")."""
    _require(value.dtype in _DTYPE_CODE, f"unsupported dtype {value.dtype}")
    if sampling_loc.dtype != value.dtype:
        sampling_loc = sampling_loc.to(value.dtype)
    if attn_weight.dtype != value.dtype:
        attn_weight = attn_weight.to(value.dtype)
    return sampling_loc, attn_weight


def _aligned(t, nbytes=16):
    return t if t.data_ptr() % nbytes == 0 else t.clone(memory_format=torch.contiguous_format)


def _stream(device):
    return torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch.cuda.current_device())


class _on_device:
    """torch.cuda.device(dev) without the context-manager cost when dev is already current."""
    __slots__ = ("ctx",)

    def __init__(self, device):
        idx = device.index
        self.ctx = None if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


# host-only size queries of the library, memoised (they depend on the numbers only; the level table enters
# through the identity + version of its host copy)
_ws_cache = {}


# bench.py sets this to a list to time individual kernels with HIP events recorded on
# the very stream the kernel is launched on: entries are (kernel_name, start, end).
_event_log = None


def _launch(name, device, fn, *args):
    if _event_log is None:
        return fn(*args)
    st = torch.cuda.current_stream(device)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record(st)
    status = fn(*args)
    t1.record(st)
    _event_log.append((name, t0, t1))
    return status


# ---- level tables the shim has never seen are checked ON THE DEVICE by the backward (no device->host copy per
# call).  A table the sorted backward cannot serve (overlapping / out-of-range / >= 65536-wide levels) gets the
# reference's float-atomic scatter IN THE SAME CALL (csrc/msda_bwd_refused.hip: three launches that read the plan's
# verdict on the device and return at once for every ordinary table), so grad_value is what the reference computes
# either way.  The plan also reports such a table through this word -- one int32 of pinned, device-visible host
# memory per device -- and the next call into the shim (or check_level_table_status()) turns it into ONE warning:
# the fallback is slow, and a caller that builds such tables should register them.  No device trap, no exception.
_status_words = {}
_BAD_TABLE_MSG = ("ms_deform_attn_backward: an earlier call on this device was handed a level table with overlapping, "
                  "out-of-range or >= 65536-wide levels that the shim had never seen; its grad_value came from the "
                  "float-atomic fallback (correct, as the reference computes it, but slow).  Register such tables "
                  "(MultiScaleDeformableAttention.register_level_tables): they then take the float-atomic path directly")
_bad_table_warned = False


def _status_word(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    w = _status_words.get(idx)
    if w is None:
        t = torch.zeros(1, dtype=torch.int32).pin_memory()
        w = _status_words[idx] = (t, t.numpy())
    return w


def check_level_table_status(device=None, synchronize=False):
    """True (and, once per process, a RuntimeWarning) if a backward on ``device`` (default: every device used so far)
    met a level table the device-side check refused since the last time this was asked; the flag is consumed.
    ``synchronize=True`` waits for the device first (a test's use; the op itself only looks at what has already
    arrived).  The gradients of such a call are correct: it took the float-atomic fallback."""
    global _bad_table_warned
    if synchronize:
        torch.cuda.synchronize(device)
    words = _status_words.values() if device is None else [_status_word(torch.device(device))]
    seen = False
    for _, flag in words:
        if flag[0]:
            flag[0] = 0
            seen = True
    if seen and not _bad_table_warned:
        _bad_table_warned = True
        import warnings
        warnings.warn(_BAD_TABLE_MSG, RuntimeWarning, stacklevel=2)
    return seen


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                           im2col_step):
    """Reference: ms_deform_attn_cuda_forward, src/cuda/ms_deform_attn_cuda.cu:21-81."""
    _require(isinstance(value, torch.Tensor) and value.is_cuda, "Not implemented on the CPU")
    if _status_words:
        check_level_table_status()
    _validate([("value", value), ("spatial_shapes", spatial_shapes),
               ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
               ("attn_weight", attn_weight)], value)
    B, S, H, D, L, Nq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    step = min(B, int(im2col_step)) if B > 0 else 1
    _require(step > 0 and B % step == 0, f"batch({B}) must divide im2col_step({step})")
    sampling_loc, attn_weight = _same_dtype(value, sampling_loc, attn_weight)
    value = _aligned(value)
    out = torch.empty((B, Nq, H * D), dtype=value.dtype, device=value.device)
    code = _DTYPE_CODE[value.dtype]
    dims = (B, S, H, D, L, Nq, P)
    with _on_device(value.device):
        stream = _stream(value.device)
        status = _launch(
            "msda_fwd", value.device, _lib.mmfs_msda_forward_flags, code, value.data_ptr(), spatial_shapes.data_ptr(),
            level_start_index.data_ptr(), sampling_loc.data_ptr(), attn_weight.data_ptr(),
            out.data_ptr(), *dims, _FWD_FLAGS[_fwd_algo], stream)
    _check(status, "ms_deform_attn_forward")
    return out


# tests / measurements: which formulation of the forward runs (include/mmfs_msda.h, mmfs_msda_forward_flags):
# "auto" | "gather" (csrc/msda_fwd.hip) | "lds" (csrc/msda_fwd_mma.hip; unsupported shapes raise)
_fwd_algo = "auto"
_FWD_FLAGS = {"auto": 0, "gather": 1, "lds": 2, "slices": 4, "waves": 8}        # "slices": csrc/msda_fwd_q8.hip; "waves": msda_fwd_wq.hip


# flags of mmfs_msda_backward (include/mmfs_msda.h)
_BWD_CANONICAL_LEVELS = 1
_BWD_FORCE_ATOMIC = 2
_BWD_DENSE_TAPS = 4
_BWD_LAZY_ZERO_ATTN = 16
_BWD_DEVICE_CHECKED_LEVELS = 32
_BWD_TAPS_ROW_GATHER = 64
_BWD_TAPS_LDS_LEVELS = 128
_E_UNSUPPORTED = -5

# tests / measurements: which formulation computes grad_loc / grad_attn (include/mmfs_msda.h):
# "auto" | "gather" (csrc/msda_bwd.hip + msda_dense.hip) | "lds" (csrc/msda_taps_mma.hip; unsupported shapes raise) |
# "sorted" (csrc/msda_bwd_taps_sorted.hip: from the grad_value sort's records -- parity-green, measured slower than the gather
# kernels on every shipped geometry, DESIGN.md 4.2c: opt-in; arguments mmfs_msda_backward_sorted does not take raise).
# MMFS_TAPS_ROUTE=sorted in the environment takes it wherever it applies, for a whole process (measurements).
_taps_algo = "auto"
_taps_prefer_sorted = os.environ.get("MMFS_TAPS_ROUTE") == "sorted"      # wherever it applies, silently not elsewhere
route_counts = {"sorted": 0}                  # backward calls that took mmfs_msda_backward_sorted (tests: "the route ran")
_TAPS_FLAGS = {"auto": 0, "gather": _BWD_TAPS_ROW_GATHER, "lds": _BWD_TAPS_LDS_LEVELS, "sorted": 0}

# tests / measurements: "auto" | "atomic" (force the float-atomic path)
_bwd_algo = "auto"
# tests / measurements: False keeps grad_loc / grad_attn of every level on the row-gather kernel
# (True: levels of at most min(256, 64*P) pixels take the matrix-core product, csrc/msda_dense.hip)
_hybrid = os.environ.get("MMFS_HYBRID", "1") != "0"

# The backward's two halves are independent (grad_loc / grad_attn read value; grad_value does not)
# and can be launched on two streams (a side stream that forks from and joins the caller's stream
# inside the call).  Measured on MI355X: both halves already fill the chip, so they mostly
# time-share: -1.2 % step time at the north-star shape, -3 % at the SD geometry, but +15 % at
# config 1, where the step is host-bound and the extra stream calls cost more than they hide.
# Off by default; MMFS_BWD_OVERLAP=1 turns it on.
_bwd_overlap = os.environ.get("MMFS_BWD_OVERLAP", "0") == "1"
_side_streams = {}


def _side_stream(device):
    st = _side_streams.get(device)
    if st is None:
        st = _side_streams[device] = torch.cuda.Stream(device=device)
    return st


class _fork:
    """with _fork(device) as side: launches inside go to the side stream (None: overlap off)."""

    def __init__(self, device):
        self.device, self.side, self.ctx = device, None, None

    def __enter__(self):
        if _bwd_overlap:
            self.main = torch.cuda.current_stream(self.device)
            self.side = _side_stream(self.device)
            self.side.wait_stream(self.main)
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)


# stage bits of mmfs_msda_backward_sorted (include/mmfs_msda.h)
_SRT_BWD_STAGES = (("msda_bwd_value_prepare", 1), ("msda_bwd_value_sort", 2), ("msda_bwd_taps", 4), ("msda_bwd_value_reduce", 8))
_SRT_BWD_ALL = 15

# stage bits of mmfs_msda_backward_hybrid (include/mmfs_msda.h)
_HYB_BWD_STAGES = (("msda_bwd_taps", 1), ("msda_bwd_taps_coarse", 2), ("msda_bwd_value_prepare", 4),
                   ("msda_bwd_value_sort", 8), ("msda_bwd_value_reduce", 16))
_HYB_BWD_ALL = 31


def levels_are_canonical(spatial_shapes, level_start_index, S):
    """True when start[l] == sum_{k<l} H_k*W_k and sum_l H_l*W_l == S (the packing every
    caller in the reference builds: modeling_llama_mmfs.py:303-305, sd_mmfs.py:35-37).

    The tables live in device memory (reference API), so this query costs one small
    device->host copy the first time a pair of tensor objects is asked about; the answer is
    cached on the tensor object (keyed by the in-place version counters).  The op itself never
    asks: it uses the answer when it is there (``mmfs_amd.levels.make_level_tables`` pre-seeds
    it) and lets the library check the table on the device otherwise."""
    return _level_info(spatial_shapes, level_start_index, S)[0]


def _level_key(spatial_shapes, level_start_index, S):
    return (spatial_shapes._version, level_start_index._version, level_start_index.data_ptr(), int(S))


def _level_info(spatial_shapes, level_start_index, S, sync=True):
    """(canonical, host copy of spatial_shapes, host copy of level_start_index) -- the host
    copies are contiguous int64 numpy arrays (what the hybrid entry points take).  Cached on
    the spatial_shapes tensor object; with sync=False returns None instead of copying."""
    key = _level_key(spatial_shapes, level_start_index, S)
    cached = getattr(spatial_shapes, "_mmfs_canonical", None)
    if cached is not None and cached[0] == key:
        return cached[1:]
    if not sync:
        return None
    sh = spatial_shapes.detach().cpu().contiguous()
    st = level_start_index.detach().cpu().reshape(-1).contiguous()
    return _seed_level_info(spatial_shapes, key, sh, st, S)


def _seed_level_info(spatial_shapes, key, sh, st, S):
    px = sh[:, 0] * sh[:, 1]
    canon = bool((sh >= 0).all()) and bool((sh < 65536).all()) and \
        bool(torch.equal(st, px.cumsum(0) - px)) and int(px.sum()) == int(S)
    info = (canon, np.ascontiguousarray(sh.numpy(), dtype=np.int64),
            np.ascontiguousarray(st.numpy(), dtype=np.int64))
    try:
        spatial_shapes._mmfs_canonical = (key,) + info
    except Exception:
        pass
    return info


def register_level_tables(spatial_shapes, level_start_index, S, host_shapes=None, host_start=None):
    """Tell the shim what a pair of device level tables contains (one device->host copy now, or
    none when the host copies are passed), so that neither pass ever has to look: enables the
    pixel-stationary backward and the hybrid routing of small levels from the first call."""
    key = _level_key(spatial_shapes, level_start_index, S)
    if host_shapes is None or host_start is None:
        return _level_info(spatial_shapes, level_start_index, S)
    sh = torch.as_tensor(host_shapes, dtype=torch.long).reshape(-1, 2).contiguous()
    st = torch.as_tensor(host_start, dtype=torch.long).reshape(-1).contiguous()
    return _seed_level_info(spatial_shapes, key, sh, st, S)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                            grad_output, im2col_step, lazy_zero_attn=False):
    """Reference: ms_deform_attn_cuda_backward, src/cuda/ms_deform_attn_cuda.cu:84-166.
    Returns [grad_value, grad_sampling_loc, grad_attn_weight] shaped and typed like the
    corresponding inputs.  ``lazy_zero_attn`` (an addition, MMFS_BWD_LAZY_ZERO_ATTN): the caller
    never reads grad_attn_weight where attn_weight is exactly 0 (MMFS's masked softmax multiplies it
    by the weight), so those entries may come back as 0 without their value rows being read."""
    _require(isinstance(value, torch.Tensor) and value.is_cuda, "Not implemented on the CPU")
    if _status_words:
        check_level_table_status()
    _validate([("value", value), ("spatial_shapes", spatial_shapes),
               ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
               ("attn_weight", attn_weight), ("grad_output", grad_output)], value)
    B, S, H, D, L, Nq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight)
    _require(grad_output.numel() == B * Nq * H * D, "grad_output must be [B, Nq, H*D]")
    step = min(B, int(im2col_step)) if B > 0 else 1
    _require(step > 0 and B % step == 0, f"batch({B}) must divide im2col_step({step})")
    loc_dtype, attn_dtype = sampling_loc.dtype, attn_weight.dtype
    sampling_loc, attn_weight = _same_dtype(value, sampling_loc, attn_weight)
    if grad_output.dtype != value.dtype:
        grad_output = grad_output.to(value.dtype)
    value, grad_output = _aligned(value), _aligned(grad_output)

    dt = value.dtype
    code = _DTYPE_CODE[dt]
    dims = (B, S, H, D, L, Nq, P)
    flags = (_BWD_LAZY_ZERO_ATTN if lazy_zero_attn else 0) | _TAPS_FLAGS[_taps_algo]
    info = None
    if _bwd_algo == "atomic":
        flags |= _BWD_FORCE_ATOMIC
    else:
        # What the level tables contain is only looked up, never fetched: tables that came from
        # mmfs_amd.levels.make_level_tables / register_level_tables (or were seen by levels_are_canonical)
        # carry their host copy; for any other pair -- the reference's own callers build fresh tensors on
        # every call -- the library checks the table on the device (no device->host copy, no sync).
        info = _level_info(spatial_shapes, level_start_index, S, sync=False)
        if info is None:
            flags |= _BWD_DEVICE_CHECKED_LEVELS
        elif info[0]:
            flags |= _BWD_CANONICAL_LEVELS
    grad_value = torch.empty(value.shape, dtype=dt, device=value.device)
    grad_loc = torch.empty(sampling_loc.shape, dtype=dt, device=value.device)
    grad_attn = torch.empty(attn_weight.shape, dtype=dt, device=value.device)
    with _on_device(value.device):
        stream = _stream(value.device)
        status = _E_UNSUPPORTED
        hyb_bytes = 0
        # opt-in: grad_loc / grad_attn from the grad_value sort's records (csrc/msda_bwd_taps_sorted.hip): a host-verified
        # canonical table, 16-bit storage, shapes the sort's kept scan takes
        srt_bytes = 0
        if (_taps_algo == "sorted" or (_taps_prefer_sorted and _taps_algo == "auto")) and _bwd_algo != "atomic" and info is not None and (flags & _BWD_CANONICAL_LEVELS) and code in (1, 2):
            skey = ("sorted", code, dims, flags)
            srt_bytes = _ws_cache.get(skey)
            if srt_bytes is None:
                if len(_ws_cache) > 4096:
                    _ws_cache.clear()
                srt_bytes = _ws_cache[skey] = _lib.mmfs_msda_backward_sorted_workspace_bytes(code, *dims, flags)
        _require(srt_bytes > 0 or _taps_algo != "sorted", "taps algo 'sorted': mmfs_msda_backward_sorted does not apply to these arguments")
        if srt_bytes > 0:
            route_counts["sorted"] += 1
            ws = torch.empty(srt_bytes, dtype=torch.uint8, device=value.device)
            hs = info[1]
            blocks4 = int((((hs[:, 0] + 3) // 4) * ((hs[:, 1] + 3) // 4))[(hs[:, 0] > 0) & (hs[:, 1] > 0)].sum())
            sargs = (code, value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                     sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                     grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(), ws.data_ptr(), srt_bytes,
                     *dims, blocks4, flags)
            if _event_log is None:
                status = _lib.mmfs_msda_backward_sorted(*sargs, _SRT_BWD_ALL, stream)
            else:
                for name, bit in _SRT_BWD_STAGES:
                    status = _launch(name, value.device, _lib.mmfs_msda_backward_sorted, *sargs, bit, _stream(value.device))
                    if status != 0:
                        break
            _check(status, "ms_deform_attn_backward")
            if loc_dtype != dt:
                grad_loc = grad_loc.to(loc_dtype)
            if attn_dtype != dt:
                grad_attn = grad_attn.to(attn_dtype)
            return [grad_value, grad_loc, grad_attn]
        if _hybrid and info is not None and (flags & _BWD_CANONICAL_LEVELS) and code in (1, 2):
            flags |= _BWD_DENSE_TAPS
            hs, hst = info[1].ctypes.data, info[2].ctypes.data
            # (keyed by the table's CONTENT: a freed host copy's address can come back with another table)
            key = (code, dims, flags, info[1].tobytes(), info[2].tobytes())
            hyb_bytes = _ws_cache.get(key)
            if hyb_bytes is None:
                if len(_ws_cache) > 4096:
                    _ws_cache.clear()
                hyb_bytes = _ws_cache[key] = _lib.mmfs_msda_backward_hybrid_workspace_bytes(code, hs, hst, *dims, flags)
        if hyb_bytes > 0:
            # small levels on the matrix cores, the others through the gather / sort / reduce kernels
            ws = torch.empty(hyb_bytes, dtype=torch.uint8, device=value.device)
            args = (code, value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), hs, hst,
                    sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                    grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(), ws.data_ptr(), hyb_bytes,
                    *dims, flags)
            fkey = ("fused", code, dims, flags)
            fused = _ws_cache.get(fkey)
            if fused is None:
                fused = _ws_cache[fkey] = _lib.mmfs_msda_backward_taps_fused(code, *dims, flags)
            sorted_levels = int(((info[1][:, 0] > 0) & (info[1][:, 1] > 0)).sum())

            def run_stages(stages):
                st = 0
                if fused:               # one kernel does every level: the dense stage has nothing to launch
                    stages = tuple(sb for sb in stages if sb[0] != "msda_bwd_taps_coarse")
                if sorted_levels == 0:
                    stages = tuple(sb for sb in stages if sb[0] not in ("msda_bwd_value_prepare", "msda_bwd_value_sort",
                                                                        "msda_bwd_value_reduce"))
                if _event_log is None:
                    bits = sum(bit for _, bit in stages)
                    return _lib.mmfs_msda_backward_hybrid(*args, bits, _stream(value.device))
                for name, bit in stages:
                    st = _launch(name, value.device, _lib.mmfs_msda_backward_hybrid, *args, bit, _stream(value.device))
                    if st != 0:
                        break
                return st
            if _event_log is None and not _bwd_overlap:
                status = _lib.mmfs_msda_backward_hybrid(*args, _HYB_BWD_ALL, stream)    # one call, one stream
            else:
                fork = _fork(value.device)
                with fork:
                    status = run_stages(_HYB_BWD_STAGES[:2])         # grad_loc / grad_attn (side stream)
                if status == 0:
                    status = run_stages(_HYB_BWD_STAGES[2:])         # grad_value
                fork.join()
        ws_bytes = 0
        if hyb_bytes == 0:
            key = (code, dims, flags)
            ws_bytes = _ws_cache.get(key)
            if ws_bytes is None:
                if len(_ws_cache) > 4096:
                    _ws_cache.clear()
                ws_bytes = _ws_cache[key] = _lib.mmfs_msda_backward_workspace_bytes(code, *dims, flags)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=value.device) if ws_bytes else None
        ws_ptr = ws.data_ptr() if ws is not None else None
        if hyb_bytes == 0 and (flags & (_BWD_CANONICAL_LEVELS | _BWD_DEVICE_CHECKED_LEVELS)) and _event_log is None and not _bwd_overlap:
            pass                                 # the library's own sequence below: one call (same kernels)
        elif hyb_bytes == 0 and (flags & _BWD_CANONICAL_LEVELS):
            # sorted backward, stage by stage (so each kernel can be timed / the halves can overlap)
            fork = _fork(value.device)
            with fork:
                status = _launch("msda_bwd_taps", value.device, _lib.mmfs_msda_backward_taps, code,
                                 value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                                 sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                                 grad_loc.data_ptr(), grad_attn.data_ptr(), *dims, _stream(value.device))
            if status == 0:
                status = _launch("msda_bwd_value_prepare", value.device, _lib.mmfs_msda_backward_value_prepare,
                                 code, sampling_loc.data_ptr(), attn_weight.data_ptr(), ws_ptr, ws_bytes,
                                 *dims, stream)
            if status == 0:
                status = _launch("msda_bwd_value_sort", value.device, _lib.mmfs_msda_backward_value_sort, code,
                                 spatial_shapes.data_ptr(), level_start_index.data_ptr(), ws_ptr, ws_bytes,
                                 *dims, stream)
            if status == 0:
                status = _launch("msda_bwd_value_reduce", value.device, _lib.mmfs_msda_backward_value_reduce,
                                 code, grad_output.data_ptr(), grad_value.data_ptr(), ws_ptr, ws_bytes,
                                 *dims, stream)
            fork.join()
        if status == _E_UNSUPPORTED:
            # the library's own sequence: taps + sort + reduce when the level table is canonical and
            # the head width has a vector path, else (odd head width, fp64, gapped or overlapping
            # levels) float-atomic accumulation (needs an fp32 scratch for 16-bit storage)
            word = _status_word(value.device)[0].data_ptr() if (flags & _BWD_DEVICE_CHECKED_LEVELS) else None
            status = _launch("msda_bwd_atomic", value.device, _lib.mmfs_msda_backward_checked, code,
                             value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                             sampling_loc.data_ptr(), attn_weight.data_ptr(), grad_output.data_ptr(),
                             grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
                             ws_ptr, ws_bytes, *dims, flags, word, stream)
        _check(status, "ms_deform_attn_backward")
    if loc_dtype != dt:
        grad_loc = grad_loc.to(loc_dtype)
    if attn_dtype != dt:
        grad_attn = grad_attn.to(attn_dtype)
    return [grad_value, grad_loc, grad_attn]
