"""CPU-side checks of the drop-in boundary: libmmfs_msda.so loads, exports every
symbol include/mmfs_msda.h declares, validates arguments without touching a GPU, and
the Python shim keeps the reference's error behaviour (no CPU path:
ops/src/ms_deform_attn.h:29-38)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mmfs_msda.h")
LIB = os.path.join(ROOT, "mm-interleaved_amd", "libmmfs_msda.so")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmfs_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = declared_functions()
    for n in ("mmfs_msda_forward", "mmfs_msda_backward", "mmfs_msda_backward_taps",
              "mmfs_msda_backward_value", "mmfs_msda_backward_value_prepare", "mmfs_msda_backward_value_run",
              "mmfs_msda_backward_value_sort", "mmfs_msda_backward_value_reduce",
              "mmfs_msda_backward_workspace_bytes",
              "mmfs_msda_cast_from_f32", "mmfs_msda_abi_version", "mmfs_msda_status_string"):
        assert n in names


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "run __graft_entry__.build() / make -C mm-interleaved_amd/csrc"
    lib = ctypes.CDLL(LIB)
    for n in declared_functions():
        assert hasattr(lib, n), f"{n} declared in include/mmfs_msda.h but not exported"
    lib.mmfs_msda_abi_version.restype = ctypes.c_int
    want = int(re.search(r"#define\s+MMFS_MSDA_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert lib.mmfs_msda_abi_version() == want


def test_argument_errors_do_not_need_a_gpu():
    lib = ctypes.CDLL(LIB)
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    lib.mmfs_msda_forward.restype = ctypes.c_int
    lib.mmfs_msda_forward.argtypes = [ctypes.c_int] + [vp] * 6 + [i64] * 7 + [vp]
    lib.mmfs_msda_status_string.restype = ctypes.c_char_p
    lib.mmfs_msda_status_string.argtypes = [ctypes.c_int]
    null = [None] * 6
    assert lib.mmfs_msda_forward(9, *null, 1, 1, 1, 1, 1, 1, 1, None) == -1       # dtype
    assert lib.mmfs_msda_forward(0, *null, -1, 1, 1, 1, 1, 1, 1, None) == -2      # dims
    assert lib.mmfs_msda_forward(0, *null, 1, 1, 1, 1, 1, 1, 1, None) == -3       # null
    assert lib.mmfs_msda_forward(0, *null, 0, 1, 1, 1, 1, 1, 1, None) == 0        # empty batch
    # the LDS-resident formulation exists for 16-bit storage and heads of 64 / 128 channels only
    lib.mmfs_msda_forward_flags.restype = ctypes.c_int
    lib.mmfs_msda_forward_flags.argtypes = [ctypes.c_int] + [vp] * 6 + [i64] * 7 + [ctypes.c_uint, vp]
    fake = [ctypes.c_void_p(4096)] * 6                                             # aligned, never dereferenced
    assert lib.mmfs_msda_forward_flags(0, *fake, 1, 64, 1, 128, 1, 1, 1, 2, None) == -5     # fp32
    assert lib.mmfs_msda_forward_flags(2, *fake, 1, 64, 1, 32, 1, 1, 1, 2, None) == -5      # D = 32
    assert lib.mmfs_msda_forward_flags(2, *fake, 1, 64, 1, 128, 65, 1, 1, 2, None) == -5    # L > 64
    assert b"dtype" in lib.mmfs_msda_status_string(-1)
    assert lib.mmfs_msda_status_string(-99) is not None


def test_bank_entry_points_validate_on_the_host():
    """mmfs_bank_gather / mmfs_bank_scatter: argument errors and empty problems return before any launch."""
    lib = ctypes.CDLL(LIB)
    i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
    for f in (lib.mmfs_bank_gather, lib.mmfs_bank_scatter):
        f.restype = ci
        f.argtypes = [ci, ci, vp, vp, vp, vp, i64, i64, i64, vp]
    ptrs = (vp * 2)(0x1000, 0x2000)                   # never dereferenced on these paths
    hw = (i64 * 2)(64, 16)
    P, H = ctypes.cast(ptrs, vp), ctypes.cast(hw, vp)
    for f in (lib.mmfs_bank_gather, lib.mmfs_bank_scatter):
        assert f(3, 2, P, H, None, None, 4, 8, 4, None) == -1          # fp64 is not a bank dtype
        assert f(2, 0, P, H, None, None, 4, 8, 4, None) == -5          # no levels
        assert f(2, 9, P, H, None, None, 4, 8, 4, None) == -5          # more than 8 levels
        assert f(2, 2, None, H, None, None, 4, 8, 4, None) == -3       # no level table
        assert f(2, 2, P, H, None, None, 4, -8, 4, None) == -2         # negative width
        assert f(2, 2, P, ctypes.cast((i64 * 2)(64, 0), vp), None, None, 4, 8, 4, None) == -2   # empty level
        assert f(2, 2, P, H, None, None, 4, 8, 4, None) == -3          # non-empty problem, null tensors
    assert lib.mmfs_bank_gather(2, 2, P, H, None, None, 4, 8, 0, None) == 0      # no slots
    assert lib.mmfs_bank_gather(2, 2, P, H, None, None, 4, 8, 70000, None) == -2  # slot axis exceeds the grid
    assert lib.mmfs_bank_scatter(2, 2, P, H, None, None, 0, 8, 4, None) == 0     # no images


def test_shim_has_the_reference_surface_and_no_cpu_path():
    import MultiScaleDeformableAttention as MSDA
    assert callable(MSDA.ms_deform_attn_forward) and callable(MSDA.ms_deform_attn_backward)
    v = torch.zeros(1, 4, 1, 8)
    sh = torch.tensor([[2, 2]]); st = torch.tensor([0])
    loc = torch.zeros(1, 1, 1, 1, 1, 2); attn = torch.zeros(1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="CPU"):
        MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 1)
    with pytest.raises(RuntimeError, match="CPU"):
        MSDA.ms_deform_attn_backward(v, sh, st, loc, attn, torch.zeros(1, 1, 8), 1)


def test_autograd_function_surface():
    from mmfs_amd.functions import MSDeformAttnFunction, ms_deform_attn_core_pytorch
    assert hasattr(MSDeformAttnFunction, "apply")
    with pytest.raises(RuntimeError):
        ms_deform_attn_core_pytorch(torch.zeros(1, 4, 1, 8), [(2, 2)], torch.zeros(1, 1, 1, 1, 1, 2),
                                    torch.zeros(1, 1, 1, 1, 1))


def test_backward_workspace_query_is_host_only():
    lib = ctypes.CDLL(LIB)
    i64 = ctypes.c_int64
    f = lib.mmfs_msda_backward_workspace_bytes
    f.restype = i64
    f.argtypes = [ctypes.c_int] + [i64] * 7 + [ctypes.c_uint]
    dims = (8, 5440, 8, 128, 4, 4096, 4)
    pts = 8 * 4096 * 8 * 4 * 4
    # bf16, canonical levels: re-packed loc/attn + level cursors + plan + per-pixel / per-cell run
    # table + sorted records (pixel-stationary: 32 B per sample; block-stationary: 16 B) + the queue
    # and fp32 partial sums for long lists (block-stationary) + the work-item queue and fp32 partial
    # tiles of the matrix-core reduce (4x4 pixels x D channels each, bounded by twice the expected
    # record visits / records per item: 105 MB here)
    base = 3 * pts * 2 + 8 * 8 * 4 * 4 + 8 * 8 * 5440 * 8 + pts * 4 * 8
    assert base < f(2, *dims, 1) <= base + (176 << 20)
    assert f(2, *dims, 0) == 8 * 5440 * 8 * 128 * 4          # unknown level table: fp32 image for atomics
    assert f(2, *dims, 3) == 8 * 5440 * 8 * 128 * 4          # forced atomic
    assert f(0, *dims, 0) == 0                               # fp32 atomics accumulate in grad_value itself
    assert f(2, 8, 5440, 8, 24, 4, 4096, 4, 1) > 0           # D=24: no vector path


def test_canonical_level_table_detection_and_cache():
    import MultiScaleDeformableAttention as MSDA
    sh = torch.tensor([[4, 4], [2, 2]]); st = torch.tensor([0, 16])
    assert MSDA.levels_are_canonical(sh, st, 20)
    assert getattr(sh, "_mmfs_canonical")[1] is True
    assert not MSDA.levels_are_canonical(sh, st, 21)          # S mismatch (gap at the end)
    assert not MSDA.levels_are_canonical(sh, torch.tensor([0, 15]), 20)   # overlap
    st[1] = 17                                               # in-place edit invalidates the cache
    assert not MSDA.levels_are_canonical(sh, st, 20)


def test_hybrid_workspace_queries_are_host_only():
    """mmfs_msda_backward_hybrid routes grad_loc / grad_attn of levels of <= 256 pixels to the matrix
    cores when the caller hands over a HOST copy of the level table; its workspace query computes on
    the host."""
    import numpy as np
    lib = ctypes.CDLL(LIB)
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    bw = lib.mmfs_msda_backward_hybrid_workspace_bytes
    bw.restype = i64
    bw.argtypes = [ctypes.c_int, vp, vp] + [i64] * 7 + [ctypes.c_uint]
    plain = lib.mmfs_msda_backward_workspace_bytes
    plain.restype = i64
    plain.argtypes = [ctypes.c_int] + [i64] * 7 + [ctypes.c_uint]
    sh = np.array([[64, 64], [32, 32], [16, 16], [8, 8]], dtype=np.int64)
    st = np.array([0, 4096, 5120, 5376], dtype=np.int64)
    dims = (8, 5440, 8, 128, 4, 4096, 4)
    hs, hst = sh.ctypes.data, st.ctypes.data
    CANON, TAPS = 1, 4
    base = plain(2, *dims, CANON)
    assert bw(2, hs, hst, *dims, CANON | TAPS) >= base     # dense dot products need no extra scratch
    assert bw(2, None, None, *dims, CANON | TAPS) == 0     # no host table: plain entry point
    assert bw(2, hs, hst, *dims, CANON) == 0               # dense part not asked for
    assert bw(2, hs, hst, *dims, TAPS) == 0                # not canonical: atomic path only
    assert bw(0, hs, hst, *dims, CANON | TAPS) == 0        # fp32 storage
    big = np.array([[64, 64], [32, 32]], dtype=np.int64)   # no level small enough
    assert bw(2, big.ctypes.data, np.array([0, 4096], dtype=np.int64).ctypes.data, 8, 5120, 8, 128, 2, 4096, 4, CANON | TAPS) == 0


def test_sample_forward_argument_errors_return_before_any_launch():
    """mmfs_sample_forward (plan -> sampler in one kernel): dtype / dims / unsupported shapes / NULL pointers are
    answered on the host."""
    lib = ctypes.CDLL(LIB)
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    f = lib.mmfs_sample_forward
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_int] + [vp] * 12 + [i64] * 11 + [vp]
    null = [None] * 12
    #            N  S     Lq  H  D   L  P  n  M  Lr Nr
    dims = [2, 1344, 16, 16, 64, 3, 8, 1, 8, 1, 1]
    assert f(9, *null, *dims, None) == -1                               # dtype
    assert f(2, *null, *(dims[:6] + [5] + dims[7:]), None) == -5        # P = 5: unsupported by the plan
    assert f(2, *null, *(dims[:6] + [16] + dims[7:]), None) == -5       # P = 16: two-kernel path
    assert f(2, *null, *(dims[:4] + [24] + dims[5:]), None) == -5       # D = 24: no 16-byte vector rows
    assert f(2, *null, *dims, None) == -3                               # NULL tensors
    assert f(2, *null, *([0] + dims[1:]), None) == 0                    # empty batch


def test_environment_knobs_are_one_table():
    """Every ``MMFS_*`` variable the library reads is an entry of csrc/msda_env.hip's table (read once; ``mmfs_env_reload``
    reads it again) -- no kernel launcher calls getenv() -- and INTEGRATION.md names each of them."""
    import glob
    csrc = os.path.join(os.path.dirname(HEADER), "..", "mm-interleaved_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        if os.path.basename(path) == "msda_env.hip":
            continue
        text = open(path).read()
        assert "getenv(" not in text, path
    lib = ctypes.CDLL(LIB)
    lib.mmfs_env_knob.restype = ctypes.c_int
    lib.mmfs_env_knob.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_char_p)] * 3
    lib.mmfs_env_reload.restype = None
    names, i = [], 0
    name, doc, val = ctypes.c_char_p(), ctypes.c_char_p(), ctypes.c_char_p()
    while lib.mmfs_env_knob(i, ctypes.byref(name), ctypes.byref(doc), ctypes.byref(val)) == 0:
        assert name.value.startswith(b"MMFS_") and len(doc.value) > 10
        names.append(name.value.decode())
        i += 1
    assert len(names) == len(set(names)) >= 28
    integration = open(os.path.join(os.path.dirname(HEADER), "..", "INTEGRATION.md")).read()
    for n in names:
        assert n in integration, n
    # the table follows the environment only when told to
    probe = "MMFS_NORM_BWD_GRID"
    k = names.index(probe)
    old = os.environ.get(probe)
    try:
        os.environ[probe] = "77"
        lib.mmfs_env_knob(k, None, None, ctypes.byref(val))
        assert val.value != b"77"
        lib.mmfs_env_reload()
        lib.mmfs_env_knob(k, None, None, ctypes.byref(val))
        assert val.value == b"77"
    finally:
        if old is None:
            os.environ.pop(probe, None)
        else:
            os.environ[probe] = old
        lib.mmfs_env_reload()
