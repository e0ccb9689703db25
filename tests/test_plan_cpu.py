"""CPU test of the sorted backward's tiling rules (csrc/msda_bwd_block.h: level_tiling, tile_local_blocks,
block_is_tile_local), through the library's host-only self-check: for every level extent and tile count,
the sort tiles cover each cell exactly once within the LDS counters' capacity, and every 4x4 block of the
matrix-core reduce is planned exactly once -- by the one sort tile that holds its five cell rows or, on a seam
between two tiles, by the slice's last workgroup.  (The GPU side of the same rules:
tests/test_op_gpu.py::test_cell_sort_routes_match_oracle.)"""
import ctypes
import itertools
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mm-interleaved_amd", "libmmfs_msda.so")


@pytest.fixture(scope="module")
def selfcheck():
    assert os.path.exists(LIB), "run __graft_entry__.build() / make -C mm-interleaved_amd/csrc"
    lib = ctypes.CDLL(LIB)
    fn = lib.mmfs_msda_plan_selfcheck
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
    return fn


def test_levels_of_the_reference_configs(selfcheck):
    # BASELINE.json configs: 64/32/16/8 pyramids, the 16x16 ViT map, LLM 32/16/8 levels
    for hw in (64, 32, 16, 8, 4, 2, 1):
        for nt in (1, 2, 3, 4, 8, 32, 256):
            assert selfcheck(hw, hw, nt) == 0, (hw, nt)


def test_every_small_extent_and_tile_count(selfcheck):
    bad = [(h, w, nt) for h, w, nt in itertools.product(range(1, 41), range(1, 41), (1, 2, 3, 5, 7, 16))
           if selfcheck(h, w, nt) != 0]
    assert not bad, bad[:10]


def test_large_and_degenerate_extents(selfcheck):
    cases = [(1, 65535), (65535, 1), (2, 5119), (2, 5120), (3, 5121), (71, 71), (72, 71), (255, 257), (1000, 37),
             (37, 1000), (4, 20000), (20000, 4), (511, 513), (5119, 2), (5120, 5120)]
    for (h, w), nt in itertools.product(cases, (1, 2, 6, 64, 256)):
        assert selfcheck(h, w, nt) == 0, (h, w, nt)


def test_levels_without_tiles_say_so(selfcheck):
    for h, w in ((0, 5), (5, 0), (-1, 3), (65536, 2), (2, 65536)):
        assert selfcheck(h, w, 1) == -1, (h, w)


def test_token_rows_hands_over_column_ranges_only_when_the_vector_loads_allow():
    """``mmfs_plan_func._token_rows`` (host logic of ``mmfs_sample_forward_heads``): a column range of a wider
    row-major matrix is passed as it lies -- with the distance between two token rows -- when its rows start on the
    kernel's vector boundary; anything else is copied (distance 0 = packed)."""
    import torch
    from mmfs_amd.functions.mmfs_plan_func import _token_rows
    both = torch.zeros(3, 5, 64 + 128 + 16, dtype=torch.bfloat16)          # 416-byte rows: whole 32-byte vectors
    off, att = both[..., :64], both[..., 64:192]
    t, ld = _token_rows(off, 16)
    assert t.data_ptr() == off.data_ptr() and ld == 208
    t, ld = _token_rows(att, 8)
    assert t.data_ptr() == att.data_ptr() and ld == 208
    t, ld = _token_rows(both[..., 1:65], 16)                 # rows start 2 bytes off the 32-byte boundary: copied
    assert ld == 0 and t.is_contiguous()
    t, ld = _token_rows(torch.zeros(3, 5, 64, dtype=torch.bfloat16), 16)
    assert ld == 0                                            # packed already
    one = both[:, :1]                                         # one token per sequence: rows one batch stride apart
    t, ld = _token_rows(one[..., :64], 16)
    assert t.data_ptr() == one.data_ptr() and ld == 5 * 208
    wide = torch.zeros(3, 5, 198, dtype=torch.float32)[..., :64]          # 792-byte rows: not a multiple of 64 bytes
    t, ld = _token_rows(wide, 16)
    assert ld == 0 and t.is_contiguous()
    t, ld = _token_rows(both.transpose(0, 1)[..., :64], 16)   # token rows not one after the other: copied
    assert ld == 0 and t.is_contiguous()
