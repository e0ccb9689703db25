"""CPU model of two index rules csrc/msda_bwd_taps_sorted.hip relies on (no GPU):

* the chunk swizzle of its grad_out row slots -- row r stores logical 16-byte chunk c at position c ^ swz(r) -- makes the
  B operand's ``ds_read_b128`` (lane (n, g) reads row n, chunk 4 t + g of product t) free of bank conflicts for every head
  width: brute force over the instruction's four 16-lane groups (MI355X_MICROARCH.md, LDS);
* the mapping of the block's 5x5 pixels onto the two accumulator tiles puts, for every sample position (iy, ix) of a
  block, each of its four corners where the selection of the kernel looks for it.
"""
import itertools

import pytest

# lane groups of ds_read_b128: one LDS cycle each when conflict-free
B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
               [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
               [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
               [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63]]


def swz(D, r):                 # TsGeom<D>::swz
    return r if D == 128 else (r >> 1) & 7 if D == 64 else ((r >> 3) & 1) * 3


@pytest.mark.parametrize("D", [32, 64, 128])
def test_b_operand_reads_are_conflict_free(D):
    RB, LPR, KT = D * 2, D * 2 // 16, D // 32
    for t in range(KT):
        for grp in B128_GROUPS:
            slots = {}
            for lane in grp:
                n, g = lane % 16, lane // 16
                pc = (4 * t + g) ^ swz(D, n)
                assert 0 <= pc < LPR                                # the swizzle stays inside the row
                addr = n * RB + pc * 16
                slots.setdefault((addr // 16) % 16, set()).add(addr)   # 16-byte bank slots of the 256-byte bank row
            assert max(len(v) for v in slots.values()) == 1, (D, t, grp)
    # the DMA writes lane-linearly: lane i of instruction u serves row u * RPI + i // LPR, position i % LPR -- every
    # (row, position) of a slot exactly once
    RPI = 64 // LPR
    seen = {(u * RPI + i // LPR, i % LPR) for u in range(16 // RPI) for i in range(64)}
    assert len(seen) == 16 * LPR


def pixel_of(group, m):
    """A operand row m of accumulator group 0 / 1 -> pixel (py, px) of the 5x5, or None (csrc/msda_bwd_taps_sorted.hip)."""
    jj, s = m >> 2, m & 3
    if group == 0:
        return (jj, s)
    if s == 0:
        return (jj, 4)
    if s == 1:
        return (4, jj)
    if s == 2 and jj == 0:
        return (4, 4)
    return None


def test_every_pixel_of_the_5x5_has_one_place_and_the_selection_finds_the_corners():
    places = {}
    for group, m in itertools.product((0, 1), range(16)):
        p = pixel_of(group, m)
        if p is not None:
            assert p not in places
            places[p] = (group, m)
    assert sorted(places) == [(y, x) for y in range(5) for x in range(5)]
    # accumulator layout of v_mfma_f32_16x16x32: lane (n, j) holds rows 4 j + i (i = 0..3) of column n.  The kernel's lane
    # row j reads: d0[i] = pixel (j, i); d1[0] = (j, 4); d1[1] = (4, j); d1[2] = (4, 4) in lane row 0.
    for j in range(4):
        for i in range(4):
            assert pixel_of(0, 4 * j + i) == (j, i)
        assert pixel_of(1, 4 * j) == (j, 4) and pixel_of(1, 4 * j + 1) == (4, j)
    assert pixel_of(1, 2) == (4, 4)
    # selection: for a sample with top-left (iy, ix) every corner (iy + cy, ix + cx) inside the 5x5 is contributed by
    # exactly one (lane row, register) under the kernel's conditions
    for iy, ix in itertools.product(range(5), range(5)):
        for cy, cx in itertools.product((0, 1), (0, 1)):
            y, x = iy + cy, ix + cx
            if y > 4 or x > 4:
                continue                                   # outside the 5x5: outside the map (last block row / column)
            hits = 0
            for j in range(4):
                top, bot = iy == j, iy + 1 == j
                if (top and cy == 0) or (bot and cy == 1):          # pixel row j of the lane row: columns ix, ix + 1
                    hits += 1 if x <= 4 and y == j else 0
                if (iy == 4 and cy == 0) or (iy == 3 and cy == 1):  # the fifth row: pixel (4, j), and (4, 4) in lane row 0
                    hits += 1 if (y == 4 and x == j) else 0
                    hits += 1 if (y == 4 and x == 4 and j == 0) else 0
            assert hits == 1, (iy, ix, cy, cx)
