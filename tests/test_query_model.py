"""Lane-level model of the LDS tile of csrc/mmfs_query.hip (the image decoder block's layout kernels): a lane that
moves 8 pixels of one channel touches 8 rows of one tile column; the lanes of a wave differ in the pixel octet and in
the channel.  With vector v of row p stored at slot v ^ ((p >> 3) & 7) and a row pitch of 32 k + 4 words, every such
wave instruction touches each of the 64 banks with ONE dword at most -- the claim of the kernel's header, checked here
against the same index arithmetic (``tile_pitch`` / ``tile_at``); the unswizzled tile is 2- to 4-way conflicted."""
import pytest

PAD = 8


def tile_pitch(C):
    return (C + 63) // 64 * 64 + PAD


def tile_at(p, c, pitch):
    return p * pitch + ((((c >> 3) ^ (p >> 3)) & 7) | ((c >> 3) & ~7)) * 8 + (c & 7)


def plain_at(p, c, pitch):
    return p * pitch + c


def conflict_degree(units, pitch, at):
    """units: list of (first pixel, channel) per lane of ONE wave instruction series (8 halfword accesses each)."""
    worst = 0
    for i in range(8):
        banks = {}
        for p0, c in units:
            dword = (at(p0, c, pitch) + i * pitch) // 2
            banks.setdefault(dword % 64, set()).add(dword)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


# query_prep: unit u -> channel u >> shift, pixel octet u & (2^shift - 1); 64 consecutive units per wave instruction
@pytest.mark.parametrize("C,shift", [(320, 2), (640, 1), (1280, 0), (64, 3), (328, 2), (2048, 0), (16, 3)])
def test_query_prep_scatter_is_conflict_free(C, shift):
    pitch = tile_pitch(C)
    assert pitch % 8 == 0 and (pitch // 2) % 32 == 4            # rows stay 16-byte aligned; 32 k + 4 words
    total = C << shift
    plain_worst = 0
    for u0 in range(0, min(total, 4096), 64):
        units = [((u & ((1 << shift) - 1)) * 8, u >> shift) for u in range(u0, min(u0 + 64, total))]
        assert conflict_degree(units, pitch, tile_at) == 1
        plain_worst = max(plain_worst, conflict_degree(units, pitch, plain_at))
    if shift >= 2:
        assert plain_worst >= 2                                  # what the swizzle is for
    # the swizzle is a permutation of a row's vectors: every (pixel, channel) has a slot of its own inside the row
    for p in (0, 9, 23):
        slots = {tile_at(p, c, pitch) for c in range(C)}
        assert len(slots) == C and min(slots) >= p * pitch and max(slots) < (p + 1) * pitch


# tokens_add: 64 pixels x 128 channels, pitch 136; unit u -> channel u >> 3, pixel octet u & 7
def test_tokens_add_column_reads_are_conflict_free():
    pitch = 128 + PAD
    for u0 in range(0, 1024, 64):
        units = [((u & 7) * 8, u >> 3) for u in range(u0, u0 + 64)]
        assert conflict_degree(units, pitch, tile_at) == 1
