"""Lane-level model of the wave-per-query forward (csrc/msda_fwd_wq.hip) on the CPU.

What the kernel rests on, none of it checked by a compiler: that a sample's four pixel rows AS A WAVE-WIDE LOAD DELIVERS
THEM (lane l = 16 bytes = channels 8 (l % 16) .. + 7 of corner l // 16) are the B operand of v_mfma_f32_16x16x32; that the
A operand built by four ANDs of a weight word with lane-constant masks is the diagonal A[m][8j + i] = (i == m % 8) ?
part_{m / 8}(w_j) : 0; that the accumulator's rows are (part, channel-in-lane); that two v_permlane32_swap + two adds
leave lane l with channels 8n + 4g + {0, 1} (l < 32) / {2, 3} (l >= 32); and the record layout ([corner][sample] offsets,
[part x corner][sample] weight words, batches at a 208-byte stride).  This model executes exactly those formulas
(transcribed from the kernel, names kept) on top of the hardware behaviours they assume -- the operand layouts are pinned on
the GPU by tools/ubench/mfma16_probe.hip:

  * v_mfma_f32_16x16x32: A lane l = row l % 16, k = 8 (l // 16) + i; B lane l = column l % 16, same k; D lane l = column
    l % 16, rows 4 (l // 16) + i;
  * v_permlane32_swap(x, y): x' = [x.lanes 0-31, y.lanes 0-31], y' = [x.lanes 32-63, y.lanes 32-63];

and compares a wave's output row with a direct weighted sum of rows.  It needs no GPU and no library."""
import numpy as np
import pytest

K_CHUNK, BATCH, BATCH_W, QSTRIDE = 16, 208, 64, 4 * 208


def bf16_trunc(x):
    """Leading 16 bits of an fp32 (what FwdMma<bf16>::split_dup keeps as the hi part)."""
    return (np.float32(x).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def bf16_rne(x):
    u = np.float32(x).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return np.uint32(u).view(np.float32)


def mfma(A, B, C):
    """A[64][8], B[64][8] per-lane operands (as real numbers), C[64][4] -> D[64][4]."""
    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for l in range(64):
        for i in range(8):
            Am[l % 16, 8 * (l // 16) + i] = A[l, i]
            Bm[8 * (l // 16) + i, l % 16] = B[l, i]
    Dm = Am @ Bm
    D = C.copy()
    for l in range(64):
        for i in range(4):
            D[l, i] += Dm[4 * (l // 16) + i, l % 16]
    return D


def permlane32_swap(x, y):
    return np.concatenate([x[:32], y[:32]]), np.concatenate([x[32:], y[32:]])


def stage(records, sq, ridx, off, w):
    """The staging lane of sample `ridx` of query `sq`: offsets and duplicated weight parts into the batch-major records
    (one dword per store, as the kernel's dst[4 * c], dst[BW / 4 + 4 * c], dst[BW / 4 + 16 + 4 * c])."""
    base = sq * QSTRIDE + (ridx >> 2) * BATCH + (ridx & 3) * 4
    for c in range(4):
        hi = bf16_trunc(w[c]); lo = bf16_rne(np.float32(w[c]) - hi)
        records[(base + 16 * c) // 4] = ("off", off[c])
        records[(base + BATCH_W + 16 * c) // 4] = ("w", float(hi))            # the 16-bit part in BOTH halves of the word
        records[(base + BATCH_W + 64 + 16 * c) // 4] = ("w", float(lo))


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_one_product_per_sample_sums_the_four_corners_of_all_channels(seed):
    rng = np.random.default_rng(seed)
    n_rows, D = 40, 128
    value = rng.standard_normal((n_rows, D)).astype(np.float32)
    value = bf16_rne(value).astype(np.float64)                                # 16-bit storage
    QG = 4
    n_samples = 16
    rows = rng.integers(0, n_rows, size=(QG, n_samples, 4))
    weights = rng.random((QG, n_samples, 4)).astype(np.float32) * 0.25
    records = {}
    for sq in range(QG):
        for k in range(n_samples):
            stage(records, sq, k, rows[sq, k], weights[sq, k])
    for qi in range(QG):
        acc = np.zeros((64, 4))
        for bt in range(4):
            for u in range(4):
                A = np.zeros((64, 8)); B = np.zeros((64, 8))
                for lane in range(64):
                    cj, cn = lane >> 4, lane & 15
                    slot_off = cj * 16
                    slot_w = BATCH_W + ((cn >> 3) * 4 + cj) * 16
                    ra = qi * QSTRIDE + bt * BATCH
                    kind, off = records[(ra + slot_off + 4 * u) // 4]
                    assert kind == "off"
                    kind, ww = records[(ra + slot_w + 4 * u) // 4]
                    assert kind == "w"
                    # amask[r] = (r == (cn & 7) >> 1) ? (cn & 1 ? hi half : lo half) : 0   ->  halfword cn & 7 of the fragment
                    A[lane, cn & 7] = ww
                    B[lane, :] = value[off, 8 * cn: 8 * cn + 8]              # the lane's 16 bytes of corner cj's row
                acc = mfma(A, B, acc)
        # epilogue: swap(a4[0], a4[2]), swap(a4[1], a4[3]); e = s02[0] + s02[1]; f = s13[0] + s13[1]
        s02 = permlane32_swap(acc[:, 0], acc[:, 2])
        s13 = permlane32_swap(acc[:, 1], acc[:, 3])
        e, f = s02[0] + s02[1], s13[0] + s13[1]
        out = np.zeros(D)
        for lane in range(64):
            cj, cn = lane >> 4, lane & 15
            out_dword = 4 * cn + 2 * (cj & 1) + (lane >> 5)
            out[2 * out_dword], out[2 * out_dword + 1] = e[lane], f[lane]
        # what the products should have summed: hi + lo parts of every weight times its row, channel by channel
        want = np.zeros(D)
        for k in range(n_samples):
            for c in range(4):
                hi = bf16_trunc(weights[qi, k, c]); lo = bf16_rne(np.float32(weights[qi, k, c]) - hi)
                want += (float(hi) + float(lo)) * value[rows[qi, k, c]]
        assert np.abs(out - want).max() <= 1e-9
        exact = sum(float(weights[qi, k, c]) * value[rows[qi, k, c]] for k in range(n_samples) for c in range(4))
        assert np.abs(out - exact).max() <= 2.0 ** -16 * 16 * 4               # hi + lo carries >= 16 significant bits of a weight


def test_a_non_finite_element_spreads_inside_its_lane_group_only_and_is_always_seen():
    """The zeros of A multiply the other channels of a sampled row: an Inf in channel c turns the lane group's 8 channels into
    NaN (0 x Inf) -- never another group's, and never silently: every product row of the column sees it, so the epilogue's
    finiteness test of the query's sums catches every contamination (the kernel then recomputes the query channel by
    channel)."""
    value = np.ones((4, 128)); value[2, 37] = np.inf                          # corner 2, channel 37 = lane group n = 4
    A = np.zeros((64, 8)); B = np.zeros((64, 8))
    for lane in range(64):
        cj, cn = lane >> 4, lane & 15
        A[lane, cn & 7] = 0.25
        B[lane, :] = value[cj, 8 * cn: 8 * cn + 8]
    with np.errstate(invalid="ignore"):
        acc = mfma(A, B, np.zeros((64, 4)))
    bad_cols = sorted({lane & 15 for lane in range(64) if not np.isfinite(acc[lane]).all()})
    assert bad_cols == [4]
    assert all(not np.isfinite(acc[lane, i]) for lane in range(64) if (lane & 15) == 4 for i in range(4))
