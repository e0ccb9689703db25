"""Lane-level model of the LDS-resident forward (csrc/msda_fwd_mma.hip) on the CPU.

The kernel's correctness rests on index arithmetic that no compiler checks: the channel permutation of
the LDS image, the addresses its lanes hand to the transposing LDS read, the packing of the weight
operand, and where the product's rows land among the 64 lanes.  This model executes exactly those
formulas (transcribed from the kernel, names kept) on top of the three hardware behaviours they assume --
pinned on the GPU by tools/ubench/mfma16_probe.hip:

  * ds_read_b64_tr_b16: in every 16-lane group, output lane j element e = halfword j % 4 of the 8 bytes
    whose address lane 4e + j // 4 supplied;
  * v_mfma_f32_16x16x32: A lane l = row l % 16, k = 8 (l // 16) + i; B lane l = column l % 16, same k;
    D lane l = column l % 16, rows 4 (l // 16) + i;
  * DPP row_ror:8 = lane ^ 8;

and compares a wave's accumulators with a direct bilinear sum.  It needs no GPU and no library.
"""
import numpy as np
import pytest

K_CHUNK = 16


def geom(D):
    RB = D * 2
    LPI = RB // 16
    return dict(RB=RB, LPI=LPI, QPW=64 // LPI, NG=D // 16, RP=RB + 32, QSTRIDE=(2 * K_CHUNK + 1) * 16)


def img_pos(D, lig, i):
    if D == 128:
        return i * 16 + lig
    return i * 16 + lig if i < 4 else (i - 4) * 16 + lig + 8


def line_pitch(D, W):
    raw = W * geom(D)["RP"]
    return raw + ((64 - raw % 256) & 255)


def tr_read(lds, addr):
    """ds_read_b64_tr_b16 of one wave: addr[64] byte addresses -> out[64][4] halfwords."""
    out = np.zeros((64, 4), dtype=np.float64)
    for l in range(64):
        grp, j = l // 16, l % 16
        for e in range(4):
            src = grp * 16 + 4 * e + j // 4
            a = addr[src] + 2 * (j % 4)
            assert addr[src] % 8 == 0
            out[l, e] = lds[a // 2]
    return out


def mfma(A, B):
    """A[64][8], B[64][8] per-lane operands -> D[64][4] per-lane results."""
    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for l in range(64):
        for i in range(8):
            Am[l % 16, 8 * (l // 16) + i] = A[l, i]
            Bm[8 * (l // 16) + i, l % 16] = B[l, i]
    Dm = Am @ Bm
    out = np.zeros((64, 4))
    for l in range(64):
        for i in range(4):
            out[l, i] = Dm[4 * (l // 16) + i, l % 16]
    return out


@pytest.mark.parametrize("D", [128, 64])
def test_wave_model_matches_bilinear(D):
    g = geom(D)
    LPI, QPW, NG, RP = g["LPI"], g["QPW"], g["NG"], g["RP"]
    rng = np.random.default_rng(D)
    # two resident levels (non-square), all 16 samples of the chunk in LDS levels -> two batches of 8
    levels = [(5, 7), (3, 4)]
    P = 8
    # image: halfword array, "values" are small integers so every product is exact
    base, lps, imgs = [], [], []
    off = RP                                         # the zero row first
    for (H, W) in levels:
        lp = line_pitch(D, W)
        assert lp % 256 == 64 and lp % 8 == 0
        base.append(off); lps.append(lp)
        off += H * lp
    lds = np.zeros(off // 2 + 16)
    vals = [rng.integers(-8, 9, size=(H, W, D)).astype(np.float64) for (H, W) in levels]
    for li, (H, W) in enumerate(levels):
        for y in range(H):
            for x in range(W):
                for lig in range(LPI):
                    for i in range(8):
                        lds[(base[li] + y * lps[li] + x * RP) // 2 + img_pos(D, lig, i)] = vals[li][y, x, 8 * lig + i]
    # samples: QPW queries x 16 samples; weights split in hi / lo parts held as two numbers
    n_l = K_CHUNK
    rec_off = np.zeros((QPW, K_CHUNK, 4), dtype=np.int64)           # record [query][index] -> 4 image offsets
    rec_hi = np.zeros((QPW, K_CHUNK, 4)); rec_lo = np.zeros((QPW, K_CHUNK, 4))
    want = np.zeros((QPW, D))
    for q in range(QPW):
        for kk in range(K_CHUNK):
            li = kk // P
            H, W = levels[li]
            y0 = int(rng.integers(-1, H)); x0 = int(rng.integers(-1, W))
            w = rng.integers(1, 5, size=4) / 4.0
            ridx = K_CHUNK - 1 - kk                                  # every sample is an LDS sample: rank = kk
            for c in range(4):
                yy, xx = y0 + (c >> 1), x0 + (c & 1)
                ok = 0 <= yy < H and 0 <= xx < W
                if ok:
                    rec_off[q, ridx, c] = base[li] + yy * lps[li] + xx * RP
                    hi = np.floor(w[c] * 2) / 2; lo = w[c] - hi
                    rec_hi[q, ridx, c] = hi; rec_lo[q, ridx, c] = lo
                    want[q] += w[c] * vals[li][yy, xx]
    # ---- the wave
    acc = np.zeros((64, 8))
    for b8 in range((n_l + 7) // 8):
        A = np.zeros((64, 8))
        for lane in range(64):
            am, akb = lane & 15, lane >> 4
            if D == 128:
                a_q, a_part, on = am >> 2, am & 3, (am & 3) < 2
            else:
                a_q, a_part, on = 2 * (am >> 2) + ((am >> 1) & 1), am & 1, True
            r0 = 8 * b8 + 2 * akb
            src = rec_lo if (a_part & 1) else rec_hi
            if on and r0 < n_l:
                A[lane, 0:4] = src[a_q, K_CHUNK - 1 - r0]
            if on and r0 + 1 < n_l:
                A[lane, 4:8] = src[a_q, K_CHUNK - 2 - r0]
        for j in range(QPW):
            ad = np.zeros((2, 64), dtype=np.int64)
            for lane in range(64):
                bG, be, bc = lane >> 4, (lane >> 2) & 3, lane & 3
                for t in range(2):
                    r = 8 * b8 + 2 * bG + t
                    o = rec_off[j, K_CHUNK - 1 - r, be] if r < n_l else 0
                    ad[t, lane] = o + 8 * bc
            for gg in range(NG):
                B = np.zeros((64, 8))
                for t in range(2):
                    B[:, 4 * t:4 * t + 4] = tr_read(lds, ad[t] + 32 * gg)
                T = mfma(A, B)
                for lane in range(64):
                    qi = lane // LPI
                    if D == 128:
                        if qi == j:
                            acc[lane, gg] += T[lane, 0] + T[lane, 1]
                    else:
                        v = lambda ln: (T[ln, 2] + T[ln, 3]) if (j & 1) else (T[ln, 0] + T[ln, 1])
                        if qi == j:
                            if j & 1:
                                acc[lane, gg] += v(lane ^ 8); acc[lane, gg + 4] += v(lane)
                            else:
                                acc[lane, gg] += v(lane); acc[lane, gg + 4] += v(lane ^ 8)
    got = np.zeros((QPW, D))
    for lane in range(64):
        qi, lig = lane // LPI, lane % LPI
        got[qi, 8 * lig:8 * lig + 8] = acc[lane]
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)


@pytest.mark.parametrize("D", [128, 64])
def test_footprint_corners_never_share_a_bank_slot(D):
    """The four rows a sample's footprint hands to one transposing read sit in four different 32-byte bank slots."""
    g = geom(D)
    for W in range(1, 70):
        lp = line_pitch(D, W)
        for x in range(W - 1):
            offs = [x * g["RP"], (x + 1) * g["RP"], lp + x * g["RP"], lp + (x + 1) * g["RP"]]
            slots = {(o % 256) // 32 for o in offs}
            assert len(slots) == 4, (W, x, offs)


def test_taps_wave_model_matches_dots():
    """csrc/msda_taps_mma.hip: the gathered value rows of 4 samples x 4 corners of ONE query as the A operand
    (a lane reads 16 bytes of its row out of the natural-order image), the wave's four grad_out rows as the B
    operand (column n = query n mod 4), K chained over D / 32 steps.  Lane (column j, row quad s) must end up with
    the four corner dots of sample s of query j."""
    D, QPW, NKS = 128, 4, 4
    g = geom(D)
    RP = g["RP"]
    rng = np.random.default_rng(7)
    H, W = 5, 6
    lp = line_pitch(D, W)
    base = RP
    img = np.zeros((base + H * lp) // 2 + 64)                       # halfword array, natural channel order
    vals = rng.integers(-4, 5, size=(H, W, D)).astype(np.float64)
    for y in range(H):
        for x in range(W):
            img[(base + y * lp + x * RP) // 2:(base + y * lp + x * RP) // 2 + D] = vals[y, x]
    gout = rng.integers(-3, 4, size=(QPW, D)).astype(np.float64)   # gsh: [query][channel]
    n_l = 7                                                         # a ragged tile: samples 4..6 in the second tile
    rec_off = np.zeros((QPW, K_CHUNK, 4), dtype=np.int64)
    want = np.zeros((QPW, n_l, 4))
    for q in range(QPW):
        for r in range(n_l):
            y0 = int(rng.integers(-1, H)); x0 = int(rng.integers(-1, W))
            for c in range(4):
                yy, xx = y0 + (c >> 1), x0 + (c & 1)
                if 0 <= yy < H and 0 <= xx < W:
                    rec_off[q, K_CHUNK - 1 - r, c] = base + yy * lp + xx * RP
                    want[q, r, c] = vals[yy, xx] @ gout[q]
    got = np.full((QPW, n_l, 4), np.nan)
    for t4 in range((n_l + 3) // 4):
        for j in range(QPW):
            acc = np.zeros((64, 4))
            for ks in range(NKS):
                A = np.zeros((64, 8)); B = np.zeros((64, 8))
                for lane in range(64):
                    am, akb = lane & 15, lane >> 4
                    r = 4 * t4 + (am >> 2)
                    off = rec_off[j, K_CHUNK - 1 - r, am & 3] if r < n_l else 0
                    a0 = (off + 16 * akb + 64 * ks) // 2
                    A[lane] = img[a0:a0 + 8]
                    bn, bkb = lane & 15, lane >> 4
                    b0 = (4 * ks + bkb) * 8
                    B[lane] = gout[bn & (QPW - 1), b0:b0 + 8]
                acc += mfma(A, B)
            for lane in range(64):
                rs = 4 * t4 + (lane >> 4)
                if (lane & 15) == j and rs < n_l:
                    got[j, rs] = acc[lane]
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
