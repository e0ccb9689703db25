"""Host plan and lane-level model of the workgroup-local grad_value kernel (csrc/msda_gv_mma.hip) on the CPU.

Two things no compiler checks:
  * the HOST PLAN (mmfs::gv::make_plan, read back through the C ABI's test hook): which levels the kernel takes,
    how they are grouped, how a block's samples are dealt over virtual blocks, how the queries are cut -- its
    invariants are what the kernel's indexing relies on;
  * the kernel's INDEX ARITHMETIC: the XOR swizzle of the grad_out rows in LDS and the addresses its lanes hand to
    the transposing read, the weight tile's layout, which lane holds which pixel / channel of the product, the
    epilogue's sums over a block's virtual blocks.  The model below executes those formulas (transcribed from the
    kernel, names kept) on top of the hardware behaviours pinned on the GPU by tools/ubench/mfma16_probe.hip
    (see tests/test_fwd_mma_model.py) and compares a workgroup's output rows with a direct scatter sum.
No GPU needed: the plan is host code, the model is numpy.
"""
import ctypes
import os

import numpy as np
import pytest

from test_fwd_mma_model import mfma, tr_read

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mm-interleaved_amd", "libmmfs_msda.so")

K_WAVES, K_THREADS, K_MAX_LEVELS, K_MAX_GROUPS, K_MAX_SEGS, K_TB = 16, 1024, 16, 32, 4, 4
K_LDS, K_CTRL, K_ATILE = 160 * 1024, 6144, 2048
K_ROWS0 = K_CTRL + K_WAVES * K_ATILE
K_MAX_SAMPLES, K_MAX_QC = 2048, 256


class Level(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("level", "Hl", "Wl", "lstart", "nbx", "nby")]


class Seg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint16) for n in ("lslot", "log2s", "v0", "rb0")]


class Group(ctypes.Structure):
    _fields_ = [("seg", Seg * K_MAX_SEGS)] + [(n, ctypes.c_uint16) for n in ("nseg", "nvb", "qc", "nrb", "qparts", "wg0")] + \
               [("pbase", ctypes.c_uint32)]


class Table(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("n_levels", "n_groups", "wgs_per_slab", "ptiles_per_slab")] + \
               [("skip", ctypes.c_uint64 * 2), ("lv", Level * K_MAX_LEVELS), ("g", Group * K_MAX_GROUPS)]


def load_lib():
    if not os.path.exists(LIB):
        pytest.skip("libmmfs_msda.so not built")
    lib = ctypes.CDLL(LIB)
    lib.mmfs_msda_debug_value_plan.restype = ctypes.c_int64
    lib.mmfs_msda_debug_value_plan.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int64] * 7 + \
        [ctypes.c_uint, ctypes.c_void_p, ctypes.c_int64]
    return lib


def plan(shapes, B, H, D, Nq, P, flags=512, dtype=2):          # 512: MMFS_BWD_VALUE_LDS_BLOCKS (the kernel is opt-in)
    lib = load_lib()
    hs = np.ascontiguousarray(np.array(shapes, dtype=np.int64).reshape(-1, 2))
    px = hs[:, 0] * hs[:, 1]
    hst = np.ascontiguousarray(np.cumsum(px) - px)
    t = Table()
    assert lib.mmfs_msda_debug_value_plan(dtype, hs.ctypes.data, hst.ctypes.data, B, int(px.sum()), H, D, len(hs), Nq, P,
                                          flags, ctypes.addressof(t), ctypes.sizeof(t)) == ctypes.sizeof(t)
    return t, hs, hst


def rows_alloc(qc, lpr):
    return ((qc * lpr + K_THREADS - 1) // K_THREADS * K_THREADS) // lpr


def lds_bytes(qc, ns, rb):
    return K_ROWS0 + (rows_alloc(qc, rb // 16) + 1) * rb + ((ns * 8 + 15) & ~15) + 4 * ns * 4


GEOMETRIES = {
    # name: (shapes, B, H, D, Nq, P)
    "north_star": ([(64, 64), (32, 32), (16, 16), (8, 8)], 8, 8, 128, 4096, 4),
    "sd_real": ([(64, 64), (32, 32), (16, 16), (8, 8)], 8, 16, 64, 4096, 8),
    "llm_n4": ([(32, 32), (16, 16), (8, 8)] * 4, 4, 16, 64, 2048, 8),
    "ref_speed": ([(16, 16), (8, 8)], 32, 8, 128, 128, 64),
    "ragged": ([(13, 9), (7, 21), (3, 3), (1, 5), (40, 40)], 2, 4, 64, 700, 4),
    "one_pixel_levels": ([(1, 1), (2, 2), (1, 1)], 1, 2, 128, 300, 2),
}


@pytest.mark.parametrize("name", sorted(GEOMETRIES))
def test_plan_invariants(name):
    shapes, B, H, D, Nq, P = GEOMETRIES[name]
    t, hs, hst = plan(shapes, B, H, D, Nq, P)
    VB = 32 if D >= 128 else 64
    RB = 2 * D
    if name == "ref_speed":
        assert t.n_groups == 0                         # 128 queries: below the kernel's threshold
        return
    assert t.n_groups >= 1 and t.n_levels >= 1
    served = set()
    slots = set()
    wg = 0
    ptiles = 0
    for gi in range(t.n_groups):
        g = t.g[gi]
        assert 1 <= g.nseg <= K_MAX_SEGS and 1 <= g.nvb <= VB and g.wg0 == wg and g.qparts >= 1
        wg += g.qparts
        assert g.qc % 16 == 0 and 16 <= g.qc <= K_MAX_QC
        ns = g.qc * g.nseg * P
        assert ns <= K_MAX_SAMPLES and lds_bytes(g.qc, ns, RB) <= K_LDS
        assert g.qparts <= (Nq + g.qc - 1) // g.qc        # every query range holds at least one chunk
        v0 = rb0 = 0
        last_l2 = 99
        for si in range(g.nseg):
            sg = g.seg[si]
            lv = t.lv[sg.lslot]
            assert sg.lslot not in slots
            slots.add(sg.lslot)
            assert (lv.Hl, lv.Wl) == tuple(hs[lv.level]) and lv.lstart == hst[lv.level]
            assert lv.nbx == (lv.Wl + 3) // 4 and lv.nby == (lv.Hl + 3) // 4
            assert lv.level not in served
            served.add(lv.level)
            assert sg.v0 == v0 and sg.rb0 == rb0
            assert sg.log2s <= 4 and sg.log2s <= last_l2      # descending splits: a block's virtual blocks share a slot
            assert v0 % (1 << sg.log2s) == 0
            last_l2 = sg.log2s
            v0 += (lv.nbx * lv.nby) << sg.log2s
            rb0 += lv.nbx * lv.nby
        assert v0 == g.nvb and rb0 == g.nrb
        if g.qparts > 1:
            assert g.pbase == ptiles
            ptiles += g.qparts * g.nrb
    assert wg == t.wgs_per_slab and ptiles == t.ptiles_per_slab
    mask = sum(1 << l for l in served)
    assert (t.skip[0] | (t.skip[1] << 64)) == mask
    # what the kernel is FOR: the north star's two small levels, all of the LLM path's, three of the UNet's four
    want = {"north_star": {2, 3}, "sd_real": {1, 2, 3}, "llm_n4": set(range(12))}.get(name)
    if want is not None:
        assert served == want


def test_plan_is_off_unless_asked_for():
    shapes, B, H, D, Nq, P = GEOMETRIES["north_star"]
    for flags in (0, 256, 256 | 512):                          # default; MMFS_BWD_VALUE_SORTED_ONLY; ... wins over LDS_BLOCKS
        t, _, _ = plan(shapes, B, H, D, Nq, P, flags=flags)
        assert t.n_groups == 0 and t.skip[0] == 0


# ---------------------------------------------------------------------------------------------- the kernel's model
def swz(D, r):
    return (r & 7) if D >= 128 else ((r >> 1) & 3)


def perm(p):
    return (p & 0x13) | ((p & 4) << 1) | ((p & 8) >> 1)


def tr_read_conflicts(D, addr):
    """Worst number of lanes of one 32-lane phase of a transposing read that hit the same 32-byte bank slot with
    DIFFERENT rows (the model of the hardware behind the swizzle: 64 banks x 4 bytes = eight 32-byte slots)."""
    worst = 1
    for ph in range(2):
        slots = {}
        for lane in range(32 * ph, 32 * ph + 32):
            a = int(addr[lane])
            slots.setdefault((a % 256) // 32, set()).add(a // 32)
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


stats = {"conflict_free_reads": 0}


def run_workgroup(t, gi, part, D, Nq, P, samples, grad_out, out, partial_out):
    """One workgroup of msda_gv_mma: group gi, query range `part` of one slab.
    samples[level][q][p] = (y0, x0, fy, fx, a, live); grad_out[q][D].
    Adds what the workgroup would store into out[level][y][x][D] (qparts == 1) or partial_out[part][rb][px][D]."""
    g = t.g[gi]
    RB, LPR, NT = 2 * D, 2 * D // 16, D // 16
    SLOTS = 2 if D >= 128 else 4
    QC, nvb, nseg, nrb = g.qc, g.nvb, g.nseg, g.nrb
    NLP = nseg * P
    NS = QC * NLP
    nchunks = (Nq + QC - 1) // QC
    c0, c1 = part * nchunks // g.qparts, (part + 1) * nchunks // g.qparts
    seg = []
    for si in range(nseg):
        sg = g.seg[si]
        lv = t.lv[sg.lslot]
        seg.append(dict(H=lv.Hl, W=lv.Wl, nbx=lv.nbx, v0=sg.v0, l2=sg.log2s, level=lv.level, rb0=sg.rb0))
    vbd = []
    for v in range(nvb):
        s = 0
        while s + 1 < nseg and seg[s + 1]["v0"] <= v:
            s += 1
        rel = v - seg[s]["v0"]
        rb = rel >> seg[s]["l2"]
        by = rb // seg[s]["nbx"]
        vbd.append((s, by, rb - by * seg[s]["nbx"], rel & ((1 << seg[s]["l2"]) - 1)))
    QZ = rows_alloc(QC, LPR)
    acc = np.zeros((K_WAVES, SLOTS, NT, 64, 4))
    for c in range(c0, c1):
        q0 = c * QC
        # ---- rows: halfword image of the LDS row area (+ the row of zeros)
        rows = np.zeros(((QZ + 1) * RB) // 2)
        for i in range(QC * LPR):
            r, cpos = i // LPR, i % LPR
            if q0 + r < Nq:
                src = (cpos ^ (2 * swz(D, r))) * 8                      # halfwords
                rows[i * 8:i * 8 + 8] = grad_out[q0 + r, src:src + 8]
        # ---- bin
        cnt = [0] * nvb
        ccnt = np.zeros((nvb, 8), dtype=np.int64)
        keys = []
        samp = {}
        for sidx in range(NS):
            ql = sidx // NLP
            rem = sidx - ql * NLP
            ls = rem // P
            p = rem - ls * P
            if q0 + ql >= Nq:
                continue
            st = seg[ls]
            y0, x0, fy, fx, a, live = samples[st["level"]][q0 + ql][p]
            samp[sidx] = (y0, x0, fy, fx, a)
            if not live:
                continue
            ya, yb = max(y0, 0) >> 2, min(y0 + 1, st["H"] - 1) >> 2
            xa, xb = max(x0, 0) >> 2, min(x0 + 1, st["W"] - 1) >> 2
            v00 = st["v0"] + (sidx & ((1 << st["l2"]) - 1))
            for j in range(4):
                by, bx = (yb if j >> 1 else ya), (xb if j & 1 else xa)
                if ((j >> 1) and yb == ya) or ((j & 1) and xb == xa):
                    continue
                v = v00 + ((by * st["nbx"] + bx) << st["l2"])
                assert 0 <= v < nvb
                cl = ql & 7
                keys.append((v, cl, int(ccnt[v, cl]), sidx, ql))
                ccnt[v, cl] += 1
                cnt[v] += 1
        lbase = np.concatenate([[0], np.cumsum(cnt)])[:-1]
        recs = np.full(max(1, sum(cnt)), -1, dtype=np.int64)
        for v, cl, r, sidx, ql in keys:
            pos = lbase[v] + sum(min(int(ccnt[v, c2]), r + (1 if c2 < cl else 0)) for c2 in range(8))
            assert recs[pos] == -1 and pos < lbase[v] + cnt[v]
            recs[pos] = sidx | (ql << 16)
        assert (recs[:sum(cnt)] >= 0).all()
        # ---- products
        for wave in range(K_WAVES):
            for SL in range(SLOTS):
                v = wave + K_WAVES * SL
                if v >= nvb or cnt[v] == 0:
                    continue
                n = cnt[v]
                lst = recs[lbase[v]:lbase[v] + n]
                s, by, bx, _ = vbd[v]
                by4, bx4, Hl, Wl = 4 * by, 4 * bx, seg[s]["H"], seg[s]["W"]
                for p0 in range(0, n, 32):
                    cs = min(32, n - p0)
                    atile = np.zeros(K_ATILE // 2)                       # halfwords; the weights' hi / lo parts as numbers
                    for tt in range(2):
                        for lane in range(64):
                            wr, wcy, wcx = lane >> 2, (lane >> 1) & 1, lane & 1
                            pl = 16 * tt + wr
                            k = perm(pl)
                            if pl >= cs:
                                continue
                            rec = int(lst[p0 + pl])
                            y0, x0, fy, fx, a = samp[rec & 0xffff]
                            yy, xx = y0 + wcy, x0 + wcx
                            py, px = yy - by4, xx - bx4
                            if 0 <= py < 4 and 0 <= px < 4 and yy < Hl and xx < Wl:
                                wgt = (fy if wcy else 1 - fy) * (fx if wcx else 1 - fx) * a
                                hi = np.floor(wgt * 8) / 8
                                lo = wgt - hi
                                m = py * 4 + px
                                o = m * 64 + ((((k >> 3) ^ (m >> 2)) & 3) << 4) + ((k & 7) << 1)
                                assert atile[o // 2] == 0 and o < 1024
                                atile[o // 2] = hi
                                atile[(1024 + o) // 2] = lo
                    Ah = np.zeros((64, 8)); Al = np.zeros((64, 8))
                    ad = np.zeros((2, 64), dtype=np.int64)
                    xs = np.zeros((2, 64), dtype=np.int64)
                    for lane in range(64):
                        am, akc = lane & 15, lane >> 4
                        a_off = am * 64 + (((akc ^ (am >> 2)) & 3) << 4)
                        Ah[lane] = atile[a_off // 2:a_off // 2 + 8]
                        Al[lane] = atile[(1024 + a_off) // 2:(1024 + a_off) // 2 + 8]
                        bG, be, bc = lane >> 4, (lane >> 2) & 3, lane & 3
                        for tt in range(2):
                            pl = perm(8 * bG + 4 * tt + be)
                            q = int(lst[p0 + pl]) >> 16 if pl < cs else QZ
                            ad[tt, lane] = q * RB + 8 * bc
                            xs[tt, lane] = swz(D, q) << 5
                    # a step whose 32 places hold four full rounds of the eight classes reads without a bank conflict
                    full = cs == 32 and all(sorted((int(lst[p0 + 8 * r8 + i]) >> 16) & 7 for i in range(8)) == list(range(8)) for r8 in range(4))
                    for nt in range(NT):
                        Bv = np.zeros((64, 8))
                        for tt in range(2):
                            a = ad[tt] + ((nt << 5) ^ xs[tt])
                            if full:
                                assert tr_read_conflicts(D, a) == 1
                                stats["conflict_free_reads"] += 1
                            Bv[:, 4 * tt:4 * tt + 4] = tr_read(rows, a)
                        acc[wave, SL, nt] += mfma(Ah, Bv) + mfma(Al, Bv)
    # ---- epilogue
    for sl in range(SLOTS):
        tl = np.zeros((K_WAVES, 16, D))
        for wave in range(K_WAVES):
            if wave + K_WAVES * sl < nvb:
                for nt in range(NT):
                    for lane in range(64):
                        for i in range(4):
                            tl[wave, 4 * (lane >> 4) + i, 16 * nt + (lane & 15)] = acc[wave, sl, nt, lane, i]
        for e in range(K_WAVES * 16 * LPR):
            w, px, c8 = e // (16 * LPR), (e // LPR) & 15, e % LPR
            vv = K_WAVES * sl + w
            if vv >= nvb or vbd[vv][3] != 0:
                continue
            s = vbd[vv][0]
            ns = 1 << seg[s]["l2"]
            sm = np.zeros(8)
            for j in range(ns):
                sm += tl[w + j, px, c8 * 8:c8 * 8 + 8]
            rbl = (vv - seg[s]["v0"]) >> seg[s]["l2"]
            if g.qparts == 1:
                by = rbl // seg[s]["nbx"]
                bx = rbl - by * seg[s]["nbx"]
                y, x = 4 * by + (px >> 2), 4 * bx + (px & 3)
                if y < seg[s]["H"] and x < seg[s]["W"]:
                    out[seg[s]["level"]][y, x, c8 * 8:c8 * 8 + 8] += sm
            else:
                partial_out[part, seg[s]["rb0"] + rbl, px, c8 * 8:c8 * 8 + 8] += sm
    return seg


MODEL_CASES = [
    # (shapes, H, D, Nq, P): small enough for python loops, every mechanism on
    ([(9, 6), (5, 5), (2, 3)], 1, 128, 300, 2),         # D = 128; blocks at the map's edge; one group, several ranges
    ([(6, 11), (4, 4)], 1, 64, 272, 4),                 # D = 64: four slots per wave
]


@pytest.mark.parametrize("target", [3, 1])
@pytest.mark.parametrize("case", range(len(MODEL_CASES)))
def test_workgroup_model_matches_scatter(case, target, monkeypatch):
    shapes, H, D, Nq, P = MODEL_CASES[case]
    # (read per call) 3: one level per group, several query ranges; 1: several levels per group, one range
    monkeypatch.setenv("MMFS_GV_TARGET_WGS", str(target))
    t, hs, hst = plan(shapes, 1, H, D, Nq, P)
    assert t.n_groups >= 1
    if target == 1:
        assert any(t.g[gi].nseg > 1 for gi in range(t.n_groups))
    elif case == 1:
        assert any(t.g[gi].qparts > 1 for gi in range(t.n_groups))
    rng = np.random.default_rng(7 + case)
    L = len(shapes)
    samples = []
    for (Hl, Wl) in shapes:
        lv = []
        for q in range(Nq):
            row = []
            for p in range(P):
                y0, x0 = int(rng.integers(-1, Hl)), int(rng.integers(-1, Wl))
                fy, fx = rng.integers(0, 4) / 4.0, rng.integers(0, 4) / 4.0
                a = float(rng.integers(0, 3)) / 2.0
                live = a != 0.0 and rng.random() > 0.1          # (a dead sample: outside the map)
                row.append((y0, x0, fy, fx, a, live))
            lv.append(row)
        samples.append(lv)
    grad_out = rng.integers(-4, 5, size=(Nq, D)).astype(np.float64)
    want = [np.zeros((Hl, Wl, D)) for (Hl, Wl) in shapes]
    for l, (Hl, Wl) in enumerate(shapes):
        for q in range(Nq):
            for p in range(P):
                y0, x0, fy, fx, a, live = samples[l][q][p]
                if not live:
                    continue
                for cy in range(2):
                    for cx in range(2):
                        yy, xx = y0 + cy, x0 + cx
                        if 0 <= yy < Hl and 0 <= xx < Wl:
                            want[l][yy, xx] += (fy if cy else 1 - fy) * (fx if cx else 1 - fx) * a * grad_out[q]
    out = [np.zeros((Hl, Wl, D)) for (Hl, Wl) in shapes]
    served = set()
    for gi in range(t.n_groups):
        g = t.g[gi]
        partial = np.zeros((g.qparts, g.nrb, 16, D))
        seg = None
        for part in range(g.qparts):
            seg = run_workgroup(t, gi, part, D, Nq, P, samples, grad_out, out, partial)
        if g.qparts > 1:                                   # the range that arrives last adds the partial tiles up
            for rbg in range(g.nrb):
                s = 0
                while s + 1 < g.nseg and seg[s + 1]["rb0"] <= rbg:
                    s += 1
                rbl = rbg - seg[s]["rb0"]
                by = rbl // seg[s]["nbx"]
                bx = rbl - by * seg[s]["nbx"]
                for px in range(16):
                    y, x = 4 * by + (px >> 2), 4 * bx + (px & 3)
                    if y < seg[s]["H"] and x < seg[s]["W"]:
                        out[seg[s]["level"]][y, x] += partial[:, rbg, px].sum(0)
        for si in range(g.nseg):
            served.add(t.lv[g.seg[si].lslot].level)
    assert served and stats["conflict_free_reads"] > 0
    for l in served:
        np.testing.assert_allclose(out[l], want[l], rtol=0, atol=1e-9)
