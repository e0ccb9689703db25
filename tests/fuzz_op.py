#!/usr/bin/env python3
"""Randomised parity sweep of the HIP op against the CPU oracle (run on the GPU box):
    python tests/fuzz_op.py [n_cases] [seed] [big]
Random shapes (every dispatch path: vector / scalar head widths, hybrid routing on and off, both
grad_value generations, hot spots that overflow the block lists), random location distributions
(uniform, out of range, clustered on a point, NaN / Inf sprinkled in), all storage types."""
import os
import sys
import random

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd"), os.path.join(ROOT, "tests")]
import MultiScaleDeformableAttention as MSDA  # noqa: E402
from oracle import msda_oracle  # noqa: E402

TOL = {torch.float64: 1e-12, torch.float32: 1e-5, torch.float16: 1e-3, torch.bfloat16: 8e-3}


BIG = False
SORTED = False     # round 6: every case prefers the backward on the cell-sorted records where it applies (mmfs_msda_backward_sorted)
WAVES = False      # round 5: every case a 16-bit head of 128 channels through the forward's fourth kernel (csrc/msda_fwd_wq.hip)


def one_case(rng, idx):
    dtype = rng.choice([torch.float32, torch.float16, torch.bfloat16, torch.bfloat16, torch.float64])
    B = rng.randint(1, 3)
    H = rng.choice([1, 2, 3, 4, 8, 16])
    D = rng.choice([8, 16, 24, 32, 64, 64, 128, 128, 256])
    P = rng.choice([1, 2, 3, 4, 4, 8, 8, 16, 32, 64])
    if WAVES:
        dtype, D = rng.choice([torch.float16, torch.bfloat16]), 128
        P = rng.choice([1, 2, 3, 4, 4, 4, 5, 8, 16])
    L = rng.randint(1, 6)
    shapes = [(rng.randint(1, 24), rng.randint(1, 24)) for _ in range(L)]
    if rng.random() < 0.3:
        shapes[rng.randrange(L)] = (rng.choice([32, 40]), rng.choice([32, 33]))
    Nq = rng.choice([1, 7, 31, 32, 33, 64, 100, 257, 600])
    if BIG:         # enough samples for level splits, chunked hot lists and full queue lanes
        B, H, D, P = rng.randint(1, 2), rng.choice([4, 8, 16]), rng.choice([32, 64, 128]), rng.choice([4, 8])
        shapes = [(rng.randint(1, 64), rng.randint(1, 64)) for _ in range(L)]
        Nq = rng.choice([1024, 3000, 4097])
    elif rng.random() < 0.08:      # many small slices (the encoders' regime): the cell sort's 256- / 512-lane workgroups
        B, H, D, P = rng.choice([8, 16]), 16, 32, rng.choice([4, 8])
        L = rng.choice([2, 4])
        shapes = [(rng.randint(1, 16), rng.randint(1, 16)) for _ in range(L)]
        Nq = rng.choice([33, 130, 600])
        if dtype == torch.float64:
            dtype = torch.bfloat16
    elif B * Nq * H * L * P * D > 6e7:
        Nq = 33 if P < 32 else 130
    g = torch.Generator().manual_seed(idx)
    sh = torch.tensor(shapes, dtype=torch.long)
    st = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
    S = int(sh.prod(1).sum())
    dist = rng.choice(["uniform", "wide", "point", "point", "edge"])
    loc = torch.rand(B, Nq, H, L, P, 2, generator=g)
    if dist == "wide":
        loc = loc * 1.6 - 0.3
    elif dist == "point":
        loc = loc * rng.choice([0.02, 0.1, 0.3]) + rng.choice([0.0, 0.45, 0.9])
    elif dist == "edge":
        loc = torch.round(loc * 8) / 4 - 0.5      # exact pixel centres / edges, in and out of range
    if rng.random() < 0.3 and loc.numel() > 8:
        flat = loc.view(-1)
        for _ in range(4):
            flat[rng.randrange(flat.numel())] = rng.choice([float("nan"), float("inf"), -float("inf"), 1e30])
    value = torch.rand(B, S, H, D, generator=g) - 0.3
    attn = torch.rand(B, Nq, H, L, P, generator=g) + 1e-5
    if rng.random() < 0.35:         # images a token cannot see: whole levels at exactly zero weight, per query
        keep = (torch.rand(B, Nq, 1, L, 1, generator=g) < 0.5).to(attn.dtype)
        keep[:, :, :, rng.randrange(L)] = 1.0
        attn = attn * keep
    attn = attn / attn.sum((-1, -2), keepdim=True)
    grad = torch.randn(B, Nq, H * D, generator=g)
    rt = lambda t: t.to(dtype).to(torch.float64)
    value, loc, attn, grad = rt(value), rt(loc), rt(attn), rt(grad)
    hybrid = rng.random() < 0.7
    registered = rng.random() < 0.6          # else: fresh device tables, checked on the device by the backward
    algo = rng.choice(["tile", "tile", "block", "pixel"])
    if algo == "tile":                       # the default: 4x4 blocks on the matrix cores for 16-bit storage
        os.environ.pop("MMFS_VALUE_ALGO", None)
    else:
        os.environ["MMFS_VALUE_ALGO"] = algo
    MSDA._hybrid = hybrid
    MSDA._bwd_algo = "atomic" if rng.random() < 0.08 else "auto"
    # round 6: the backward on the cell-sorted records (mmfs_msda_backward_sorted) wherever it applies -- a registered table,
    # 16-bit storage, P a power of two, the sort's kept scan -- silently the other routes elsewhere
    MSDA._taps_prefer_sorted = SORTED or rng.random() < 0.3
    # round 3's routes: the grad_value plan hosted by the taps kernels or launched on its own; a small sort window (records
    # placed window by window); the scalar scan for queries of many points
    os.environ["MMFS_PREPARE_IN_TAPS"] = rng.choice(["0", "1", "1"])
    if rng.random() < 0.25:
        os.environ["MMFS_SORT_WINDOW_KB"] = "21"
    else:
        os.environ.pop("MMFS_SORT_WINDOW_KB", None)
    os.environ["MMFS_SORT_MANY_POINTS"] = rng.choice(["0", "1", "1"])
    os.environ["MMFS_SORT_SMALL"] = rng.choice(["0", "1", "1"])
    # round 4: the forward's third kernel wherever it is supported (the library's own choice takes it only where the whole
    # pyramid is resident), next to the library's choice and the two older kernels
    MSDA._fwd_algo = rng.choice(["auto", "auto", "slices", "slices", "gather", "lds"])
    if WAVES:
        MSDA._fwd_algo = "waves"
    MSDA.reload_env()                        # (the library reads its knobs once: csrc/msda_env.h)
    dev = lambda t: t.to("cuda", dtype) if t.is_floating_point() else t.to("cuda")
    desc = (f"#{idx} {str(dtype)[6:]} B{B} H{H} D{D} P{P} Nq{Nq} {shapes} {dist} hybrid={hybrid} registered={registered} "
            f"value={algo} bwd={MSDA._bwd_algo} fwd={MSDA._fwd_algo} sorted={MSDA._taps_prefer_sorted}")
    dsh, dst = dev(sh), dev(st)
    if registered:
        MSDA.register_level_tables(dsh, dst, S, sh.numpy(), st.numpy())
    try:
        out = MSDA.ms_deform_attn_forward(dev(value), dsh, dst, dev(loc), dev(attn), 1)
    except RuntimeError as e:            # a forced kernel refuses shapes outside its range: the library's choice then
        if MSDA._fwd_algo == "auto" or f"status {MSDA._E_UNSUPPORTED}" not in str(e):
            raise
        MSDA._fwd_algo = "auto"
        desc += " (forced kernel: unsupported)"
        out = MSDA.ms_deform_attn_forward(dev(value), dsh, dst, dev(loc), dev(attn), 1)
    gv, gl, ga = MSDA.ms_deform_attn_backward(dev(value), dsh, dst, dev(loc), dev(attn), dev(grad), 1)
    torch.cuda.synchronize()
    want = msda_oracle.forward(value, sh, st, loc, attn)
    wgv, wgl, wga = msda_oracle.backward(value, sh, st, loc, attn, grad)
    # grad_loc is discontinuous where a pixel coordinate crosses an integer (the bilinear cell
    # changes); fp32 and fp64 arithmetic can land on different sides, so samples within 1e-4 of a
    # crossing are left out of the grad_loc comparison (their other outputs are continuous and stay in)
    pix = loc.numpy() * sh.numpy()[None, None, None, :, None, ::-1] - 0.5
    with np.errstate(invalid="ignore"):
        near = (np.abs(pix - np.round(pix)) < 1e-4).any(-1, keepdims=True)
    near = np.broadcast_to(near, pix.shape)
    worst = 0.0
    for name, got, ref in (("out", out, want), ("grad_value", gv, wgv), ("grad_loc", gl, wgl), ("grad_attn", ga, wga)):
        got = got.double().cpu().numpy().reshape(ref.shape)
        ok = np.isfinite(ref)
        if name == "grad_loc":
            got = np.where(near, ref, got)
        if not np.array_equal(np.isfinite(got), ok):
            return f"FAIL {desc}: {name} finiteness differs"
        err = float(np.abs(got[ok] - ref[ok]).max()) if ok.any() else 0.0
        scale = max(1.0, float(np.abs(ref[ok]).max())) if ok.any() else 1.0
        worst = max(worst, err / (TOL[dtype] * scale))
        if err > TOL[dtype] * scale:
            return f"FAIL {desc}: {name} err {err:.3e} > {TOL[dtype]:.0e} * {scale:.3g}"
    return f"ok   {desc}  ({worst:.2f} of the bar)"


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    BIG = len(sys.argv) > 3 and "big" in sys.argv[3:]
    SORTED = "sorted" in sys.argv[3:]
    rng = random.Random(seed)
    fails = 0
    for i in range(n):
        try:
            r = one_case(rng, seed * 100000 + i)
        except Exception as e:          # noqa: BLE001
            r = f"FAIL #{i}: exception {type(e).__name__}: {e}"
        if r.startswith("FAIL"):
            fails += 1
            print(r, flush=True)
        elif i % (5 if BIG else 20) == 0:
            print(r, flush=True)
    print(f"{n - fails}/{n} cases within the bars", flush=True)
    sys.exit(1 if fails else 0)
