"""Localise a parity failure of the sliced forward: per-query / per-channel error against the row-gather kernel."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import MultiScaleDeformableAttention as MSDA
from helpers import make_inputs

def run(x, dtype, algo):
    dev = lambda t: t.to("cuda", dtype) if t.is_floating_point() else t.to("cuda")
    MSDA._fwd_algo = algo
    o = MSDA.ms_deform_attn_forward(dev(x["value"]), dev(x["shapes"]), dev(x["start"]), dev(x["loc"]), dev(x["attn"]), 1)
    torch.cuda.synchronize()
    return o.double().cpu().numpy()

PYR = [(64, 64), (32, 32), (16, 16), (8, 8)]
CASES = {
    "pyramid, 1 head, 1 slice, 8 q": (1, 1, 32, 8, 4, PYR),
    "pyramid, 1 head, 1 slice, 64 q": (1, 1, 32, 64, 4, PYR),
    "pyramid, 1 head, 4 slices, 64 q": (1, 1, 128, 64, 4, PYR),
    "pyramid, 8 heads, 1 slice, 64 q": (1, 8, 32, 64, 4, PYR),
    "pyramid, 8 heads, 4 slices, 8 q": (1, 8, 128, 8, 4, PYR),
    "north star": (1, 8, 128, 64, 4, PYR),
    "north star again": (1, 8, 128, 64, 4, PYR),
    "3 resident levels only, 8 heads 4 slices": (1, 8, 128, 64, 4, PYR[1:]),
    "64^2 only (gather), 8 heads 4 slices": (1, 8, 128, 64, 4, PYR[:1]),
}
for name, (B, H, D, Nq, P, shapes) in CASES.items():
    x = make_inputs(B, H, D, Nq, P, shapes, seed=21, loc_range=(0.05, 0.95), dtype=torch.float16)
    a, g = run(x, torch.float16, "slices").reshape(B, Nq, H, D), run(x, torch.float16, "gather").reshape(B, Nq, H, D)
    a2 = run(x, torch.float16, "slices").reshape(B, Nq, H, D)
    err = np.abs(a - g)
    print("%-42s max err %.3e  (second run differs: %s)" % (name, err.max(), bool((a != a2).any())), end="")
    if err.max() > 2e-3:
        bad = err.reshape(B, Nq, H, D // 32, 32).max(-1) > 2e-3
        bq = np.argwhere(bad)
        print("  %d bad (b, q, h, slice); q mod 8: %s; slices: %s; heads: %s; first: %s" % (
            len(bq), sorted(set((bq[:, 1] % 8).tolist())), sorted(set(bq[:, 3].tolist())), sorted(set(bq[:, 2].tolist())), bq[:6].tolist()))
    else:
        print()
