import os, sys, itertools
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

LV = {"8": [(8, 8)], "32": [(32, 32)], "32,16,8": [(32, 32), (16, 16), (8, 8)], "16,8": [(16, 16), (8, 8)]}
for lv, H, Nq, P, D in itertools.product(LV, (1, 2, 8), (8, 16, 64), (4, 8), (32,)):
    fails, where = 0, set()
    for rep in range(4):
        x = make_inputs(1, H, D, Nq, P, LV[lv], seed=21 + rep, loc_range=(0.05, 0.95), dtype=torch.float16)
        g = run(x, torch.float16, "gather").reshape(Nq, H, D)
        for _ in range(3):
            a = run(x, torch.float16, "slices").reshape(Nq, H, D)
            bad = np.argwhere(np.abs(a - g).max(-1) > 2e-3)
            if len(bad):
                fails += 1
                where |= set((int(q) % 8, int(q) // 8) for q, h in bad)
    if fails:
        print("levels %-8s H=%d Nq=%-3d P=%d: %2d of 12 runs wrong; (q mod 8, wave): %s" % (lv, H, Nq, P, fails, sorted(where)))
print("scan done")
