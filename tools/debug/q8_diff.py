"""Where do repeated runs of the sliced forward differ?  (MMFS_MSDA_LIB = an experimental build, e.g. the unrepaired
kernel: RAW=1 tools/exp_build1.sh q8_raw msda_fwd_q8 "-DQ8_PROBE" -- see tools/debug/q8_probe.py.)
Prints, for the SD block's shape in fp16 and bf16: how many elements of run i differ from run 0, the largest difference,
the same against the row gather, and where the differing elements sit (query mod 8, channel mod 32, head, level of ... )."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import MultiScaleDeformableAttention as MSDA
from test_stress_gpu import _inputs, SD_BLOCK, LLM_N4

for name, cfg in (("SD_BLOCK", SD_BLOCK), ("LLM_N4", LLM_N4)):
    for dtype in (torch.float16, torch.bfloat16):
        value, sh, st, loc, attn, _ = _inputs(cfg, dtype)
        MSDA._fwd_algo = "gather"
        ref = MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 1).float()
        MSDA._fwd_algo = "slices"
        outs = [MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 1) for _ in range(12)]
        torch.cuda.synchronize()
        MSDA._fwd_algo = "auto"
        o0 = outs[0].float()
        print(f"{name} {dtype}: run 0 vs row gather: max {float((o0 - ref).abs().max()):.3e}, finite {bool(torch.isfinite(o0).all())}")
        B, Nq, HD = o0.shape
        H = cfg[2]
        for i, o in enumerate(outs[1:], 1):
            d = (o.float() - o0)
            nz = d != 0
            n = int(nz.sum())
            if n == 0:
                continue
            idx = nz.nonzero()
            q, c = idx[:, 1], idx[:, 2]
            print(f"  run {i}: {n} elements differ, max {float(d.abs().max()):.3e}; queries mod 8: {torch.bincount(q % 8, minlength=8).tolist()}; "
                  f"channel mod 32 (first 8 bins of 4): {torch.bincount((c % 32) // 4, minlength=8).tolist()}; distinct (b, q): {len(set(zip(idx[:, 0].tolist(), q.tolist())))}")
            if i >= 3:
                break
