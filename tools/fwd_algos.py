"""Forward only, every formulation the shim can force, at the named workloads: us per call (HIP events, 30 calls)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
import bench

for name in (sys.argv[1:] or ["cfg2_sd_real", "cfg5_llm_n4", "cfg3_llm_n1"]):
    w = bench.WORKLOADS[name]
    value, shapes, start, loc, attn, grad = bench.make_inputs(w, "cuda", 0)
    ref = None
    for algo in ("auto", "gather", "lds", "slices", "waves"):
        MSDA._fwd_algo = algo
        try:
            fwd = lambda: MSDA.ms_deform_attn_forward(value, shapes, start, loc, attn, 1)
            for _ in range(5):
                out = fwd()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                out = fwd()
            e1.record(); torch.cuda.synchronize()
            if ref is None:
                ref = out.float()
            print("%-14s %-7s %8.1f us   max |out - auto| %.2e" % (name, algo, e0.elapsed_time(e1) / 30 * 1e3, float((out.float() - ref).abs().max())))
        except Exception as e:      # noqa: BLE001
            print("%-14s %-7s refused: %s" % (name, algo, str(e)[:80]))
MSDA._fwd_algo = "auto"
