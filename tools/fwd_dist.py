"""The forward alone, per formulation and location distribution (bench.py's workloads and --loc-dist):
python tools/fwd_dist.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_northstar"
w = bench.WORKLOADS[name]


def timed(fn, n=40):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for dist in ("uniform", "centre"):
    value, shapes, start, loc, attn, grad = bench.make_inputs(w, "cuda", 0, loc_dist=dist)
    line = []
    for algo in ("gather", "lds", "waves", "slices"):
        MSDA._fwd_algo = algo
        try:
            t = timed(lambda: MSDA.ms_deform_attn_forward(value, shapes, start, loc, attn, 1))
            line.append("%s %.1f us" % (algo, t))
        except RuntimeError as e:
            line.append("%s n/a" % algo)
    MSDA._fwd_algo = "auto"
    print("%s, %s: %s" % (name, dist, "; ".join(line)))
