"""A workload's forward at 64 ... 4096 queries, per formulation: python tools/fwd_nq.py [workload]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
import bench


def timed(fn, n=60):
    for _ in range(15):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_northstar"
for nq in (64, 128, 256, 512, 1024, 2048, 4096):
    w = dict(bench.WORKLOADS[name]); w["Nq"] = nq
    value, shapes, start, loc, attn, grad = bench.make_inputs(w, "cuda", 0)
    line = []
    for algo in ("gather", "lds", "waves", "slices", "auto"):
        MSDA._fwd_algo = algo
        try:
            line.append("%s %.1f" % (algo, timed(lambda: MSDA.ms_deform_attn_forward(value, shapes, start, loc, attn, 1))))
        except RuntimeError:
            line.append("%s n/a" % algo)
    MSDA._fwd_algo = "auto"
    print("Nq %5d: %s us" % (nq, "; ".join(line)))
