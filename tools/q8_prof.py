"""Phase clocks of the sliced forward (build: tools/exp_build.sh q8prof "-DMMFS_PROFILE_Q8";
run: MMFS_MSDA_LIB=.../build/exp/q8prof.so python tools/q8_prof.py [workload ...])."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
import bench

MSDA._fwd_algo = "slices"
for name in (sys.argv[1:] or ["cfg2_northstar"]):
    w = bench.WORKLOADS[name]
    value, shapes, start, loc, attn, grad = bench.make_inputs(w, "cuda", 0)
    fwd = lambda: MSDA.ms_deform_attn_forward(value, shapes, start, loc, attn, 1)
    for _ in range(5):
        fwd()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    MSDA._lib.mmfs_debug_q8_profile(buf, 1)
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fwd()
    e1.record(); torch.cuda.synchronize()
    MSDA._lib.mmfs_debug_q8_profile(buf, 0)
    v = [x / n for x in buf]
    passes, kblocks = max(v[7], 1), max(v[6], 1)
    runs = w["B"] * w["H"] * (w["D"] // 32) * max(1, -(-w["Nq"] // 512)) * 16
    print("%s: %.1f us per forward (instrumented); %d wave-passes, %d resident K-blocks per call" % (name, e0.elapsed_time(e1) / n * 1e3, v[7], v[6]))
    print("  table / barrier + image fill   %9.0f clk per (wave, run)  (x %d)" % (v[0] / runs, runs))
    for i, nm in ((1, "stage"), (2, "first row requests"), (3, "resident K-blocks"), (4, "row gather"), (5, "epilogue")):
        print("  %-30s %9.0f clk per wave-pass" % (nm, v[i] / passes))
    print("  resident K-blocks              %9.0f clk per K-block" % (v[3] / kblocks))
    print("  all phases %.0f clk per wave-pass, %.0f per (wave, run)" % (sum(v[1:6]) / passes, sum(v[:6]) / runs))
