"""Phase clocks of the wave-per-query forward (build: tools/exp_build.sh wqprof "-DMMFS_PROFILE_WQ";
run: MMFS_MSDA_LIB=.../build/exp/wqprof.so python tools/wq_prof.py [workload])."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_northstar"
w = bench.WORKLOADS[name]
value, shapes, start, loc, attn, grad = bench.make_inputs(w, "cuda", 0)
fwd = lambda: MSDA.ms_deform_attn_forward(value, shapes, start, loc, attn, 1)
for _ in range(5):
    fwd()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
MSDA._lib.mmfs_debug_wq_profile(buf, 1)
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    fwd()
e1.record(); torch.cuda.synchronize()
MSDA._lib.mmfs_debug_wq_profile(buf, 0)
v = [x / n for x in buf]
B, Nq, H = w["B"], w["Nq"], w["H"]
steps = max(v[7], 1)
print("%s: %.1f us per forward (instrumented); %d wave-steps (4 queries x 16 samples) per call" % (name, e0.elapsed_time(e1) / n * 1e3, v[7]))
runs = B * H * 16 * max(1, -(-Nq // 256))
print("  %-60s %10.0f clk per (wave, run)   (x %d)" % ("run setup + barrier (previous image free)", v[0] / runs, runs))
print("  %-60s %10.0f clk per (wave, run)" % ("wait for the image (after staging the first group) + barrier", v[4] / runs))
for i, nm in ((1, "stage + next requests"), (2, "products (4 queries; single-chunk: + epilogue)"), (3, "epilogue (multi-chunk)")):
    print("  %-60s %10.0f clk per wave-step" % (nm, v[i] / steps))
print("  sum of the step phases %.0f clk per wave-step; all phases %.0f clk per (wave, run)" % (sum(v[j] for j in (1, 2, 3)) / steps, sum(v[:7]) / runs))
