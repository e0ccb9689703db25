"""Phase clocks of the LDS-resident grad_loc / grad_attn kernel (build: tools/exp_build.sh tapsprof "-DMMFS_PROFILE_TAPS";
run: MMFS_MSDA_LIB=.../build/exp/tapsprof.so python tools/taps_prof.py [workload])."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_northstar"
w = bench.WORKLOADS[name]
value, shapes, start, loc, attn, grad = bench.make_inputs(w, "cuda", 0)
bwd = lambda: MSDA.ms_deform_attn_backward(value, shapes, start, loc, attn, grad, 1)
for _ in range(3):
    bwd()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
MSDA._lib.mmfs_debug_taps_profile(buf, 1)
n = 10
for _ in range(n):
    bwd()
torch.cuda.synchronize()
MSDA._lib.mmfs_debug_taps_profile(buf, 0)
v = [x / n for x in buf]
B, Nq, H = w["B"], w["Nq"], w["H"]
steps = max(v[7], 1)
runs = B * H * 16 * max(1, -(-Nq // 256))
print("%s: %d wave-steps per call" % (name, v[7]))
for i, nm in ((0, "run setup + barrier (previous image free)"), (1, "image fill")):
    print("  %-56s %10.0f clk per (wave, run)   (x %d)" % (nm, v[i] / runs, runs))
for i, nm in ((2, "stage"), (3, "first issues + next requests"), (4, "matrix-core phase"), (5, "gather loop"), (6, "per-sample algebra + stores")):
    print("  %-56s %10.0f clk per wave-step" % (nm, v[i] / steps))
print("  sum of the step phases %.0f clk per wave-step; all phases %.0f clk per (wave, run)" % (sum(v[j] for j in (2, 3, 4, 5, 6)) / steps, sum(v[:7]) / runs))
