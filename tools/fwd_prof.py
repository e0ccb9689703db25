"""Phase clocks of the LDS-resident forward (build: tools/exp_build.sh fprof "-DMMFS_PROFILE_FWD";
run: MMFS_MSDA_LIB=.../build/exp/fprof.so python tools/fwd_prof.py [workload])."""
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
MSDA._lib.mmfs_debug_fwd_profile(buf, 1)
n = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    fwd()
e1.record(); torch.cuda.synchronize()
MSDA._lib.mmfs_debug_fwd_profile(buf, 0)
v = [x / n for x in buf]
B, Nq, H = w["B"], w["Nq"], w["H"]
steps = max(v[7], 1)
print("%s: %.1f us per forward (instrumented); %d wave-steps per call" % (name, e0.elapsed_time(e1) / n * 1e3, v[7]))
names = ["barrier + image writes (per wave and run)", "stage", "first issues + next requests", "matrix-core phase", "gather loop",
         "run setup + image requests (per wave and run; + the general gather path)", "store"]
runs = B * H * 16 * max(1, -(-Nq // 256))
for i in (5, 0):
    print("  %-72s %10.0f clk per (wave, run)   (x %d)" % (names[i], v[i] / runs, runs))
for i in (1, 2, 3, 4, 6):
    print("  %-72s %10.0f clk per wave-step" % (names[i], v[i] / steps))
print("  sum of the step phases %.0f clk per wave-step; all phases %.0f clk per (wave, run)" % (sum(v[j] for j in (1, 2, 3, 4, 6)) / steps, sum(v[:7]) / runs))
