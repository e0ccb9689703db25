"""Phase clocks of the workgroup-local grad_value kernel (build: tools/exp_build.sh gprof "-DMMFS_PROFILE_GV";
run: MMFS_MSDA_LIB=.../build/exp/gprof.so python tools/gv_prof.py [workload]).  Thread 0 of every workgroup
(wave 0's view: the barriers make it the workgroup's)."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd")]
import torch
import MultiScaleDeformableAttention as MSDA
import bench
from mmfs_amd.levels import make_level_tables

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2_northstar"
w = bench.WORKLOADS[name]
value, shapes, start, loc, attn, grad = bench.make_inputs(w, "cuda", 0)
shapes, start = make_level_tables(w["shapes"], w["n"], "cuda")[:2]
out = MSDA.ms_deform_attn_forward(value, shapes, start, loc, attn, 1)
bwd = lambda: MSDA.ms_deform_attn_backward(value, shapes, start, loc, attn, grad.reshape(out.shape), 1)
for _ in range(5):
    bwd()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
MSDA._lib.mmfs_debug_gv_profile(buf, 1)
n = 10
for _ in range(n):
    bwd()
torch.cuda.synchronize()
MSDA._lib.mmfs_debug_gv_profile(buf, 0)
v = [x / n for x in buf]
chunks = max(v[6], 1)
lv = MSDA.value_lds_levels(bench.DTYPES[w["dtype"]], w["shapes"] * w["n"], w["B"], w["H"], w["D"], w["Nq"], w["P"])
print("%s: levels %s in the kernel; %d (workgroup, chunk) pairs per call" % (name, lv, chunks))
names = ["setup (per workgroup)", "rows + decode + count", "prefix + place", "products", "epilogue: sums over virtual blocks, rows / partial tiles (per workgroup)",
         "partial tiles added up (per workgroup)"]
for i in (1, 2, 3):
    print("  %-80s %9.0f clk per chunk" % (names[i], v[i] / chunks))
print("  %-80s %9.0f clk per chunk" % ("all three", sum(v[1:4]) / chunks))
print("  setup + epilogue + partials, summed over workgroups: %.0f clk = %.1f %% of the chunk phases" % (v[0] + v[4] + v[5], 100 * (v[0] + v[4] + v[5]) / max(1, sum(v[1:4]))))
