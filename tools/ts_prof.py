"""Phase clocks of the sorted-records grad_loc / grad_attn kernel (build: tools/exp_build1.sh tsprof msda_bwd_taps_sorted
"-DMMFS_PROFILE_TS"; run: MMFS_MSDA_LIB=.../build/exp/tsprof.so python tools/ts_prof.py [workload])."""
import ctypes, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "mm-interleaved_amd"))
sys.path.insert(0, ROOT)
import torch
import MultiScaleDeformableAttention as MSDA
from mmfs_amd.levels import make_level_tables

dev = "cuda"
which = sys.argv[1] if len(sys.argv) > 1 else "northstar"
B, H, D, Nq, P, pyr, nimg = {"northstar": (8, 8, 128, 4096, 4, [(64, 64), (32, 32), (16, 16), (8, 8)], 1),
                             "sd": (8, 16, 64, 4096, 8, [(64, 64), (32, 32), (16, 16), (8, 8)], 1),
                             "llm4": (4, 16, 64, 2048, 8, [(32, 32), (16, 16), (8, 8)], 4)}[which]
shapes, start, S = make_level_tables(pyr, nimg, dev)
L = len(pyr) * nimg
g = torch.Generator(device=dev).manual_seed(0)
value = torch.rand(B, S, H, D, device=dev, generator=g).bfloat16()
loc = torch.rand(B, Nq, H, L, P, 2, device=dev, generator=g).bfloat16()
attn = torch.rand(B, Nq, H, L, P, device=dev, generator=g)
attn = (attn / attn.sum((-1, -2), keepdim=True)).bfloat16()
grad = torch.randn(B, Nq, H * D, device=dev, generator=g).bfloat16()
for _ in range(3):
    MSDA.ms_deform_attn_backward(value, shapes, start, loc, attn, grad, 1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
MSDA._lib.mmfs_debug_ts_profile(buf, 1)
n = 10
for _ in range(n):
    MSDA.ms_deform_attn_backward(value, shapes, start, loc, attn, grad, 1)
torch.cuda.synchronize()
MSDA._lib.mmfs_debug_ts_profile(buf, 0)
v = list(buf)
items = max(v[8], 1)
names = ["descriptors", "value rows + first records", "first rows", "rounds", "drain"]
print("%s: items per call %.0f, records per item %.1f, rounds per item %.2f, steps per item %.2f"
      % (which, v[8] / n, v[11] / items, v[9] / items, v[10] / items))
for i, nm in enumerate(names):
    print("  %-28s %8.0f clk per item" % (nm, v[i] / items))
print("  total %.0f clk per item; rounds: %.0f clk per step" % (sum(v[:5]) / items, v[3] / max(v[10], 1)))
