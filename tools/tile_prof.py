"""Phase clocks of the matrix-core grad_value reduce (build: tools/exp_build.sh tprof "-DMMFS_PROFILE_TILE";
run: MMFS_MSDA_LIB=.../build/exp/tprof.so python tools/tile_prof.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mm-interleaved_amd"))
import torch
import MultiScaleDeformableAttention as MSDA

dev = "cuda"
B, H, D, Nq, P = 8, 8, 128, 4096, 4
shapes = torch.tensor([(64, 64), (32, 32), (16, 16), (8, 8)], device=dev)
start = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
S, L = int(shapes.prod(1).sum()), 4
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
MSDA._lib.mmfs_debug_tile_profile(buf, 1)
n = 10
for _ in range(n):
    MSDA.ms_deform_attn_backward(value, shapes, start, loc, attn, grad, 1)
torch.cuda.synchronize()
MSDA._lib.mmfs_debug_tile_profile(buf, 0)
v = list(buf)
items = max(v[8], 1)
names = ["descriptor", "clear LDS + first records", "first rows", "rounds", "drain", "epilogue"]
print("items per call %.0f, rounds per item %.2f, steps per item %.2f" % (v[8] / n, v[9] / items, v[10] / items))
for i, nm in enumerate(names):
    print("  %-28s %8.0f clk per item" % (nm, v[i] / items))
print("  total %.0f clk per item" % (sum(v[:6]) / items))
npart = max(v[7], 1)
print("  epilogue of the %.0f items per call that leave a partial tile: %.0f clk each; of the %.0f whole blocks: %.0f clk each"
      % (v[7] / n, v[6] / npart, (v[8] - v[7]) / n, v[11] / max(v[8] - v[7], 1)))
