"""Phase clocks of the cell sort, per level (build: tools/exp_build.sh sprof "-DMMFS_PROFILE_SORT";
run: MMFS_MSDA_LIB=.../build/exp/sprof.so python tools/sort_prof.py)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mm-interleaved_amd"))
import torch
import MultiScaleDeformableAttention as MSDA

dev = "cuda"
B, H, D, Nq, P = 8, 8, 128, 4096, 4
lv = [(64, 64), (32, 32), (16, 16), (8, 8)]
if len(sys.argv) > 1 and sys.argv[1] == "injector":            # ViT-Adapter injector: 512 small slices
    B, H, D, Nq, P, lv = 32, 16, 32, 256, 4, [(32, 32), (16, 16), (8, 8)]
shapes = torch.tensor(lv, device=dev)
start = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
S, L = int(shapes.prod(1).sum()), len(lv)
g = torch.Generator(device=dev).manual_seed(0)
value = torch.rand(B, S, H, D, device=dev, generator=g).bfloat16()
loc = torch.rand(B, Nq, H, L, P, 2, device=dev, generator=g).bfloat16()
attn = torch.rand(B, Nq, H, L, P, device=dev, generator=g)
attn = (attn / attn.sum((-1, -2), keepdim=True)).bfloat16()
grad = torch.randn(B, Nq, H * D, device=dev, generator=g).bfloat16()
for _ in range(3):
    MSDA.ms_deform_attn_backward(value, shapes, start, loc, attn, grad, 1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
MSDA._lib.mmfs_debug_sort_profile(buf, 1)
n = 10
for _ in range(n):
    MSDA.ms_deform_attn_backward(value, shapes, start, loc, attn, grad, 1)
torch.cuda.synchronize()
MSDA._lib.mmfs_debug_sort_profile(buf, 0)
names = ["clear counters", "count scan", "prefix sum", "scatter scan", "cell table", "arrive", "plan blocks (last wg)"]
wgs = n * B * H
for lvl in range(L):
    row = [buf[lvl * 8 + i] / wgs for i in range(7)]
    print("level %d: " % lvl + "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, row)) + "   total %.0f clk" % sum(row))

# timeline of the LAST launch: when each workgroup started / finished its own tile / left (us from the first start)
tl = (ctypes.c_ulonglong * (4096 * 3))()
MSDA._lib.mmfs_debug_sort_timeline(tl)
rows = [(tl[3 * i], tl[3 * i + 1], tl[3 * i + 2]) for i in range(4096) if tl[3 * i]]
t0 = min(r[0] for r in rows)
rows = sorted(((a - t0) / 100.0, (b - t0) / 100.0, (c - t0) / 100.0) for a, b, c in rows)
print("%d workgroups; starts: first %.1f median %.1f last %.1f us; last own-work end %.1f us; last exit %.1f us" % (
    len(rows), rows[0][0], rows[len(rows) // 2][0], rows[-1][0], max(r[1] for r in rows), max(r[2] for r in rows)))
dur = sorted(r[1] - r[0] for r in rows)
print("own-work duration: min %.1f median %.1f max %.1f us" % (dur[0], dur[len(dur) // 2], dur[-1]))
for q in range(0, len(rows), max(1, len(rows) // 16)):
    print("  wg@%4d start %.1f own %.1f exit %.1f" % (q, *rows[q]))
