"""Which part of a wrong row of the sliced forward is missing?  (MMFS_MSDA_LIB = the unrepaired kernel, tools/debug/q8_probe.py.)
value = 1 everywhere, every sample at the same bilinear fractions (Q8_FX, Q8_FY) inside its map, the attention weight of sample k
= (k + 1) / 1024 (Q8_FLAT: 1 / 64): with FX = 0.25, FY = 0.125 the four corner weights are 21, 7, 3, 1 thirty-seconds of the
sample's weight, every subset has its own sum, and the deficit of a wrong row says which corners of which samples it lost --
round 6: corner 2 (fy * gx) of ALL EIGHT samples of one pass, queries 6 and 7 of a tile, both the matrix-core and the row-gather
levels: the staging's packed multiply, not the products.  Q8_ONLY = resident / gather zeroes the other levels' weights."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "mm-interleaved_amd"), os.path.join(ROOT, "tests")]
import torch
import MultiScaleDeformableAttention as MSDA
from collections import Counter

B, Nq, H, D, P = 8, 4096, 16, 64, 8
shapes = [(64, 64), (32, 32), (16, 16), (8, 8)]
dt = torch.float16
sh = torch.tensor(shapes, dtype=torch.long, device="cuda")
st = torch.cat((sh.new_zeros(1), sh.prod(1).cumsum(0)[:-1]))
S, L = int(sh.prod(1).sum()), len(shapes)
g = torch.Generator(device="cuda").manual_seed(0)
value = torch.ones(B, S, H, D, device="cuda", dtype=dt)
# every sample exactly ON a pixel centre: bilinear weights (1, 0, 0, 0), a lost sample k is a deficit of exactly (k + 1) / 1024
FX, FY = float(os.environ.get("Q8_FX", "0")), float(os.environ.get("Q8_FY", "0"))
loc = torch.empty(B, Nq, H, L, P, 2, device="cuda")
for l, (Hl, Wl) in enumerate(shapes):
    loc[:, :, :, l, :, 0] = (torch.randint(1, Wl - 2, (B, Nq, H, P), device="cuda", generator=g).float() + 0.5 + FX) / Wl
    loc[:, :, :, l, :, 1] = (torch.randint(1, Hl - 2, (B, Nq, H, P), device="cuda", generator=g).float() + 0.5 + FY) / Hl
loc = loc.to(dt)
ONLY = os.environ.get("Q8_ONLY")           # "resident" / "gather": the other levels' weights are zero
MSDA._fwd_algo = "slices"
K = L * P
# weights 2^-k would underflow: sample k carries (k + 1) / 1024 -- a missing sample k shows as a deficit of (k + 1) / 1024
attn = ((torch.arange(K, device="cuda") + 1).float() / 1024 if os.environ.get("Q8_FLAT") is None else torch.full((K,), 1 / 64, device="cuda")).to(dt).view(1, 1, 1, L, P).expand(B, Nq, H, L, P).contiguous()
if ONLY == "resident": attn[:, :, :, 0] = 0
if ONLY == "gather": attn[:, :, :, 1:] = 0
want = float(attn[0, 0, 0].float().sum())
deficits, cols = Counter(), Counter()
for run in range(4):
    out = MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 1).float().view(B, Nq, H, D)
    torch.cuda.synchronize()
    err = out - want
    bad = err.abs() > 1e-4
    rows = bad.any(-1)
    idx = rows.nonzero()
    print(f"run {run}: {int(rows.sum())} wrong rows; q mod 8 {torch.bincount(idx[:, 1] % 8, minlength=8).tolist()}; "
          f"wave (q // 8 % 16) {torch.bincount(idx[:, 1] // 8 % 16, minlength=16).tolist()}")
    for b, q, h in idx[:3000].tolist():
        e = err[b, q, h]
        cols["".join("x" if bool(bad[b, q, h, 16 * i:16 * i + 16].all()) else ("." if not bool(bad[b, q, h, 16 * i:16 * i + 16].any()) else "p") for i in range(4))] += 1
        for v in set(round(float(x) * 1024, 2) for x in e[bad[b, q, h]]):
            deficits[v] += 1
print("which 16-channel pieces of a wrong row are wrong (x all, p some, . none):", cols.most_common(8))
print("deficit * 1024 (= -(k + 1) if sample k is missing):", sorted(deficits.items(), key=lambda kv: -kv[1])[:40])
