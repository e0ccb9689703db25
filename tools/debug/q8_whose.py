"""Whose weights does a wrong row of the sliced forward carry?  value = 1 everywhere, every sample well inside its map, and the
attention weights of query q all equal to (q % 64 + 1) / 4096: out[b, q, :] = K * (q % 64 + 1) / 4096 exactly (fp32 sums of fp16
values) -- a wrong row names the query whose weights it was multiplied with.  (MMFS_MSDA_LIB = the -DQ8_BUILTIN_MFMA build.)"""
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
loc = (torch.rand(B, Nq, H, L, P, 2, device="cuda", generator=g) * 0.6 + 0.2).to(dt)
qv = ((torch.arange(Nq, device="cuda") % 64 + 1).float() / 4096).to(dt)
attn = qv.view(1, Nq, 1, 1, 1).expand(B, Nq, H, L, P).contiguous()
MSDA._fwd_algo = "slices"
K = L * P
want = (qv.float() * K).view(1, Nq, 1)
hist = Counter()
for run in range(6):
    out = MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 1).float()
    torch.cuda.synchronize()
    rows = out.view(B, Nq, H, D)
    bad = (rows - want.view(1, Nq, 1, 1)).abs().max(-1).values > 2e-3 * want.view(1, Nq, 1)
    idx = bad.nonzero()
    print(f"run {run}: {int(bad.sum())} wrong (b, q, h) rows of {bad.numel()}; q mod 8: {torch.bincount(idx[:, 1] % 8, minlength=8).tolist()}")
    for b, q, h in idx[:4000].tolist():
        got = float(rows[b, q, h].mean()) * 4096 / K          # the (q' % 64 + 1) whose weights these are (if one query's)
        hist[round(got - (q % 64 + 1), 2)] += 1
print("wrong row's implied (q' - q), most common:", hist.most_common(12))
