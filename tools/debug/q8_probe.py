"""The sliced forward's round-4 heisenbug, caught in the act (profiles/r06_experiments.md).  Reads the staging probe of a -DQ8_PROBE
build of msda_fwd_q8 (MMFS_MSDA_LIB) after a few forwards at the image decoder's shape with fixed bilinear fractions: what did a lane store for corner 2 of its sample, what should it have stored, and what
does the register that held the weight hold now?

    RAW=1 tools/exp_build1.sh q8_raw msda_fwd_q8 "-DQ8_PROBE"      # what hipcc alone makes of the kernel: ~14000 wrong rows per run,
                                                                   # every logged lane in 48..63, corner 2's weight (fy * gx * a) == 0
    tools/exp_build1.sh q8_fixed_probe msda_fwd_q8 "-DQ8_PROBE"    # the library's build (tools/fix_pk_opsel.py): 0
    MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/q8_raw.so python tools/debug/q8_probe.py"""
import os, sys, ctypes, struct
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
FX, FY = 0.25, 0.125
loc = torch.empty(B, Nq, H, L, P, 2, device="cuda")
for l, (Hl, Wl) in enumerate(shapes):
    loc[:, :, :, l, :, 0] = (torch.randint(1, Wl - 2, (B, Nq, H, P), device="cuda", generator=g).float() + 0.5 + FX) / Wl
    loc[:, :, :, l, :, 1] = (torch.randint(1, Hl - 2, (B, Nq, H, P), device="cuda", generator=g).float() + 0.5 + FY) / Hl
loc = loc.to(dt)
K = L * P
attn = torch.full((B, Nq, H, L, P), 1 / 64, device="cuda", dtype=dt)
MSDA._fwd_algo = "slices"
lib = ctypes.CDLL(os.environ["MMFS_MSDA_LIB"])
buf = (ctypes.c_uint * (4096 * 8))()
n = ctypes.c_uint(0)
f = lambda u: struct.unpack("f", struct.pack("I", u))[0]
for run in range(3):
    out = MSDA.ms_deform_attn_forward(value, sh, st, loc, attn, 1).float().view(B, Nq, H, D)
    torch.cuda.synchronize()
    bad = ((out - 0.5).abs() > 1e-4).any(-1)
    rc = lib.mmfs_debug_q8_probe(buf, ctypes.byref(n), 1)
    print(f"run {run}: wrong rows {int(bad.sum())}; probe rc {rc}, mismatches logged {n.value}")
    lanes, kinds = Counter(), Counter()
    for i in range(min(n.value, 4096)):
        o = buf[8 * i:8 * i + 8]
        lanes[o[0] // 16] += 1
        kinds[(o[1] >> 24, hex(o[2]), hex(o[3]), f(o[4]), f(o[5]), f(o[6]), f(o[7]))] += 1
    print("  lane group:", dict(lanes))
    for k, c in kinds.most_common(8):
        print(f"  x{c}: resident {k[0]} want {k[1]} got {k[2]} | w2 register now {k[3]:.6g}, fy {k[4]:.4g} gx {k[5]:.4g} aa {k[6]:.6g}")
