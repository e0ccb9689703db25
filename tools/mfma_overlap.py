"""Scan the kernels' ISA (make -C mm-interleaved_amd/csrc asm -> build/*.s) for matrix-core products whose destination
registers overlap their A or B operand's (profiles/r04_experiments.md r04g-m, r04zs) -- python tools/mfma_overlap.py"""
import glob, os, re
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
pat = re.compile(r'v_mfma_\w+\s+(a|v)\[(\d+):(\d+)\],\s*(a|v)\[(\d+):(\d+)\],\s*(a|v)\[(\d+):(\d+)\],')
for f in sorted(glob.glob(os.path.join(ROOT, "mm-interleaved_amd", "csrc", "build", "*.s"))):
    n = ov = 0
    first = ""
    for line in open(f, errors="ignore"):
        m = pat.search(line)
        if not m:
            continue
        n += 1
        dk, d0, d1, ak, a0, a1, bk, b0, b1 = m.groups()
        d0, d1, a0, a1, b0, b1 = map(int, (d0, d1, a0, a1, b0, b1))
        hit = (dk == ak and not (d1 < a0 or a1 < d0)) or (dk == bk and not (d1 < b0 or b1 < d0))
        ov += hit
        if hit and not first:
            first = line.strip()
    if n:
        print("%-22s %3d products, destination on an operand in %3d   %s" % (os.path.basename(f), n, ov, first))
