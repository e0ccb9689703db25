#!/bin/bash
# Register use of every kernel of the CURRENT sources: regenerates build/*.s and build/*.remarks (`make asm`; plain `make`
# does NOT -- a stale remarks file once hid 80 spilled VGPRs, profiles/r03_experiments.md r03bj) and lists the kernels
# that spill or use scratch.
cd "$(dirname "$0")/../mm-interleaved_amd/csrc" && make asm > /dev/null 2>&1
python3 - <<'PY'
import glob, re
bad = 0
for f in sorted(glob.glob("build/*.remarks")):
    name = None
    for line in open(f, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m: name = m.group(1); vg = None
        m = re.search(r"\bVGPRs: (\d+)", line)
        if m: vg = m.group(1)
        m = re.search(r"(ScratchSize \[bytes/lane\]|VGPRs Spill): (\d+)", line)
        if m and int(m.group(2)) > 0:
            print("%-28s %-14s %4s   %s (VGPRs %s)" % (f[6:-8], m.group(1)[:12], m.group(2), name[:90], vg)); bad += 1
print("kernels with VGPR spills / scratch:", bad)
PY
