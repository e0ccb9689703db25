#!/bin/bash
# Profiles bench.py on the GPU box.  Outputs (CSV) land in gpurun_out/prof_<tag>/.
#   tools/prof.sh <tag> [bench args...]
# Passes: (1) kernel trace + stats; (2..) PMC counter groups, each in its own run
# (never combined with the trace domains gpurun refuses).
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# the trace pass runs the default command (20 warm-up + 100 timed steps: its per-kernel averages are
# the ones bench.py's HIP-event timings must agree with); the counter passes are short
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $root/bench.py --no-cpu-baseline $* > $out/trace.log 2>&1
args="--no-cpu-baseline --steps 10 --warmup 3 $*"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $out/pmc$i -- python $root/bench.py $args > $out/pmc$i.log 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== kernel stats", f)
    for r in list(csv.DictReader(open(f)))[:12]:
        print("  %-90s calls=%s avg_ns=%s pct=%s" % (r.get("Name","")[:90], r.get("Calls"), r.get("AverageNs"), r.get("Percentage")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "msda" not in k and "repack" not in k: continue
        import re
        m = re.search(r"(msda_[a-z_0-9]+|repack_kernel<\d+>|repack_kernel)", k)
        short = m.group(1) if m else k[:60]
        flags = re.search(r"msda_bwd_vecI\w+?Li\d+ELb(\d)E", k)          # first bool = SCATTER
        if flags: short += "_atomic" if flags.group(1) == "1" else "_taps"
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_summary.txt", "w") as fo:
    for k, cs in sorted(agg.items()):
        line = k + ": " + ", ".join("%s=%.4g" % (c, sum(v)/len(v)) for c, v in sorted(cs.items()))
        print(line); fo.write(line + "\n")
PY
