#!/bin/bash
# Memory-path counters of the op's kernels (texture addresser / data, vector L1, L2), one rocprofv3 --pmc
# pass per group (never combined with trace domains).  usage: tools/pmc_mem.sh <tag> [bench args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/pmcmem_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--no-cpu-baseline --steps 4 --warmup 1 $*"
i=0
# at most 2 counters of the texture blocks (TA / TD) and 4 of TCP / TCC per pass: more is refused
# ("exceeds the capabilities of the hardware") and the refused run then hangs until its timeout
for grp in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_REQ_sum" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 75 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python $root/bench.py $args > $out/p$i.log 2>&1 || echo "pass $i failed: $(tail -2 $out/p$i.log)"
done
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        m = re.search(r"(msda_[a-z_]+)", k)
        if not m: continue
        short = m.group(1)
        fl = re.search(r"msda_bwd_vecI\w+?Li\d+ELb(\d)E", k)
        if fl: short += "_atomic" if fl.group(1) == "1" else "_taps"
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(agg):
        fo.write(k + ":\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            fo.write("    %-44s %14.4g   (mean of %d dispatches)\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/summary.txt").read()[:200])
PY
