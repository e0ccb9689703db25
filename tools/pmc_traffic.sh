#!/bin/bash
# HBM-side traffic per kernel of one bench workload: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes
# usage: tools/pmc_traffic.sh tag workload [env...]   -> gpurun_out/prof_<tag>_<workload>/pmc_summary.txt
tag=$1; w=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
mkdir -p $root/gpurun_out/prof_${tag}_$w
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
  timeout 180 env "$@" rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/prof_${tag}_$w/$c -- python $root/bench.py --no-cpu-baseline --steps 6 --warmup 2 --workload $w > $root/gpurun_out/prof_${tag}_$w/$c.log 2>&1
done
python3 - $tag $w <<'PY'
import csv, glob, collections, re, os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()); tag, w = sys.argv[1:3]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/gpurun_out/prof_%s_%s/**/*counter_collection.csv" % (tag, w), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        m = re.search(r"(msda_[a-z_]+)", k)
        if not m: continue
        agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(root + "/gpurun_out/prof_%s_%s/pmc_summary.txt" % (tag, w), "w") as fo:
    for k in sorted(agg):
        c = {n: sum(x) / len(x) for n, x in agg[k].items()}
        extra = ""
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:       # KB units; FETCH x2 on gfx950 (MI355X_MICROARCH.md)
            extra = "  -> traffic %.1f MB" % ((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 / 1e6)
        line = k + ": " + ", ".join("%s=%.4g" % (n, v) for n, v in sorted(c.items())) + extra
        print(w, line); fo.write(line + "\n")
PY
