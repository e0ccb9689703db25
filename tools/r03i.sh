#!/bin/bash
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], r.get("kernels_mean_us"), "frac", r.get("fwdbwd_hbm_frac"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
B="timeout 150 python bench.py --no-cpu-baseline"
$B --steps 50 --warmup 10 > gpurun_out/bench_r03i.json 2>gpurun_out/bench_r03i.err; show gpurun_out/bench_r03i.json
for v in ntrecs ntstore ntboth; do
MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/$v.so $B --steps 50 --warmup 10 > gpurun_out/bench_r03i_$v.json 2>/dev/null; show gpurun_out/bench_r03i_$v.json
done
$B --steps 50 --warmup 10 > gpurun_out/bench_r03i_again.json 2>/dev/null; show gpurun_out/bench_r03i_again.json
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for v in base ntboth; do
  lib=$root/mm-interleaved_amd/libmmfs_msda.so; [ $v = base ] || lib=$root/mm-interleaved_amd/csrc/build/exp/$v.so
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    MMFS_MSDA_LIB=$lib timeout 120 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/pmc_r03i_${v}_$tag -- python $root/bench.py --no-cpu-baseline --steps 6 --warmup 2 > $root/gpurun_out/pmc_r03i_${v}_$tag.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, re, os
root = os.environ["GRAFT_REPO_ROOT"]
for v in ("base", "ntboth"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(root + "/gpurun_out/pmc_r03i_%s_*/**/*counter_collection.csv" % v, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(msda_[a-z_]+)", r.get("Kernel_Name", ""))
            if m: agg[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print(v, k, ", ".join("%s=%.4g" % (c, sum(x) / len(x)) for c, x in sorted(agg[k].items())))
PY
