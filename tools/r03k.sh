#!/bin/bash
# round 3, the consolidated visit: full GPU suite, bench lines of every workload, rocprofv3 trace + counters of the
# default command, counters of the other workloads (HBM traffic), module-level lines, phase clocks.
mkdir -p gpurun_out
echo "== all gpu tests"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03k_pytest_all.log 2>&1; tail -5 gpurun_out/r03k_pytest_all.log | cut -c1-250
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], r.get("kernels_mean_us"), "frac", r.get("fwdbwd_hbm_frac"), "dom", r["roofline"]["kernel"], r["roofline"]["frac"])
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
echo "== bench"
B="timeout 200 python bench.py"
$B --steps 20 --warmup 5 > gpurun_out/bench_r03k_cfg2_northstar.json 2>gpurun_out/bench_r03k.err; show gpurun_out/bench_r03k_cfg2_northstar.json
B="timeout 150 python bench.py --no-cpu-baseline"
$B --steps 100 --warmup 20 > gpurun_out/bench_r03k_100.json 2>/dev/null; show gpurun_out/bench_r03k_100.json
MMFS_TAPS_ALGO=vec MMFS_FWD_ALGO=vec $B --steps 100 --warmup 20 > gpurun_out/bench_r03k_rowgather.json 2>/dev/null; show gpurun_out/bench_r03k_rowgather.json
$B --steps 30 --warmup 10 --fresh-levels > gpurun_out/bench_r03k_fresh.json 2>/dev/null; show gpurun_out/bench_r03k_fresh.json
$B --steps 30 --warmup 10 --loc-dist centre > gpurun_out/bench_r03k_centre.json 2>/dev/null; show gpurun_out/bench_r03k_centre.json
for w in cfg2_sd_real cfg5_llm_n4 cfg1 enc_injector enc_extractor; do
  $B --steps 30 --warmup 10 --workload $w > gpurun_out/bench_r03k_$w.json 2>/dev/null; show gpurun_out/bench_r03k_$w.json
done
for nq in 64 256 1024; do $B --steps 30 --warmup 10 --nq $nq > gpurun_out/bench_r03k_cfg2_northstar_nq$nq.json 2>/dev/null; show gpurun_out/bench_r03k_cfg2_northstar_nq$nq.json; done
MMFS_FWD_ALGO=mma $B --steps 30 --warmup 10 --workload cfg2_sd_real > gpurun_out/bench_r03k_cfg2_sd_real_fwdlds.json 2>/dev/null; show gpurun_out/bench_r03k_cfg2_sd_real_fwdlds.json
MMFS_FWD_ALGO=mma $B --steps 30 --warmup 10 --workload cfg5_llm_n4 > gpurun_out/bench_r03k_cfg5_llm_n4_fwdlds.json 2>/dev/null; show gpurun_out/bench_r03k_cfg5_llm_n4_fwdlds.json
$B --steps 30 --warmup 10 --workload cfg5_llm_n4 --loc-dist centre > gpurun_out/bench_r03k_cfg5_llm_n4_centre.json 2>/dev/null; show gpurun_out/bench_r03k_cfg5_llm_n4_centre.json
$B --steps 30 --warmup 10 --workload cfg5_llm_n4 --visible causal > gpurun_out/bench_r03k_cfg5_llm_n4_causal.json 2>/dev/null; show gpurun_out/bench_r03k_cfg5_llm_n4_causal.json
$B --steps 30 --warmup 10 --workload cfg5_llm_n4 --visible causal --loc-dist centre > gpurun_out/bench_r03k_cfg5_llm_n4_causal_centre.json 2>/dev/null; show gpurun_out/bench_r03k_cfg5_llm_n4_causal_centre.json
$B --steps 30 --warmup 10 --visible causal --loc-dist centre > gpurun_out/bench_r03k_cfg2_northstar_centre_dummy.json 2>/dev/null
$B --steps 50 --warmup 50 --workload ref_speed_test --grad ones > gpurun_out/bench_r03k_ref_speed_test_f16.json 2>/dev/null; show gpurun_out/bench_r03k_ref_speed_test_f16.json
$B --steps 50 --warmup 50 --workload ref_speed_test --grad ones --dtype f32 > gpurun_out/bench_r03k_ref_speed_test_f32.json 2>/dev/null; show gpurun_out/bench_r03k_ref_speed_test_f32.json
echo "== rocprof (default command)"
bash tools/prof.sh r03k > gpurun_out/prof_r03k.log 2>&1; grep -A8 "== kernel stats" gpurun_out/prof_r03k.log | cut -c1-170
echo "== HBM traffic of the other workloads"
cd /tmp && export TMPDIR=/tmp
root=$GRAFT_REPO_ROOT
for w in cfg2_sd_real cfg5_llm_n4 cfg1; do
  mkdir -p $root/gpurun_out/prof_r03k_$w
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 120 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/prof_r03k_$w/$c -- python $root/bench.py --no-cpu-baseline --steps 6 --warmup 2 --workload $w > $root/gpurun_out/prof_r03k_$w/$c.log 2>&1
  done
  python3 - $w <<'PY'
import csv, glob, collections, re, os, sys
root = os.environ["GRAFT_REPO_ROOT"]; w = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/gpurun_out/prof_r03k_%s/**/*counter_collection.csv" % w, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        m = re.search(r"(msda_[a-z_]+)", k)
        if not m: continue
        short = m.group(1)
        fl = re.search(r"msda_bwd_vecI\w+?Li\d+ELb(\d)E", k)
        if fl: short += "_atomic" if fl.group(1) == "1" else "_taps"
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(root + "/gpurun_out/prof_r03k_%s/pmc_summary.txt" % w, "w") as fo:
    for k in sorted(agg):
        line = k + ": " + ", ".join("%s=%.4g" % (c, sum(x) / len(x)) for c, x in sorted(agg[k].items()))
        print(w, line); fo.write(line + "\n")
PY
done
cd $root
echo "== module bench"
timeout 900 python tools/module_bench.py cfg3 cfg4 > gpurun_out/r03k_module_bench_cfg3_cfg4.jsonl 2>gpurun_out/r03k_module_bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r03k_module_bench_cfg3_cfg4.jsonl"):
    r = json.loads(l); print(r["config"], r["what"][30:], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"], "mfma", r["gemm_mfma_util"], "op_frac", r["op_hbm_frac"])
PY
tail -2 gpurun_out/r03k_module_bench.err
