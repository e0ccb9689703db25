#!/bin/bash
# round 3, the closing visit: full GPU suite, bench lines of every workload, rocprofv3 trace + counters of the default
# command and of the other workloads, module-level lines.
mkdir -p gpurun_out
echo "== all gpu tests"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03bd_pytest_all.log 2>&1; tail -3 gpurun_out/r03bd_pytest_all.log | cut -c1-250
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], r.get("kernels_mean_us"), "frac", r.get("fwdbwd_hbm_frac"), "dom", r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"]["traffic"])
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
echo "== bench"
B="timeout 250 python bench.py"
$B --steps 20 --warmup 5 > gpurun_out/bench_r03bd_cfg2_northstar.json 2>gpurun_out/bench_r03bd.err; show gpurun_out/bench_r03bd_cfg2_northstar.json
B="timeout 200 python bench.py --no-cpu-baseline"
$B --steps 20 --warmup 5 > gpurun_out/bench_r03bd_cfg2_northstar_b.json 2>/dev/null; show gpurun_out/bench_r03bd_cfg2_northstar_b.json
$B --steps 100 --warmup 20 > gpurun_out/bench_r03bd_100.json 2>/dev/null; show gpurun_out/bench_r03bd_100.json
MMFS_TAPS_ALGO=vec MMFS_FWD_ALGO=vec MMFS_PREPARE_IN_TAPS=0 $B --steps 100 --warmup 20 > gpurun_out/bench_r03bd_rowgather.json 2>/dev/null; show gpurun_out/bench_r03bd_rowgather.json
$B --steps 30 --warmup 10 --fresh-levels > gpurun_out/bench_r03bd_fresh.json 2>/dev/null; show gpurun_out/bench_r03bd_fresh.json
$B --steps 30 --warmup 10 --loc-dist centre > gpurun_out/bench_r03bd_centre.json 2>/dev/null; show gpurun_out/bench_r03bd_centre.json
for w in cfg2_sd_real cfg5_llm_n4 cfg1 enc_injector enc_extractor; do
  $B --steps 30 --warmup 10 --workload $w > gpurun_out/bench_r03bd_$w.json 2>/dev/null; show gpurun_out/bench_r03bd_$w.json
done
$B --steps 30 --warmup 10 --workload cfg5_llm_n4 --loc-dist centre > gpurun_out/bench_r03bd_cfg5_llm_n4_centre.json 2>/dev/null; show gpurun_out/bench_r03bd_cfg5_llm_n4_centre.json
$B --steps 30 --warmup 10 --workload cfg5_llm_n4 --visible causal > gpurun_out/bench_r03bd_cfg5_llm_n4_causal.json 2>/dev/null; show gpurun_out/bench_r03bd_cfg5_llm_n4_causal.json
$B --steps 30 --warmup 10 --workload cfg5_llm_n4 --visible causal --loc-dist centre > gpurun_out/bench_r03bd_cfg5_llm_n4_causal_centre.json 2>/dev/null; show gpurun_out/bench_r03bd_cfg5_llm_n4_causal_centre.json
$B --steps 50 --warmup 50 --workload ref_speed_test --grad ones > gpurun_out/bench_r03bd_ref_speed_test_f16.json 2>/dev/null; show gpurun_out/bench_r03bd_ref_speed_test_f16.json
$B --steps 50 --warmup 50 --workload ref_speed_test --grad ones --dtype f32 > gpurun_out/bench_r03bd_ref_speed_test_f32.json 2>/dev/null; show gpurun_out/bench_r03bd_ref_speed_test_f32.json
echo "== rocprof (default command)"
bash tools/prof.sh r03bd > gpurun_out/prof_r03bd.log 2>&1; grep -A8 "== kernel stats" gpurun_out/prof_r03bd.log | cut -c1-170
echo "== rocprof (other workloads)"
for w in cfg2_sd_real cfg5_llm_n4 cfg1 ref_speed_test; do
  extra=""; [ $w = ref_speed_test ] && extra="--grad ones"
  bash tools/prof.sh r03bd_$w --workload $w $extra > gpurun_out/prof_r03bd_$w.log 2>&1; grep -A6 "== kernel stats" gpurun_out/prof_r03bd_$w.log | cut -c1-150
done
echo "== module bench"
timeout 900 python tools/module_bench.py cfg3 cfg4 > gpurun_out/r03bd_module_bench_cfg3_cfg4.jsonl 2>gpurun_out/r03bd_module_bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r03bd_module_bench_cfg3_cfg4.jsonl"):
    r = json.loads(l); print(r["config"], r["what"][30:], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"])
PY
tail -2 gpurun_out/r03bd_module_bench.err
