#!/bin/bash
mkdir -p gpurun_out
echo "== all gpu tests"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03c_pytest_all.log 2>&1; tail -12 gpurun_out/r03c_pytest_all.log | cut -c1-250
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], r.get("kernels_mean_us"), "frac", r.get("fwdbwd_hbm_frac"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
echo "== bench"
B="timeout 150 python bench.py --no-cpu-baseline"
$B --steps 50 --warmup 10 > gpurun_out/bench_r03c.json 2>gpurun_out/bench_r03c.err; show gpurun_out/bench_r03c.json
$B --steps 20 --warmup 5 > gpurun_out/bench_r03c_driver.json 2>/dev/null; show gpurun_out/bench_r03c_driver.json
$B --steps 30 --warmup 10 --fresh-levels > gpurun_out/bench_r03c_fresh.json 2>/dev/null; show gpurun_out/bench_r03c_fresh.json
$B --steps 30 --warmup 10 --loc-dist centre > gpurun_out/bench_r03c_centre.json 2>/dev/null; show gpurun_out/bench_r03c_centre.json
for w in cfg2_sd_real cfg5_llm_n4 cfg1 enc_injector enc_extractor; do
  $B --steps 30 --warmup 10 --workload $w > gpurun_out/bench_r03c_$w.json 2>/dev/null; show gpurun_out/bench_r03c_$w.json
done
$B --steps 30 --warmup 10 --workload cfg5_llm_n4 --loc-dist centre > gpurun_out/bench_r03c_cfg5_llm_n4_centre.json 2>/dev/null; show gpurun_out/bench_r03c_cfg5_llm_n4_centre.json
$B --steps 30 --warmup 10 --workload cfg5_llm_n4 --visible causal > gpurun_out/bench_r03c_cfg5_llm_n4_causal.json 2>/dev/null; show gpurun_out/bench_r03c_cfg5_llm_n4_causal.json
$B --steps 30 --warmup 10 --workload cfg5_llm_n4 --visible causal --loc-dist centre > gpurun_out/bench_r03c_cfg5_llm_n4_causal_centre.json 2>/dev/null; show gpurun_out/bench_r03c_cfg5_llm_n4_causal_centre.json
$B --steps 50 --warmup 50 --workload ref_speed_test --grad ones > gpurun_out/bench_r03c_ref_speed_test_f16.json 2>gpurun_out/bench_r03c_ref_speed_test_f16.err; show gpurun_out/bench_r03c_ref_speed_test_f16.json
$B --steps 50 --warmup 50 --workload ref_speed_test --grad ones --dtype f32 > gpurun_out/bench_r03c_ref_speed_test_f32.json 2>gpurun_out/bench_r03c_ref_speed_test_f32.err; show gpurun_out/bench_r03c_ref_speed_test_f32.json
echo "== --gpus 2 on a 1-GPU box"; python bench.py --gpus 2 --steps 2 --warmup 1; echo "rc=$?"
echo "== phase clocks"
MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/fprof.so timeout 120 python tools/fwd_prof.py cfg2_northstar 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03c_fwd_prof.log
echo "== rocprof"
bash tools/prof.sh r03c > gpurun_out/prof_r03c.log 2>&1; grep -A10 "== kernel stats" gpurun_out/prof_r03c.log | cut -c1-160; grep "msda_fwd" gpurun_out/prof_r03c.log | cut -c1-1200
