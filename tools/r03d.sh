#!/bin/bash
mkdir -p gpurun_out
echo "== taps lds tests"; timeout 900 python -m pytest tests/test_op_gpu.py -q -k "lds_levels_taps" > gpurun_out/r03d_pytest_taps.log 2>&1; tail -15 gpurun_out/r03d_pytest_taps.log | cut -c1-250
echo "== all gpu tests"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03d_pytest_all.log 2>&1; tail -8 gpurun_out/r03d_pytest_all.log | cut -c1-250
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
$B --steps 50 --warmup 10 > gpurun_out/bench_r03d.json 2>gpurun_out/bench_r03d.err; show gpurun_out/bench_r03d.json
MMFS_TAPS_ALGO=vec $B --steps 50 --warmup 10 > gpurun_out/bench_r03d_tapsvec.json 2>/dev/null; show gpurun_out/bench_r03d_tapsvec.json
for q in 128 512; do MMFS_TAPS_MMA_QPW=$q $B --steps 30 --warmup 10 > gpurun_out/bench_r03d_tq$q.json 2>/dev/null; show gpurun_out/bench_r03d_tq$q.json; done
$B --steps 30 --warmup 10 --fresh-levels > gpurun_out/bench_r03d_fresh.json 2>/dev/null; show gpurun_out/bench_r03d_fresh.json
$B --steps 30 --warmup 10 --loc-dist centre > gpurun_out/bench_r03d_centre.json 2>/dev/null; show gpurun_out/bench_r03d_centre.json
$B --steps 50 --warmup 50 --workload ref_speed_test --grad ones > gpurun_out/bench_r03d_ref_speed_test_f16.json 2>/dev/null; show gpurun_out/bench_r03d_ref_speed_test_f16.json
echo "== module bench cfg3"
timeout 600 python tools/module_bench.py cfg3 > gpurun_out/r03d_module_bench_cfg3.jsonl 2>gpurun_out/r03d_module_bench_cfg3.err; python - <<'PY'
import json
for l in open("gpurun_out/r03d_module_bench_cfg3.jsonl"):
    r = json.loads(l); print(r["what"][40:], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"])
PY
tail -3 gpurun_out/r03d_module_bench_cfg3.err
