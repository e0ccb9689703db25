#!/bin/bash
mkdir -p gpurun_out
echo "== rmsnorm tests"; timeout 600 python -m pytest tests/test_modules_gpu.py -q -k "rmsnorm or fused_norm" > gpurun_out/r03g_pytest_new.log 2>&1; tail -4 gpurun_out/r03g_pytest_new.log | cut -c1-250
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
$B --steps 50 --warmup 10 > gpurun_out/bench_r03g.json 2>gpurun_out/bench_r03g.err; show gpurun_out/bench_r03g.json
$B --steps 50 --warmup 10 > gpurun_out/bench_r03g_2.json 2>/dev/null; show gpurun_out/bench_r03g_2.json
MMFS_TAPS_ALGO=vec $B --steps 50 --warmup 10 > gpurun_out/bench_r03g_tapsvec.json 2>/dev/null; show gpurun_out/bench_r03g_tapsvec.json
MMFS_TAPS_ALGO=vec MMFS_FWD_ALGO=vec $B --steps 50 --warmup 10 > gpurun_out/bench_r03g_allvec.json 2>/dev/null; show gpurun_out/bench_r03g_allvec.json
MMFS_MMA_GRID=256 $B --steps 30 --warmup 10 > gpurun_out/bench_r03g_grid256.json 2>/dev/null; show gpurun_out/bench_r03g_grid256.json
