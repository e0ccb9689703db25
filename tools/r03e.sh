#!/bin/bash
mkdir -p gpurun_out
echo "== lds tests (fwd + taps)"; timeout 900 python -m pytest tests/test_op_gpu.py -q -k "lds_levels or many_points" > gpurun_out/r03e_pytest_lds.log 2>&1; tail -6 gpurun_out/r03e_pytest_lds.log | cut -c1-250
echo "== all gpu tests"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03e_pytest_all.log 2>&1; tail -6 gpurun_out/r03e_pytest_all.log | cut -c1-250
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
$B --steps 50 --warmup 10 > gpurun_out/bench_r03e.json 2>gpurun_out/bench_r03e.err; show gpurun_out/bench_r03e.json
$B --steps 20 --warmup 5 > gpurun_out/bench_r03e_driver.json 2>/dev/null; show gpurun_out/bench_r03e_driver.json
MMFS_TAPS_ALGO=vec $B --steps 30 --warmup 10 > gpurun_out/bench_r03e_tapsvec.json 2>/dev/null; show gpurun_out/bench_r03e_tapsvec.json
for g in 128 512 1024; do MMFS_MMA_GRID=$g $B --steps 30 --warmup 10 > gpurun_out/bench_r03e_grid$g.json 2>/dev/null; show gpurun_out/bench_r03e_grid$g.json; done
for q in 128 512; do MMFS_TAPS_MMA_QPW=$q MMFS_FWD_MMA_QPW=$q $B --steps 30 --warmup 10 > gpurun_out/bench_r03e_q$q.json 2>/dev/null; show gpurun_out/bench_r03e_q$q.json; done
$B --steps 30 --warmup 10 --loc-dist centre > gpurun_out/bench_r03e_centre.json 2>/dev/null; show gpurun_out/bench_r03e_centre.json
$B --steps 30 --warmup 10 --fresh-levels > gpurun_out/bench_r03e_fresh.json 2>/dev/null; show gpurun_out/bench_r03e_fresh.json
echo "== phase clocks"
MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/fprof.so timeout 120 python tools/fwd_prof.py cfg2_northstar 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03e_fwd_prof.log
