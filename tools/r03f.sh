#!/bin/bash
mkdir -p gpurun_out
echo "== new tests"; timeout 900 python -m pytest tests/test_op_gpu.py tests/test_modules_gpu.py -q -k "lds_levels or rmsnorm or fused_norm" > gpurun_out/r03f_pytest_new.log 2>&1; tail -12 gpurun_out/r03f_pytest_new.log | cut -c1-250
echo "== all gpu tests"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03f_pytest_all.log 2>&1; tail -6 gpurun_out/r03f_pytest_all.log | cut -c1-250
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
$B --steps 50 --warmup 10 > gpurun_out/bench_r03f.json 2>gpurun_out/bench_r03f.err; show gpurun_out/bench_r03f.json
MMFS_TAPS_ALGO=vec $B --steps 50 --warmup 10 > gpurun_out/bench_r03f_tapsvec.json 2>/dev/null; show gpurun_out/bench_r03f_tapsvec.json
MMFS_MMA_GRID=1024 $B --steps 30 --warmup 10 > gpurun_out/bench_r03f_grid1024.json 2>/dev/null; show gpurun_out/bench_r03f_grid1024.json
for q in 512 1024; do MMFS_TAPS_MMA_QPW=$q MMFS_FWD_MMA_QPW=$q $B --steps 30 --warmup 10 > gpurun_out/bench_r03f_q$q.json 2>/dev/null; show gpurun_out/bench_r03f_q$q.json; done
echo "== phase clocks"
MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/fprof.so timeout 120 python tools/fwd_prof.py cfg2_northstar 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03f_fwd_prof.log
echo "== module bench cfg3"
timeout 600 python tools/module_bench.py cfg3 > gpurun_out/r03f_module_bench_cfg3.jsonl 2>gpurun_out/r03f_module_bench_cfg3.err; python - <<'PY'
import json
for l in open("gpurun_out/r03f_module_bench_cfg3.jsonl"):
    r = json.loads(l); print(r["what"][40:], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"])
PY
tail -3 gpurun_out/r03f_module_bench_cfg3.err
