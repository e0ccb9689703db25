#!/bin/bash
# first GPU visit of the workgroup-local grad_value kernel: parity, then what it buys
mkdir -p gpurun_out
echo "== gv tests"; timeout 600 python -m pytest tests/test_op_gpu.py -q -x -k "lds_blocks" > gpurun_out/r03m_pytest_gv.log 2>&1; tail -15 gpurun_out/r03m_pytest_gv.log | cut -c1-300
b() { # name, env..., -- args
  local name=$1; shift
  ( export "$@" 2>/dev/null; true )
  :
}
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03m_$name.json 2> gpurun_out/bench_r03m_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03m_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r["kernels_mean_us"])
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03m_{sys.argv[1]}.err").read()[-1500:])
PY
}
echo "== north star"
run ns_off MMFS_GV_ALGO=off python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_t512 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_t256 MMFS_GV_TARGET_WGS=256 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_t1024 MMFS_GV_TARGET_WGS=1024 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
echo "== real geometries"
run sd_off MMFS_GV_ALGO=off python bench.py --workload cfg2_sd_real --steps 20 --warmup 5 --no-cpu-baseline
run sd_on python bench.py --workload cfg2_sd_real --steps 20 --warmup 5 --no-cpu-baseline
run llm_off MMFS_GV_ALGO=off python bench.py --workload cfg5_llm_n4 --steps 20 --warmup 5 --no-cpu-baseline
run llm_on python bench.py --workload cfg5_llm_n4 --steps 20 --warmup 5 --no-cpu-baseline
run llm_t256 MMFS_GV_TARGET_WGS=256 python bench.py --workload cfg5_llm_n4 --steps 20 --warmup 5 --no-cpu-baseline
