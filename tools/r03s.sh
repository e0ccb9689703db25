#!/bin/bash
mkdir -p gpurun_out
echo "== all gpu tests"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03s_pytest_all.log 2>&1; tail -5 gpurun_out/r03s_pytest_all.log | cut -c1-300
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03s_$name.json 2> gpurun_out/bench_r03s_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03s_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r["kernels_mean_us"])
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03s_{sys.argv[1]}.err").read()[-1500:])
PY
}
run ns_nofold MMFS_PREPARE_IN_TAPS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_fold python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_nofold2 MMFS_PREPARE_IN_TAPS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_fold2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_fold_100 python bench.py --steps 100 --warmup 20 --no-cpu-baseline
run sd python bench.py --workload cfg2_sd_real --steps 20 --warmup 5 --no-cpu-baseline
run llm python bench.py --workload cfg5_llm_n4 --steps 20 --warmup 5 --no-cpu-baseline
run speed_f16 python bench.py --workload ref_speed_test --grad ones --steps 50 --warmup 50 --no-cpu-baseline
