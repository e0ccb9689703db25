#!/bin/bash
mkdir -p gpurun_out
echo "== all gpu tests"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03ad_pytest_all.log 2>&1; tail -5 gpurun_out/r03ad_pytest_all.log | cut -c1-300
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03ad_$name.json 2> gpurun_out/bench_r03ad_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03ad_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03ad_{sys.argv[1]}.err").read()[-1500:])
PY
}
run speed_f16 python bench.py --workload ref_speed_test --grad ones --steps 50 --warmup 50 --no-cpu-baseline
run speed_f32 python bench.py --workload ref_speed_test --grad ones --dtype f32 --steps 50 --warmup 50 --no-cpu-baseline
run ns python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run nq1024 python bench.py --nq 1024 --steps 50 --warmup 20 --no-cpu-baseline
