#!/bin/bash
mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_op_gpu.py -q -x -k "many_point or sort_routes or speed" > gpurun_out/r03ab_pytest.log 2>&1; tail -4 gpurun_out/r03ab_pytest.log | cut -c1-300
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03ab_$name.json 2> gpurun_out/bench_r03ab_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03ab_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03ab_{sys.argv[1]}.err").read()[-1500:])
PY
}
run speed_f16_scalar MMFS_SORT_MANY_POINTS=0 python bench.py --workload ref_speed_test --grad ones --steps 50 --warmup 50 --no-cpu-baseline
run speed_f16 python bench.py --workload ref_speed_test --grad ones --steps 50 --warmup 50 --no-cpu-baseline
run speed_f32_scalar MMFS_SORT_MANY_POINTS=0 python bench.py --workload ref_speed_test --grad ones --dtype f32 --steps 50 --warmup 50 --no-cpu-baseline
run speed_f32 python bench.py --workload ref_speed_test --grad ones --dtype f32 --steps 50 --warmup 50 --no-cpu-baseline
run ns python bench.py --steps 20 --warmup 5 --no-cpu-baseline
