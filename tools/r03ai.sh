#!/bin/bash
mkdir -p gpurun_out
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03ai_$name.json 2> gpurun_out/bench_r03ai_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03ai_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03ai_{sys.argv[1]}.err").read()[-800:])
PY
}
E=$PWD/mm-interleaved_amd/csrc/build/exp
echo "== tests (new)"; timeout 900 python -m pytest tests/test_op_gpu.py -q -x -k "lds_levels or hosted or many_point or default or non_finite" 2>&1 | tail -2
for i in 1 2 3; do
run old$i MMFS_MSDA_LIB=$E/fillold.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run new$i MMFS_MSDA_LIB=$E/fillnew.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline
done
run speed_old MMFS_MSDA_LIB=$E/fillold.so python bench.py --workload ref_speed_test --grad ones --steps 50 --warmup 50 --no-cpu-baseline
run speed_new MMFS_MSDA_LIB=$E/fillnew.so python bench.py --workload ref_speed_test --grad ones --steps 50 --warmup 50 --no-cpu-baseline
