#!/bin/bash
mkdir -p gpurun_out
echo "== tests with the by-product re-pack"; MMFS_TAPS_REPACK=1 timeout 900 python -m pytest tests/test_op_gpu.py -q -x -k "hosted or lds_levels_taps or full_size or many_point" > gpurun_out/r03u_pytest.log 2>&1; tail -4 gpurun_out/r03u_pytest.log | cut -c1-300
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03u_$name.json 2> gpurun_out/bench_r03u_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03u_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r["kernels_mean_us"])
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03u_{sys.argv[1]}.err").read()[-1500:])
PY
}
for i in 1 2; do
run ns_base$i python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_repack$i MMFS_TAPS_REPACK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
done
run ns_repack_100 MMFS_TAPS_REPACK=1 python bench.py --steps 100 --warmup 20 --no-cpu-baseline
run ns_base_100 python bench.py --steps 100 --warmup 20 --no-cpu-baseline
