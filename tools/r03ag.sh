#!/bin/bash
mkdir -p gpurun_out
echo "== tests"; timeout 600 python -m pytest tests/test_op_gpu.py -q -x -k "lds_levels_forward or lds_forward or non_finite or default" 2>&1 | tail -2
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03ag_$name.json 2> gpurun_out/bench_r03ag_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03ag_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03ag_{sys.argv[1]}.err").read()[-1500:])
PY
}
for w in cfg2_sd_real cfg5_llm_n4; do
run ${w}_vec python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
run ${w}_mma MMFS_FWD_ALGO=mma python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
run ${w}_mma_q512 MMFS_FWD_ALGO=mma MMFS_FWD_MMA_QPW=512 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
done
run ns python bench.py --steps 20 --warmup 5 --no-cpu-baseline
