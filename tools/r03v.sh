#!/bin/bash
mkdir -p gpurun_out
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03v_$name.json 2> gpurun_out/bench_r03v_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03v_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r["kernels_mean_us"])
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03v_{sys.argv[1]}.err").read()[-1500:])
PY
}
for i in 1 2; do
run ns_base$i python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_overlap$i MMFS_BWD_OVERLAP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
done
run sd_base python bench.py --workload cfg2_sd_real --steps 20 --warmup 5 --no-cpu-baseline
run sd_overlap MMFS_BWD_OVERLAP=1 python bench.py --workload cfg2_sd_real --steps 20 --warmup 5 --no-cpu-baseline
run llm_base python bench.py --workload cfg5_llm_n4 --steps 20 --warmup 5 --no-cpu-baseline
run llm_overlap MMFS_BWD_OVERLAP=1 python bench.py --workload cfg5_llm_n4 --steps 20 --warmup 5 --no-cpu-baseline
