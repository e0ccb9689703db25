#!/bin/bash
mkdir -p gpurun_out
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03r_$name.json 2> gpurun_out/bench_r03r_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03r_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r["kernels_mean_us"])
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03r_{sys.argv[1]}.err").read()[-1500:])
PY
}
E=$PWD/mm-interleaved_amd/csrc/build/exp
for w in cfg2_northstar cfg5_llm_n4; do
  run ${w}_base python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
  run ${w}_c2048 MMFS_MSDA_LIB=$E/chunk2048.so python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
  run ${w}_c512 MMFS_MSDA_LIB=$E/chunk512.so python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
done
run llm_cc_base python bench.py --workload cfg5_llm_n4 --visible causal --loc-dist centre --steps 20 --warmup 5 --no-cpu-baseline
run llm_cc_c2048 MMFS_MSDA_LIB=$E/chunk2048.so python bench.py --workload cfg5_llm_n4 --visible causal --loc-dist centre --steps 20 --warmup 5 --no-cpu-baseline
MMFS_MSDA_LIB=$E/chunk2048.so bash tools/pmc_traffic.sh r03r_c2048 cfg2_northstar MMFS_MSDA_LIB=$E/chunk2048.so 2>&1 | grep -E "reduce"
