#!/bin/bash
mkdir -p gpurun_out
echo "== sort route + gv tests"; timeout 900 python -m pytest tests/test_op_gpu.py -q -x -k "sort_routes or lds_blocks or value_algo or hot_spot or overflow" > gpurun_out/r03p_pytest.log 2>&1; tail -5 gpurun_out/r03p_pytest.log | cut -c1-300
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03p_$name.json 2> gpurun_out/bench_r03p_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03p_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r["kernels_mean_us"])
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03p_{sys.argv[1]}.err").read()[-1500:])
PY
}
NT=$PWD/mm-interleaved_amd/csrc/build/exp/sortnt.so
for w in cfg2_sd_real cfg2_northstar cfg5_llm_n4; do
  run ${w}_base python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
  run ${w}_sortnt MMFS_MSDA_LIB=$NT python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
done
run cfg2_sd_real_norounds MMFS_SORT_ROUNDS=0 python bench.py --workload cfg2_sd_real --steps 20 --warmup 5 --no-cpu-baseline
run cfg2_sd_real_base2 python bench.py --workload cfg2_sd_real --steps 20 --warmup 5 --no-cpu-baseline
