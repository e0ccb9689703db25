#!/bin/bash
mkdir -p gpurun_out
echo "== value-path tests"; timeout 900 python -m pytest tests/test_op_gpu.py -q -x -k "sort_routes or value_algo or hot_spot or overflow or one_spot or full_size or level_rows or staged" > gpurun_out/r03q_pytest.log 2>&1; tail -4 gpurun_out/r03q_pytest.log | cut -c1-300
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03q_$name.json 2> gpurun_out/bench_r03q_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03q_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r["kernels_mean_us"])
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03q_{sys.argv[1]}.err").read()[-1500:])
PY
}
for w in cfg2_northstar cfg2_sd_real cfg5_llm_n4; do
  run ${w} python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
done
run cfg2_northstar_2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run llm_causal_centre python bench.py --workload cfg5_llm_n4 --visible causal --loc-dist centre --steps 20 --warmup 5 --no-cpu-baseline
echo "== traffic"
bash tools/pmc_traffic.sh r03q cfg2_northstar 2>&1 | grep -E "reduce|sort|fwd|taps"
