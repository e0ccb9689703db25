#!/bin/bash
mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_op_gpu.py tests/test_zz_graph_gpu.py tests/test_fuzz_gpu.py -q -x -k "hosted or hybrid or HYBRID or dense or graph or fuzz or many_point" > gpurun_out/r03y_pytest.log 2>&1; tail -4 gpurun_out/r03y_pytest.log | cut -c1-300
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03y_$name.json 2> gpurun_out/bench_r03y_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03y_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03y_{sys.argv[1]}.err").read()[-1500:])
PY
}
for w in cfg2_sd_real cfg5_llm_n4 enc_injector; do
  run ${w}_nofold MMFS_PREPARE_IN_TAPS=0 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
  run ${w}_fold python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
done
run cfg5_llm_cc_nofold MMFS_PREPARE_IN_TAPS=0 python bench.py --workload cfg5_llm_n4 --visible causal --loc-dist centre --steps 20 --warmup 5 --no-cpu-baseline
run cfg5_llm_cc_fold python bench.py --workload cfg5_llm_n4 --visible causal --loc-dist centre --steps 20 --warmup 5 --no-cpu-baseline
