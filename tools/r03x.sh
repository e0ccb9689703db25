#!/bin/bash
mkdir -p gpurun_out
echo "== forward tests"; timeout 900 python -m pytest tests/test_op_gpu.py tests/test_modules_gpu.py -q -x -k "golden or geometr or forward or fwd or lds or sample or plan" > gpurun_out/r03x_pytest.log 2>&1; tail -4 gpurun_out/r03x_pytest.log | cut -c1-300
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03x_$name.json 2> gpurun_out/bench_r03x_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03x_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03x_{sys.argv[1]}.err").read()[-1500:])
PY
}
N=$PWD/mm-interleaved_amd/csrc/build/exp/nopk.so
for w in cfg2_northstar cfg2_sd_real cfg5_llm_n4; do
  run ${w}_nopk MMFS_MSDA_LIB=$N python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
  run ${w}_pk python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline
done
run cfg2_northstar_nopk2 MMFS_MSDA_LIB=$N python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run cfg2_northstar_pk2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run cfg1_nopk MMFS_MSDA_LIB=$N python bench.py --workload cfg1 --steps 50 --warmup 20 --no-cpu-baseline
run cfg1_pk python bench.py --workload cfg1 --steps 50 --warmup 20 --no-cpu-baseline
