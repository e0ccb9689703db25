#!/bin/bash
mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_op_gpu.py tests/test_fuzz_gpu.py -q -x -k "fresh or unregistered or overlapping or device or gapped or lds_levels_taps or fuzz or status" > gpurun_out/r03z_pytest.log 2>&1; tail -4 gpurun_out/r03z_pytest.log | cut -c1-300
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03z_$name.json 2> gpurun_out/bench_r03z_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03z_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03z_{sys.argv[1]}.err").read()[-1500:])
PY
}
run fresh_nofold MMFS_PREPARE_IN_TAPS=0 python bench.py --fresh-levels --steps 20 --warmup 5 --no-cpu-baseline
run fresh_fold python bench.py --fresh-levels --steps 20 --warmup 5 --no-cpu-baseline
run fresh_nofold2 MMFS_PREPARE_IN_TAPS=0 python bench.py --fresh-levels --steps 20 --warmup 5 --no-cpu-baseline
run fresh_fold2 python bench.py --fresh-levels --steps 20 --warmup 5 --no-cpu-baseline
