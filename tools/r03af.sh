#!/bin/bash
mkdir -p gpurun_out
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03af_$name.json 2> gpurun_out/bench_r03af_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03af_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03af_{sys.argv[1]}.err").read()[-1500:])
PY
}
E=$PWD/mm-interleaved_amd/csrc/build/exp
echo "== tests (new)"; timeout 600 python -m pytest tests/test_op_gpu.py -q -x -k "lds_levels or hosted" 2>&1 | tail -2
for i in 1 2 3; do
run old$i MMFS_MSDA_LIB=$E/tabold.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run new$i MMFS_MSDA_LIB=$E/tabnew.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline
done
