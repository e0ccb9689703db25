#!/bin/bash
mkdir -p gpurun_out
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03ah_$name.json 2> gpurun_out/bench_r03ah_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03ah_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03ah_{sys.argv[1]}.err").read()[-800:])
PY
}
E=$PWD/mm-interleaved_amd/csrc/build/exp
run w16 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run w8 MMFS_MSDA_LIB=$E/w8.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run w8_q128 MMFS_MSDA_LIB=$E/w8.so MMFS_FWD_MMA_QPW=128 MMFS_TAPS_MMA_QPW=128 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run w12 MMFS_MSDA_LIB=$E/w12.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run w12_q192 MMFS_MSDA_LIB=$E/w12.so MMFS_FWD_MMA_QPW=192 MMFS_TAPS_MMA_QPW=192 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run w16b python bench.py --steps 20 --warmup 5 --no-cpu-baseline
