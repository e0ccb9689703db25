#!/bin/bash
mkdir -p gpurun_out
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03ac_$name.json 2> gpurun_out/bench_r03ac_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03ac_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r.get("kernels_mean_us"))
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03ac_{sys.argv[1]}.err").read()[-1500:])
PY
}
S="python bench.py --workload ref_speed_test --grad ones --steps 50 --warmup 50 --no-cpu-baseline"
run speed_f16 $S
run speed_f16_fwdmma MMFS_FWD_ALGO=mma $S
run speed_f16_bothmma MMFS_FWD_ALGO=mma MMFS_TAPS_ALGO=mma $S
run speed_f16_bothmma_q128 MMFS_FWD_ALGO=mma MMFS_TAPS_ALGO=mma MMFS_FWD_MMA_QPW=128 MMFS_TAPS_MMA_QPW=128 $S
run speed_f16_bothmma_q64 MMFS_FWD_ALGO=mma MMFS_TAPS_ALGO=mma MMFS_FWD_MMA_QPW=64 MMFS_TAPS_MMA_QPW=64 $S
for nq in 64 128 192; do
run ns_nq${nq}_vec python bench.py --nq $nq --steps 50 --warmup 20 --no-cpu-baseline
run ns_nq${nq}_mma MMFS_FWD_ALGO=mma MMFS_TAPS_ALGO=mma python bench.py --nq $nq --steps 50 --warmup 20 --no-cpu-baseline
done
