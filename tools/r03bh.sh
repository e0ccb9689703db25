#!/bin/bash
# second looks under persistent workgroups: queries per workgroup; heads of 64 channels through msda_fwd_mma
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], {k: round(v, 1) for k, v in (r.get("kernels_mean_us") or {}).items()})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
B="timeout 200 python bench.py --no-cpu-baseline --steps 60 --warmup 15"
run() { tag=$1; shift; env "$@" $B > gpurun_out/bench_r03bh_$tag.json 2>/dev/null; show gpurun_out/bench_r03bh_$tag.json; }
run ns_q256 X=1
run ns_q128 MMFS_FWD_MMA_QPW=128 MMFS_TAPS_MMA_QPW=128
run ns_q512 MMFS_FWD_MMA_QPW=512 MMFS_TAPS_MMA_QPW=512
run ns_q256b X=1
for w in cfg2_sd_real cfg5_llm_n4; do
B="timeout 200 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --workload $w"
run ${w}_vec X=1
run ${w}_mma MMFS_FWD_ALGO=mma
run ${w}_mma512 MMFS_FWD_ALGO=mma MMFS_FWD_MMA_QPW=512
run ${w}_mma_perrun MMFS_FWD_ALGO=mma MMFS_MMA_PERSIST=0
done
