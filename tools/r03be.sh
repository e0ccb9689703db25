#!/bin/bash
# queries per workgroup / persistent grid of the LDS-resident kernels, re-tuned on the closing build
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
run() { tag=$1; shift; env "$@" $B > gpurun_out/bench_r03be_$tag.json 2>/dev/null; show gpurun_out/bench_r03be_$tag.json; }
run base_1 X=1
run q512 MMFS_FWD_MMA_QPW=512 MMFS_TAPS_MMA_QPW=512
run q384 MMFS_FWD_MMA_QPW=384 MMFS_TAPS_MMA_QPW=384
run q192 MMFS_FWD_MMA_QPW=192 MMFS_TAPS_MMA_QPW=192
run q128 MMFS_FWD_MMA_QPW=128 MMFS_TAPS_MMA_QPW=128
run base_2 X=1
run grid256 MMFS_MMA_GRID=256
run grid512 MMFS_MMA_GRID=512
run q1024 MMFS_FWD_MMA_QPW=1024 MMFS_TAPS_MMA_QPW=1024
run base_3 X=1
