#!/bin/bash
# first batch of the image fill requested before the barrier between runs (no spills this time) vs the committed build
mkdir -p gpurun_out
echo "== op tests"; timeout 1200 python -m pytest tests/test_op_gpu.py -q -x 2>&1 | tail -2 | cut -c1-200
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], {k: round(v, 1) for k, v in (r.get("kernels_mean_us") or {}).items()})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
BASE=$PWD/mm-interleaved_amd/csrc/build/exp/base.so
run() { tag=$1; shift; env "$@" $B > gpurun_out/bench_r03bs_$tag.json 2>/dev/null; show gpurun_out/bench_r03bs_$tag.json; }
B="timeout 200 python bench.py --no-cpu-baseline --steps 60 --warmup 15"
for rep in 1 2 3; do
run early_$rep X=1
run base_$rep MMFS_MSDA_LIB=$BASE
done
B="timeout 200 python bench.py --no-cpu-baseline --steps 50 --warmup 50 --workload ref_speed_test --grad ones"
run speed_early X=1
run speed_base MMFS_MSDA_LIB=$BASE
