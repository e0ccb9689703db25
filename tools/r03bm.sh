#!/bin/bash
# the hosted grad_value plan in a workgroup of its own (persistent msda_taps_mma; msda_taps_coarse) vs as a worker's epilogue
mkdir -p gpurun_out
echo "== hosted-plan tests"; timeout 900 python -m pytest tests/test_op_gpu.py -q -x -k "hosted or backward or persistent or full_size" 2>&1 | tail -2 | cut -c1-200
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], {k: round(v, 1) for k, v in (r.get("kernels_mean_us") or {}).items()})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
BASE=$PWD/mm-interleaved_amd/csrc/build/exp/base.so
run() { tag=$1; shift; env "$@" $B > gpurun_out/bench_r03bm_$tag.json 2>/dev/null; show gpurun_out/bench_r03bm_$tag.json; }
B="timeout 200 python bench.py --no-cpu-baseline --steps 60 --warmup 15"
for rep in 1 2 3; do
run ns_new_$rep X=1
run ns_base_$rep MMFS_MSDA_LIB=$BASE
done
for w in cfg2_sd_real cfg5_llm_n4; do
B="timeout 200 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --workload $w"
for rep in 1 2; do
run ${w}_new_$rep X=1
run ${w}_base_$rep MMFS_MSDA_LIB=$BASE
done; done
