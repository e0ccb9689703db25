#!/bin/bash
# One GPU-box visit: the backward-heavy op tests, bench lines [, rocprof].  usage: tools/run_gpu.sh <tag> [prof]
tag=$1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_op_gpu.py -q -x -k "16bit or seeded or generations or one_spot or skewed or level_rows or full_size or zero_attention or lazy or hybrid" > gpurun_out/pytest_$tag.log 2>&1; tail -5 gpurun_out/pytest_$tag.log
timeout 100 python bench.py --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/bench_$tag.json 2>gpurun_out/bench_$tag.err; python - <<PY
import json
r=json.load(open("gpurun_out/bench_$tag.json")); print("cfg2_northstar", r["ms_per_step"], r["kernels_mean_us"], "frac", r["fwdbwd_hbm_frac"])
PY
for w in cfg2_sd_real cfg5_llm_n4; do timeout 100 python bench.py --no-cpu-baseline --steps 50 --warmup 10 --workload $w > gpurun_out/bench_${tag}_$w.json 2>/dev/null; python - <<PY
import json
r=json.load(open("gpurun_out/bench_${tag}_$w.json")); print("$w", r["ms_per_step"], r["kernels_mean_us"])
PY
done
if [ "$2" = prof ]; then bash tools/prof.sh $tag > gpurun_out/prof_$tag.log 2>&1; grep -A12 "== kernel stats" gpurun_out/prof_$tag.log | cut -c1-150; grep "tile_reduce:\|taps_finalize:" gpurun_out/prof_$tag.log; fi
