#!/bin/bash
# sort workgroups of neighbouring heads on one XCD (MMFS_SORT_HGROUP): alternations on one box
mkdir -p gpurun_out
echo "== sort tests"; timeout 900 python -m pytest tests/test_op_gpu.py -q -x -k "sort or kept or backward or hosted or many_point" 2>&1 | tail -2 | cut -c1-200
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], {k: round(v, 1) for k, v in (r.get("kernels_mean_us") or {}).items()})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
B="timeout 200 python bench.py --no-cpu-baseline"
for rep in 1 2 3; do
for hg in 0 4 2; do
  MMFS_SORT_HGROUP=$hg $B --steps 100 --warmup 20 > gpurun_out/bench_r03ay_ns_hg${hg}_$rep.json 2>/dev/null; show gpurun_out/bench_r03ay_ns_hg${hg}_$rep.json
done; done
for w in cfg2_sd_real cfg5_llm_n4; do for hg in 0 2 4; do
  MMFS_SORT_HGROUP=$hg $B --steps 30 --warmup 10 --workload $w > gpurun_out/bench_r03ay_${w}_hg$hg.json 2>/dev/null; show gpurun_out/bench_r03ay_${w}_hg$hg.json
done; done
