#!/bin/bash
# tile reduce with four row slots at D <= 64 (MMFS_TILE_STAGES): parity + A/B on one box
mkdir -p gpurun_out
echo "== backward tests"; timeout 1200 python -m pytest tests/test_op_gpu.py -q -x 2>&1 | tail -2 | cut -c1-200
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], {k: round(v, 1) for k, v in (r.get("kernels_mean_us") or {}).items()})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
B="timeout 200 python bench.py --no-cpu-baseline"
for rep in 1 2; do for w in cfg2_sd_real cfg5_llm_n4; do for stg in 2 4; do
  MMFS_TILE_STAGES=$stg $B --steps 30 --warmup 10 --workload $w > gpurun_out/bench_r03ba_${w}_st${stg}_$rep.json 2>/dev/null; show gpurun_out/bench_r03ba_${w}_st${stg}_$rep.json
done; done; done
for stg in 2 4; do
MMFS_TILE_STAGES=$stg $B --steps 30 --warmup 10 --workload cfg5_llm_n4 --visible causal --loc-dist centre > gpurun_out/bench_r03ba_llm_cc_st$stg.json 2>/dev/null; show gpurun_out/bench_r03ba_llm_cc_st$stg.json
MMFS_TILE_STAGES=$stg $B --steps 30 --warmup 10 --workload enc_injector > gpurun_out/bench_r03ba_enc_st$stg.json 2>/dev/null; show gpurun_out/bench_r03ba_enc_st$stg.json
done
