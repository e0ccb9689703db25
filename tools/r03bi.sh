#!/bin/bash
# tile reduce: weight tile written whole ([record][hi | lo], two 8-byte writes per lane) and read transposed
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
B="timeout 200 python bench.py --no-cpu-baseline --steps 60 --warmup 15"
run() { tag=$1; shift; env "$@" $B > gpurun_out/bench_r03bi_$tag.json 2>/dev/null; show gpurun_out/bench_r03bi_$tag.json; }
run ns_1 X=1
run ns_2 X=1
for w in cfg2_sd_real cfg5_llm_n4 enc_injector; do
B="timeout 200 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --workload $w"
run ${w} X=1
done
B="timeout 200 python bench.py --no-cpu-baseline --steps 50 --warmup 50 --workload ref_speed_test --grad ones"
run speed X=1
