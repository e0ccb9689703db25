#!/bin/bash
# LDS-resident kernels: persistent slab-major workgroups (default now) vs one workgroup per run vs the strided persistent deal
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
run() { tag=$1; shift; env "$@" $B > gpurun_out/bench_r03bf_$tag.json 2>/dev/null; show gpurun_out/bench_r03bf_$tag.json; }
for rep in 1 2 3; do
run slab_$rep X=1
run perrun_$rep MMFS_MMA_PERSIST=0
run strided256_$rep MMFS_MMA_GRID=256
done
B="timeout 200 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --loc-dist centre"
run centre_slab X=1
run centre_perrun MMFS_MMA_PERSIST=0
B="timeout 200 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --fresh-levels"
run fresh_slab X=1
run fresh_perrun MMFS_MMA_PERSIST=0
B="timeout 250 python bench.py"
$B --steps 20 --warmup 5 > gpurun_out/bench_r03bf_driver.json 2>/dev/null; show gpurun_out/bench_r03bf_driver.json
