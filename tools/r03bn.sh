#!/bin/bash
# dense-levels taps kernel: tiles per workgroup (its grad_out tiles are re-read per level: a cyclic working set at the size of L2)
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], {k: round(v, 1) for k, v in (r.get("kernels_mean_us") or {}).items()})
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
run() { tag=$1; shift; env "$@" $B > gpurun_out/bench_r03bn_$tag.json 2>/dev/null; show gpurun_out/bench_r03bn_$tag.json; }
for w in cfg5_llm_n4 cfg2_sd_real enc_injector; do
B="timeout 200 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --workload $w"
for t in 256 512 1024 2048 256; do
run ${w}_t$t MMFS_DENSE_TARGET=$t
done; done
