#!/bin/bash
mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], r.get("kernels_mean_us"), "frac", r.get("fwdbwd_hbm_frac"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
B="timeout 150 python bench.py --no-cpu-baseline"
for rep in 1 2; do
$B --steps 100 --warmup 10 > gpurun_out/bench_r03j_base_$rep.json 2>gpurun_out/bench_r03j.err; show gpurun_out/bench_r03j_base_$rep.json
for v in plain fwdnt tapsnt allnt; do
MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/$v.so $B --steps 100 --warmup 10 > gpurun_out/bench_r03j_${v}_$rep.json 2>/dev/null; show gpurun_out/bench_r03j_${v}_$rep.json
done
done
