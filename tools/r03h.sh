#!/bin/bash
mkdir -p gpurun_out
echo "== tests"; timeout 900 python -m pytest tests/test_op_gpu.py tests/test_modules_gpu.py -q -k "lds_levels or rmsnorm or fused_norm" > gpurun_out/r03h_pytest_new.log 2>&1; tail -4 gpurun_out/r03h_pytest_new.log | cut -c1-250
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], r.get("kernels_mean_us"), "frac", r.get("fwdbwd_hbm_frac"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
echo "== bench"
B="timeout 150 python bench.py --no-cpu-baseline"
$B --steps 50 --warmup 10 > gpurun_out/bench_r03h.json 2>gpurun_out/bench_r03h.err; show gpurun_out/bench_r03h.json
MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/chain1.so $B --steps 50 --warmup 10 > gpurun_out/bench_r03h_chain1.json 2>/dev/null; show gpurun_out/bench_r03h_chain1.json
MMFS_TAPS_ALGO=vec $B --steps 50 --warmup 10 > gpurun_out/bench_r03h_tapsvec.json 2>/dev/null; show gpurun_out/bench_r03h_tapsvec.json
for q in 192 320 384; do MMFS_TAPS_MMA_QPW=$q MMFS_FWD_MMA_QPW=$q $B --steps 30 --warmup 10 > gpurun_out/bench_r03h_q$q.json 2>/dev/null; show gpurun_out/bench_r03h_q$q.json; done
