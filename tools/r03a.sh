#!/bin/bash
# round 3, first GPU visit: hardware probes of the new forward's assumptions, its parity tests, the whole GPU
# suite, then bench lines with the forward's two formulations.
mkdir -p gpurun_out
echo "== probe"; timeout 60 tools/ubench/mfma16_probe > gpurun_out/r03a_probe.log 2>&1; grep -E "^Q" gpurun_out/r03a_probe.log
echo "== lds tests"; timeout 600 python -m pytest tests/test_op_gpu.py -q -k "lds_levels" > gpurun_out/r03a_pytest_lds.log 2>&1; tail -25 gpurun_out/r03a_pytest_lds.log | cut -c1-220
echo "== all gpu tests"; timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r03a_pytest_all.log 2>&1; tail -6 gpurun_out/r03a_pytest_all.log | cut -c1-220
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], r.get("kernels_mean_us"), "frac", r.get("fwdbwd_hbm_frac"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
echo "== bench"
timeout 120 python bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/bench_r03a_mma.json 2>gpurun_out/bench_r03a_mma.err; show gpurun_out/bench_r03a_mma.json
MMFS_FWD_ALGO=vec timeout 120 python bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/bench_r03a_vec.json 2>gpurun_out/bench_r03a_vec.err; show gpurun_out/bench_r03a_vec.json
for q in 64 128 512 1024 4096; do
  MMFS_FWD_MMA_QPW=$q timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/bench_r03a_mma_q$q.json 2>/dev/null; show gpurun_out/bench_r03a_mma_q$q.json
done
for w in cfg2_sd_real cfg5_llm_n4; do
  timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --workload $w > gpurun_out/bench_r03a_mma_$w.json 2>/dev/null; show gpurun_out/bench_r03a_mma_$w.json
  MMFS_FWD_ALGO=vec timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --workload $w > gpurun_out/bench_r03a_vec_$w.json 2>/dev/null; show gpurun_out/bench_r03a_vec_$w.json
done
echo "== forward alone"
timeout 120 python tools/fwd_repeat.py > gpurun_out/r03a_fwd_repeat.log 2>&1; tail -8 gpurun_out/r03a_fwd_repeat.log
