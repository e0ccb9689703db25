#!/bin/bash
mkdir -p gpurun_out
echo "== lds tests"; timeout 600 python -m pytest tests/test_op_gpu.py -q -k "lds_levels" > gpurun_out/r03b_pytest_lds.log 2>&1; tail -4 gpurun_out/r03b_pytest_lds.log | cut -c1-220
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", r["ms_per_step"], r.get("kernels_mean_us"), "frac", r.get("fwdbwd_hbm_frac"))
except Exception as e:
    print(sys.argv[1], "no line:", e)
PY
}
echo "== bench"
timeout 120 python bench.py --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/bench_r03b_mma.json 2>gpurun_out/bench_r03b_mma.err; show gpurun_out/bench_r03b_mma.json
for q in 128 512; do
  MMFS_FWD_MMA_QPW=$q timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > gpurun_out/bench_r03b_mma_q$q.json 2>/dev/null; show gpurun_out/bench_r03b_mma_q$q.json
done
for w in cfg2_sd_real cfg5_llm_n4; do
  timeout 120 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --workload $w > gpurun_out/bench_r03b_mma_$w.json 2>/dev/null; show gpurun_out/bench_r03b_mma_$w.json
done
echo "== forward alone"
timeout 120 python tools/fwd_repeat.py > gpurun_out/r03b_fwd_repeat.log 2>&1; tail -7 gpurun_out/r03b_fwd_repeat.log
echo "== phase clocks"
for w in cfg2_northstar cfg2_sd_real cfg5_llm_n4; do
MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/fprof.so timeout 120 python tools/fwd_prof.py $w 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03b_fwd_prof.log
done
