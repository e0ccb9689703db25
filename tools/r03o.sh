#!/bin/bash
mkdir -p gpurun_out
echo "== gv tests"; timeout 600 python -m pytest tests/test_op_gpu.py -q -x -k "lds_blocks" > gpurun_out/r03o_pytest_gv.log 2>&1; tail -6 gpurun_out/r03o_pytest_gv.log | cut -c1-300
for t in 256 512; do
  echo "== phase clocks, target $t"; MMFS_GV_TARGET_WGS=$t MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/gprof.so timeout 300 python tools/gv_prof.py cfg2_northstar 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03o_gv_phase_clocks.log
done
echo "== llm"; MMFS_MSDA_LIB=$PWD/mm-interleaved_amd/csrc/build/exp/gprof.so timeout 300 python tools/gv_prof.py cfg5_llm_n4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r03o_gv_phase_clocks.log
run() { local name=$1; shift; timeout 300 env "$@" > gpurun_out/bench_r03o_$name.json 2> gpurun_out/bench_r03o_$name.err || echo "FAILED $name"; python - "$name" <<'PY'
import json, sys
try:
    r = json.load(open(f"gpurun_out/bench_r03o_{sys.argv[1]}.json")); print(sys.argv[1], r["ms_per_step"], r["kernels_mean_us"])
except Exception as e:
    print(sys.argv[1], "no result", e); print(open(f"gpurun_out/bench_r03o_{sys.argv[1]}.err").read()[-1500:])
PY
}
run ns_off MMFS_GV_ALGO=off python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_t512 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_t256 MMFS_GV_TARGET_WGS=256 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run ns_t256_seg1 MMFS_GV_TARGET_WGS=256 MMFS_GV_MAX_SEGS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run sd_on python bench.py --workload cfg2_sd_real --steps 20 --warmup 5 --no-cpu-baseline
run llm_on python bench.py --workload cfg5_llm_n4 --steps 20 --warmup 5 --no-cpu-baseline
run llm_t256 MMFS_GV_TARGET_WGS=256 python bench.py --workload cfg5_llm_n4 --steps 20 --warmup 5 --no-cpu-baseline
