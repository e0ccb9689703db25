#!/bin/bash
mkdir -p gpurun_out
echo "== all gpu tests"; timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r03l_pytest_all.log 2>&1; tail -5 gpurun_out/r03l_pytest_all.log | cut -c1-250
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== plan round trip"; timeout 300 python tools/plan_roundtrip.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03l_plan_roundtrip.log
echo "== bench (driver's command)"; timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03l_cfg2_northstar.json 2>gpurun_out/bench_r03l.err; python - <<'PY'
import json
r = json.load(open("gpurun_out/bench_r03l_cfg2_northstar.json")); print(r["ms_per_step"], r["value"], r["kernels_mean_us"], r["roofline"])
PY
