#!/bin/bash
mkdir -p gpurun_out
echo "== module tests"; timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_zz_graph_gpu.py -q -x 2>&1 | tail -2
echo "== module bench cfg4"
timeout 900 python tools/module_bench.py cfg4 > gpurun_out/r03am_module_bench_cfg4.jsonl 2>gpurun_out/r03am_module_bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r03am_module_bench_cfg4.jsonl"):
    r = json.loads(l); print(r["what"][:120], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"])
PY
tail -2 gpurun_out/r03am_module_bench.err
