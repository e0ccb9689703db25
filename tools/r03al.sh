#!/bin/bash
mkdir -p gpurun_out
echo "== module tests"; timeout 900 python -m pytest tests/test_modules_gpu.py tests/test_zz_graph_gpu.py -q -x 2>&1 | tail -3
timeout 300 python tools/decode_kernels.py 1 2>&1 | grep -v "amdgpu.ids\|Warning\|_warn_once" | tee gpurun_out/r03al_decode_kernels.log | head -30 | cut -c1-170
echo "== module bench cfg3"
timeout 600 python tools/module_bench.py cfg3 > gpurun_out/r03al_module_bench_cfg3.jsonl 2>gpurun_out/r03al_module_bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r03al_module_bench_cfg3.jsonl"):
    r = json.loads(l); print(r["what"][30:], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"])
PY
