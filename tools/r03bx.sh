#!/bin/bash
# last check of the round: whole GPU suite, smoke, the driver's command, the LLM module lines
mkdir -p gpurun_out
echo "== all gpu tests"; timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03bx_pytest_all.log 2>&1; tail -2 gpurun_out/r03bx_pytest_all.log | cut -c1-200
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 250 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03bx_driver.json 2>/dev/null
python - <<'PY'
import json
r = json.load(open("gpurun_out/bench_r03bx_driver.json")); print(r["ms_per_step"], r["value"], r["roofline"]["kernel"], r["roofline"]["frac"], r["kernels_frac"])
PY
timeout 900 python tools/module_bench.py cfg3 > gpurun_out/r03bx_module_bench_cfg3.jsonl 2>gpurun_out/r03bx_module_bench.err; python - <<'PY'
import json
for l in open("gpurun_out/r03bx_module_bench_cfg3.jsonl"):
    r = json.loads(l)
    if "forward+backward" in r["what"] and "2048" not in r["what"]: continue
    print(r["config"], r["what"].split("B=")[1][:12], r["what"].split("bf16, ")[-1][:80], "| ms", r["ms"], r["kernel_us"], "launches", r["launches"])
PY
