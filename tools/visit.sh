#!/bin/bash
# One GPU visit = one call of this script on the GPU box (replaces round 3's one-off tools/r03*.sh):
#   gpurun --timeout N -- 'bash tools/visit.sh TAG step [step ...]'
# Everything a step writes lands in gpurun_out/ under the visit's TAG.  Steps (ARG parts are ':'-separated):
#   tests[:EXPR]          pytest -m gpu (optionally -k EXPR), log -> gpurun_out/TAG_pytest.log
#   file:PATH[:EXPR]      pytest -m gpu on one test file
#   smoke                 __graft_entry__.smoke()
#   bench[:ARGS]          the driver's command (cpu baseline included) + ARGS -> gpurun_out/bench_TAG.json
#   line:NAME:ARGS        bench.py --no-cpu-baseline ARGS -> gpurun_out/bench_TAG_NAME.json  (ARGS: '+' for spaces)
#   env:NAME:K=V,K=V:ARGS the same with environment variables set
#   all                   bench lines of every workload the round reports
#   prof[:WORKLOAD]       tools/prof.sh (kernel trace + counter passes) -> gpurun_out/prof_TAG[_WORKLOAD]/
#   traffic:WORKLOAD      tools/pmc_traffic.sh (FETCH_SIZE / WRITE_SIZE passes only)
#   module[:CFGS]         tools/module_bench.py (default cfg3 cfg4) -> gpurun_out/TAG_module_bench.jsonl
#   exp:VARIANTS[:ARGS]   tools/exp_run.sh over csrc/build/exp/<variant>.so (built here with tools/exp_build.sh)
#   py:SCRIPT[:ARGS]      python SCRIPT ARGS -> gpurun_out/TAG_<script>.log
#   ubench:NAME           build tools/ubench/NAME.hip for gfx950 and run it -> gpurun_out/TAG_ubench_NAME.log
tag=$1; shift
mkdir -p gpurun_out
sp() { echo "${1//+/ }"; }
show() { python - "$1" <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    k = r.get("kernels_mean_us") or {}
    print("  %-44s ms/step %.4f  %s  dom %s frac %.3f" % (sys.argv[1].split('/')[-1], r["ms_per_step"],
          " ".join("%s=%.1f" % (a.replace("msda_", "").replace("bwd_", ""), b) for a, b in k.items()),
          r["roofline"]["kernel"], r["roofline"]["frac"]))
except Exception as e:
    print("  ", sys.argv[1], "no line:", e)
PY
}
line() { name=$1; shift; timeout 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/bench_${tag}_$name.json 2>gpurun_out/bench_${tag}_$name.err; show gpurun_out/bench_${tag}_$name.json; }
for step in "$@"; do
  IFS=: read -r what a1 a2 a3 <<< "$step"
  echo "== $step"
  case $what in
    tests)  timeout 2400 python -m pytest tests -m gpu -q -x ${a1:+-k "$(sp "$a1")"} > gpurun_out/${tag}_pytest.log 2>&1; tail -4 gpurun_out/${tag}_pytest.log | cut -c1-300 ;;
    file)   timeout 1500 python -m pytest "$a1" -m gpu -q -x ${a2:+-k "$(sp "$a2")"} > gpurun_out/${tag}_pytest_$(basename $a1 .py).log 2>&1; tail -15 gpurun_out/${tag}_pytest_$(basename $a1 .py).log | cut -c1-300 ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -9 ;;
    bench)  timeout 400 python bench.py --steps 20 --warmup 5 $(sp "$a1") > gpurun_out/bench_$tag.json 2>gpurun_out/bench_$tag.err; show gpurun_out/bench_$tag.json ;;
    line)   line "$a1" $(sp "$a2") ;;
    env)    ( export $(echo "$a2" | tr ',' ' '); line "$a1" $(sp "$a3") ) ;;
    all)    line cfg2_northstar --steps 100 --warmup 20
            line fresh --steps 30 --warmup 10 --fresh-levels
            line centre --steps 30 --warmup 10 --loc-dist centre
            for w in cfg2_sd_real cfg5_llm_n4 cfg1 enc_injector enc_extractor; do line $w --steps 30 --warmup 10 --workload $w; done
            line cfg5_llm_n4_causal --steps 30 --warmup 10 --workload cfg5_llm_n4 --visible causal
            line ref_speed_test_f16 --steps 50 --warmup 50 --workload ref_speed_test --grad ones
            line ref_speed_test_f32 --steps 50 --warmup 50 --workload ref_speed_test --grad ones --dtype f32 ;;
    prof)   bash tools/prof.sh ${tag}${a1:+_$a1} ${a1:+--workload $a1} > gpurun_out/prof_${tag}${a1:+_$a1}.log 2>&1; grep -A9 "== kernel stats" gpurun_out/prof_${tag}${a1:+_$a1}.log | cut -c1-170 ;;
    traffic) bash tools/pmc_traffic.sh $tag $a1 2>&1 | tail -8 ;;
    module) timeout 1200 python tools/module_bench.py ${a1:-cfg3 cfg4} > gpurun_out/${tag}_module_bench.jsonl 2>gpurun_out/${tag}_module_bench.err
            python - gpurun_out/${tag}_module_bench.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l); print(" ", r["config"], r["what"][:70], "| ms", r["ms"], r.get("kernel_us"), "launches", r.get("launches"))
PY
            tail -2 gpurun_out/${tag}_module_bench.err ;;
    exp)    bash tools/exp_run.sh "$(sp "$a1")" $(sp "$a2") ;;
    py)     timeout 1200 python $a1 $(sp "$a2") > gpurun_out/${tag}_$(basename $a1 .py).log 2>&1; tail -${VISIT_TAIL:-25} gpurun_out/${tag}_$(basename $a1 .py).log | cut -c1-220 ;;
    ubench) ( cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $a1.hip -o /tmp/ub_$a1 2>&1 | tail -5 && timeout 300 /tmp/ub_$a1 ) > gpurun_out/${tag}_ubench_$a1.log 2>&1; tail -${VISIT_TAIL:-60} gpurun_out/${tag}_ubench_$a1.log | cut -c1-200 ;;
    *)      echo "unknown step $what" ;;
  esac
done
