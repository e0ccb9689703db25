timeout 300 python -m pytest tests/test_op_gpu.py -q -x -k "65536 or wider" 2>&1 | tail -2
tools/exp_run.sh "tc512 tc2048 tc4096" 2>&1 | grep -v Warn | cut -c1-70
for w in cfg2_sd_real cfg5_llm_n4; do for v in base tc2048 tc4096; do if [ $v = base ]; then lib=mm-interleaved_amd/libmmfs_msda.so; else lib=mm-interleaved_amd/csrc/build/exp/$v.so; fi; echo -n "$w $v "; MMFS_MSDA_LIB=$PWD/$lib timeout 100 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --workload $w 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['ms_per_step'], r['kernels_mean_us']['msda_bwd_value_reduce'])"; done; done
