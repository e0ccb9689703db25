for w in cfg2_sd_real cfg5_llm_n4 enc_extractor; do
 for a in "" q8; do
  echo "== $w algo=${a:-default}"
  MMFS_FWD_ALGO=$a python bench.py --no-cpu-baseline --steps 30 --warmup 5 --workload $w 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  ms/step', r['ms_per_step'], r['kernels_mean_us'])"
 done
done
