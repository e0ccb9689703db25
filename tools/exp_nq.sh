python tools/fwd_nq.py
for nq in 256 512 1024 2048; do for a in v m; do echo "== nq $nq taps=$a"; MMFS_TAPS_ALGO=$a python bench.py --no-cpu-baseline --steps 30 --warmup 5 --nq $nq 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  ms/step', r['ms_per_step'], r['kernels_mean_us'])"; done; done
python bench.py --no-cpu-baseline --steps 30 --warmup 5 --workload ref_speed_test --grad ones 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('speed test ms/step', r['ms_per_step'], r['kernels_mean_us'])"
