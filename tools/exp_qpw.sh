for q in 128 256 512 1024; do echo "== QPW $q"; MMFS_FWD_WQ_QPW=$q python tools/fwd_repeat.py 2>&1 | grep "random value again"; done
