tools/exp_run.sh "sp1024 sp2048 sp4096" 2>&1 | grep -v Warn | cut -c1-120
