cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|name)?\s*:?\s*(TA_|TCP_|TD_|TCC_)" | head -150 > $GRAFT_REPO_ROOT/gpurun_out/counters.txt
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC|SQ|SQC|GRBM)_[A-Z0-9_a-z]+" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/counters_all.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/counters_all.txt
