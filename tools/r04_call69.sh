#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p gpurun_out/c69
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p69
timeout 600 rocprofv3 --kernel-trace -d /tmp/p69 -o c -- env -C $R python tools/realcfg4.py 1e9 1.0 dyadic > $R/gpurun_out/c69/log.txt 2>&1
db=$(find /tmp/p69 -name '*_results.db' | head -1)
python $R/tools/rocpd_timeline.py "$db" 400 > $R/gpurun_out/c69/timeline.txt
python $R/tools/rocpd_stats.py "$db" > $R/gpurun_out/c69/stats.csv
grep -a "cfg 4\|alone" $R/gpurun_out/c69/log.txt
