#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; mkdir -p $R/gpurun_out/c79
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p79
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/p79 -o c -- env -C $R python tools/dollarprof.py 1e9 3 dollar > $R/gpurun_out/c79/log.txt 2>&1
db=$(find /tmp/p79 -name '*_results.db' | head -1)
python $R/tools/rocpd_timeline.py "$db" 60 > $R/gpurun_out/c79/timeline.txt
cut -c1-118 $R/gpurun_out/c79/timeline.txt | tail -32
