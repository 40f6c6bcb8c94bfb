#!/bin/bash
# rocprofv3 kernel stats of one command on the GPU box: tools/prof.sh NAME cmd...  -> gpurun_out/NAME_kernel_stats.csv (+ NAME.log)
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_$name
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o c -- env -C $R "$@" > $R/gpurun_out/$name.log 2>&1
db=$(find /tmp/prof_$name -name '*_results.db' | head -1)
python $R/tools/rocpd_stats.py "$db" > $R/gpurun_out/${name}_kernel_stats.csv
grep -v "rocprofv3\|SQLite\|generateRocpd\|tool.cpp" $R/gpurun_out/$name.log | tail -20
head -12 $R/gpurun_out/${name}_kernel_stats.csv | cut -c1-200
