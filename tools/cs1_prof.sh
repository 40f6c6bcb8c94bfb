#!/bin/bash
# kernel trace of the CUSUM call at floor 1e-5 (1e9 ticks): where the one-pass attempt spends its time
R=$PWD; O=$R/gpurun_out/cs1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_cs1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_cs1 -o c -- env -C $R python tools/cusumbench.py 1e9 1e-5 > $O/prof_run.txt 2>&1
cd $R
python tools/rocpd_stats.py $(find /tmp/prof_cs1 -name "*.db" | head -1) > $O/prof_kernel_stats.csv
grep sigma_floor $O/prof_run.txt; grep -i "cs1\|cusum\|scan" $O/prof_kernel_stats.csv | cut -c1-160
