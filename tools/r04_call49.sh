#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
bash tools/prof.sh c49_dollar python tools/dollarprof.py 1e9 4 dollar
head -40 gpurun_out/c49_dollar_kernel_stats.csv | cut -c1-160
