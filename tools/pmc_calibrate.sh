#!/bin/bash
# Two separate counter passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; --kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_r02_$c -o p -- python $R/tools/pmc_calibrate.py "$@" > $R/gpurun_out/pmc_r02_$c.log 2>&1
  grep -E "^probe|^dominant" $R/gpurun_out/pmc_r02_$c.log
  find $R/gpurun_out/pmc_r02_$c -name "*counter_collection.csv" | head -2
done
