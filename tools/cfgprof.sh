#!/bin/bash
# The offline profiles of bench.py's secondary configs, on the GPU box: tools/cfgprof.sh [ticks] [KEY ...]
# per KEY three runs of tools/cfgprof.py: rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE and --pmc WRITE_SIZE on their own
# (counters with --kernel-trace only, as gpurun requires); then tools/cfgprof_summarize.py -> gpurun_out/final/ + profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1e9}; shift
KEYS=${@:-cfg3_volume_index cfg3_volume_build_ohlcv cfg3_dollar_index cfg3_dollar_build_ohlcv cfg4_equal_bars cfg4_equal_bars_full_mantissa cfg4_lognormal_full_mantissa lagged_returns_5s ewmst_60s cusum_floor_5e-4 cusum_floor_1e-5}
O=$R/gpurun_out/cfgprof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for k in $KEYS; do
  rm -rf $O/${k}_trace $O/${k}_FETCH_SIZE $O/${k}_WRITE_SIZE
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/${k}_trace -o t -- env -C $R python tools/cfgprof.py $k $N 3 > $O/${k}_trace.log 2>&1
  grep CFGPROF $O/${k}_trace.log
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${k}_$c -o p -- env -C $R python tools/cfgprof.py $k $N 1 > $O/${k}_$c.log 2>&1
  done
done
cd $R && python tools/cfgprof_summarize.py
# the databases are large: only the summaries travel back
rm -rf $O/*_trace $O/*_FETCH_SIZE $O/*_WRITE_SIZE
