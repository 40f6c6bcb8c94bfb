#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
K="k_bar_dir_lanes|k_bar_dir<|k_bar_footprints<|k_bar_median_small|k_bar_ohlcv_phased|k_bar_ohlcv_small|k_bar_median<|k_bar_median_long"
bash tools/pmc_sq.sh cfg4_lognormal_head "$K" env -C $R python tools/realcfg4.py 1e9 1.0 > /dev/null 2>&1
bash tools/pmc_sq.sh cfg4_lognormal_r3 "$K" env -C $R python tools/ab_lib.py finmlkit_amd/lib/ab/libfmk_hip_r3.so tools/realcfg4.py 1e9 1.0 > /dev/null 2>&1
for t in head r3; do echo "== $t"; grep -a "^# kernel\|SQ_WAVES\|SQ_WAVE_CYCLES\|SQ_BUSY_CYCLES\|SQ_INSTS_VALU " gpurun_out/sq_cfg4_lognormal_$t.txt | cut -c1-110; done
tail -3 gpurun_out/sqp_1.log | cut -c1-200
