mkdir -p gpurun_out/c34
R=$PWD
timeout 900 bash tools/pmc_sq.sh fp "k_bar_footprints|k_bar_dir_lanes|k_bar_median_small" env -C $R python tools/realcfg4.py 1e9 0 > gpurun_out/c34/sq.log 2>&1
cp gpurun_out/sq_fp.txt gpurun_out/c34/ 2>/dev/null
cat gpurun_out/c34/sq_fp.txt
