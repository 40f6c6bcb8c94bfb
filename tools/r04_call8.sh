mkdir -p gpurun_out/c8
R=$PWD
timeout 900 bash tools/pmc_sq.sh vx "k_vx_level0|k_vx_level_up|k_vx_emit" env -C $R python tools/thrbench.py 1e9 865 volume > gpurun_out/c8/sq.log 2>&1
cp gpurun_out/sq_vx.txt gpurun_out/c8/ 2>/dev/null
cat gpurun_out/c8/sq_vx.txt
