mkdir -p gpurun_out/c31
bash tools/prof.sh c31/vol python tools/thrbench.py 1e9 865 volume 2>&1 | tail -12 | cut -c1-150
bash tools/prof.sh c31/dol python tools/cfgbench.py 2>&1 | tail -3 | cut -c1-200
grep -E "k_dl|k_dlx" gpurun_out/c31/dol_kernel_stats.csv | cut -c1-160
