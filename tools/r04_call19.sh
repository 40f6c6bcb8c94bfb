mkdir -p gpurun_out/c19
FMK_DL_FORCE_EXACT_TIER=1 bash tools/prof.sh c19/w0 python tools/whalebench.py 1e9 0 1000 0 2>&1 | grep -E "k_dl_tile_sums|k_dl_emit|block trades" | cut -c1-150
bash tools/prof.sh c19/w1 python tools/whalebench.py 1e9 1e-6 1000 0 2>&1 | grep -E "k_dl_tile_sums|k_dl_emit|block trades" | cut -c1-150
timeout 600 python -m pytest tests/test_gpu_threshold.py -q -x -k block_trades 2>&1 | grep -E "Error|assert|^E " | head -20
