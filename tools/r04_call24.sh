mkdir -p gpurun_out/c24
timeout 900 python -m pytest tests/test_gpu_ticklevel.py tests/test_gpu_features.py tests/test_gpu_dist.py -q -x 2>&1 | tail -3
for k in 1 0; do echo "FMK_EW_STORE_ALPHA=$k"; FMK_EW_STORE_ALPHA=$k timeout 300 python tools/tlbench.py 1e9 2>&1 | head -4; done > gpurun_out/c24/ewmst.txt 2>&1
cat gpurun_out/c24/ewmst.txt
bash tools/prof.sh c24/ew python tools/tlbench.py 1e9 2>&1 | grep -E "k_ew_" | cut -c1-140
timeout 600 python tools/fuzz_parity.py 4403 400 2>&1 | tail -2
