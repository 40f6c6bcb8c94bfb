mkdir -p gpurun_out/c32
timeout 600 python -m pytest tests/test_gpu_threshold.py -q -x 2>&1 | tail -2
timeout 600 python tools/fuzz_volume.py 401 300 3000000 volume 2>&1 | tail -1
for on in 1 4; do echo "FMK_VOL_EXACT_TIER=$on"; FMK_VOL_EXACT_TIER=$on timeout 200 python tools/thrbench.py 1e9 300,600,700,865,1000,1400 volume 2>&1 | grep "mean bar"; done > gpurun_out/c32/thrbench.txt 2>&1
cat gpurun_out/c32/thrbench.txt
FMK_DL_FORCE_EXACT_TIER=1 timeout 300 python tools/whalebench.py 1e9 0 1000 0 2>&1 | tail -1
