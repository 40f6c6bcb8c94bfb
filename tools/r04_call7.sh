mkdir -p gpurun_out/c7
timeout 600 python -m pytest tests/test_gpu_threshold.py -q -x 2>&1 | tail -3
timeout 300 python tools/fuzz_volume.py 104 300 3000000 volume 2>&1 | tail -2
for on in 1 3; do echo "FMK_VOL_EXACT_TIER=$on"; FMK_VOL_EXACT_TIER=$on timeout 200 python tools/thrbench.py 1e9 300,600,865,1100,1400 volume 2>&1 | grep "mean bar"; done > gpurun_out/c7/thrbench.txt 2>&1
cat gpurun_out/c7/thrbench.txt
