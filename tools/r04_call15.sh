mkdir -p gpurun_out/c15
timeout 600 python -m pytest tests/test_gpu_threshold.py -q -x 2>&1 | tail -3
for s in 1 2 3; do timeout 600 python tools/fuzz_whales.py $s 150 1500000 2>&1 | tail -6; done > gpurun_out/c15/fuzz_whales.txt 2>&1
cat gpurun_out/c15/fuzz_whales.txt | cut -c1-260
timeout 600 python tools/fuzz_volume.py 201 200 2000000 dollar 2>&1 | tail -3
timeout 600 python tools/whalebench.py 1e8 1e-4 1000 1 2>&1 | tail -4
