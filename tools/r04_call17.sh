mkdir -p gpurun_out/c17
for s in 21 22; do timeout 900 python tools/fuzz_whales.py $s 250 2000000 2>&1 | tail -4; done > gpurun_out/c17/fuzz_whales.txt 2>&1
cat gpurun_out/c17/fuzz_whales.txt | cut -c1-260
timeout 600 python tools/fuzz_volume.py 202 200 2000000 dollar 2>&1 | tail -2; timeout 900 python -m pytest tests/test_gpu_threshold.py -q -x 2>&1 | tail -3
{ timeout 900 python tools/whalebench.py 1e9 1e-4 1000 1 2>&1 | tail -3
timeout 900 python tools/whalebench.py 1e9 1e-4 10000 1 2>&1 | tail -3
timeout 900 python tools/whalebench.py 1e9 1e-3 3000 1 2>&1 | tail -3; } > gpurun_out/c17/whalebench.txt 2>&1
cat gpurun_out/c17/whalebench.txt
bash tools/prof.sh c17/whale python tools/whalebench.py 1e9 1e-4 1000 0 2>&1 | tail -14 | cut -c1-150
