#!/bin/bash
mkdir -p gpurun_out/c57
timeout 1200 python -m pytest tests/test_gpu_barlengths.py tests/test_gpu_core.py tests/test_gpu_f32amounts.py -m gpu -x -q 2>&1 | tail -3
{
for s in 401 402 403 404; do timeout 1200 python tools/fuzz_longbars.py 80 $s short 2>&1 | tail -1; done
} > gpurun_out/c57/fuzz.txt 2>&1
cat gpurun_out/c57/fuzz.txt
for h in 33 0; do echo "FMK_OHLCV_HALF_MIN_MEAN=$h"; FMK_OHLCV_HALF_MIN_MEAN=$h timeout 600 python tools/shortbars.py 1e9 1.3,1.7,2,2.3,2.5,3 2>&1 | grep "median=True"; done | tee gpurun_out/c57/half.txt
