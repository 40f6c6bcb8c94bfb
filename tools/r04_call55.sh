#!/bin/bash
mkdir -p gpurun_out/c55
timeout 1200 python -m pytest tests/test_gpu_barlengths.py tests/test_gpu_core.py tests/test_gpu_f32amounts.py -m gpu -x -q 2>&1 | tail -3
{
for s in 391 392 393; do timeout 1200 python tools/fuzz_longbars.py 80 $s short 2>&1 | tail -1; done
} > gpurun_out/c55/fuzz.txt 2>&1
cat gpurun_out/c55/fuzz.txt
timeout 600 python tools/shortbars.py 1e9 3,4,5,7.5,10 2>&1 | tee gpurun_out/c55/shortbars.txt
