#!/bin/bash
mkdir -p gpurun_out/c52
timeout 1200 python -m pytest tests/test_gpu_features.py tests/test_gpu_fused.py tests/test_gpu_barlengths.py tests/test_gpu_core.py -m gpu -x -q 2>&1 | tail -4
{
for s in 361 362 363; do timeout 1200 python tools/fuzz_longbars.py 150 $s 2>&1 | tail -1; done
for s in 372 373 374; do timeout 1200 python tools/fuzz_longbars.py 80 $s short 2>&1 | tail -1; done
for s in 381 382; do timeout 1200 python tools/fuzz_longbars.py 100 $s mid 2>&1 | tail -1; done
} > gpurun_out/c52/fuzz.txt 2>&1
cat gpurun_out/c52/fuzz.txt
