#!/bin/bash
mkdir -p gpurun_out/c62
timeout 1500 python -m pytest tests/test_gpu_threshold.py -m gpu -x -q 2>&1 | tail -3
{
for s in 441 442 443; do timeout 900 python tools/fuzz_volume.py $s 300 3000000 dollar 2>&1 | tail -1; done
for s in 451 452 453; do timeout 900 python tools/fuzz_whales.py $s 300 3000000 2>&1 | tail -1; done
} > gpurun_out/c62/fuzz.txt 2>&1
cat gpurun_out/c62/fuzz.txt | cut -c1-200
bash tools/prof.sh c62_dollar python tools/dollarprof.py 1e9 4 dollar | head -12
