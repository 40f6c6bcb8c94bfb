#!/bin/bash
mkdir -p gpurun_out/c63
timeout 1800 python -m pytest tests/test_gpu_threshold.py tests/test_gpu_core.py tests/test_gpu_f32amounts.py tests/test_gpu_next.py tests/test_gpu_barlengths.py tests/test_gpu_refcalls.py -m gpu -x -q 2>&1 | tail -3
{
timeout 900 python tools/fuzz_volume.py 461 300 3000000 dollar 2>&1 | tail -1
for s in 471 472; do timeout 1200 python tools/fuzz_parity.py $s 2500 2>&1 | tail -1; done
timeout 1200 python tools/fuzz_longbars.py 100 481 mid 2>&1 | tail -1
} > gpurun_out/c63/fuzz.txt 2>&1
cut -c1-160 gpurun_out/c63/fuzz.txt
bash tools/prof.sh c63_dollar python tools/dollarprof.py 1e9 4 dollar | head -7
