#!/bin/bash
mkdir -p gpurun_out/c70
timeout 1500 python -m pytest tests/test_gpu_ticklevel.py tests/test_gpu_next.py tests/test_gpu_cusum.py tests/test_gpu_dist.py tests/test_gpu_refcalls.py tests/test_gpu_kits.py -m gpu -x -q 2>&1 | tail -5
{
for s in 491 492 493; do timeout 1200 python tools/fuzz_parity.py $s 2500 2>&1 | tail -1; done
} > gpurun_out/c70/fuzz.txt 2>&1; cat gpurun_out/c70/fuzz.txt
timeout 600 python tools/tlbench.py 1e9 2>&1 | tail -6 | tee gpurun_out/c70/tlbench.txt
