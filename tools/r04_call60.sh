#!/bin/bash
mkdir -p gpurun_out/c60
timeout 1500 python -m pytest tests/test_gpu_barlengths.py tests/test_gpu_core.py tests/test_gpu_f32amounts.py tests/test_gpu_fuzz.py tests/test_gpu_kits.py tests/test_gpu_refcalls.py tests/test_gpu_timebars_fused.py tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -3
{
for s in 421 422 423 424 425 426; do timeout 1200 python tools/fuzz_longbars.py 80 $s short 2>&1 | tail -1; done
for s in 431 432; do timeout 1200 python tools/fuzz_parity.py $s 2500 2>&1 | tail -1; done
} > gpurun_out/c60/fuzz.txt 2>&1
cat gpurun_out/c60/fuzz.txt
timeout 900 python tools/shortbars.py 1e9 1,1.3,1.7,2,2.3,2.5,3,3.5,4,5,7.5,10,15 2>&1 | tee gpurun_out/c60/shortbars.txt
