#!/bin/bash
mkdir -p gpurun_out/c51
timeout 1200 python tools/fuzz_longbars.py 150 361 > gpurun_out/c51/long361.txt 2>&1
grep -a "FAIL" gpurun_out/c51/long361.txt | cut -c1-1800 | head -5; tail -1 gpurun_out/c51/long361.txt
timeout 1200 python tools/fuzz_longbars.py 60 371 short > gpurun_out/c51/short371.txt 2>&1
grep -a "FAIL" gpurun_out/c51/short371.txt | cut -c1-1200 | head -3; tail -1 gpurun_out/c51/short371.txt
timeout 600 python tools/shortbars.py 1e9 2.5,3,4,5,7.5,10 > gpurun_out/c51/shortbars.txt 2>&1; cat gpurun_out/c51/shortbars.txt
