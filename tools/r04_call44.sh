#!/bin/bash
mkdir -p gpurun_out/c44
timeout 600 python tools/placemap.py 144 1e9 > gpurun_out/c44/map.txt 2>&1
cat gpurun_out/c44/map.txt
