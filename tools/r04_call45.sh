#!/bin/bash
mkdir -p gpurun_out/c45
timeout 600 python tools/placemap2.py 144 > gpurun_out/c45/map2.txt 2>&1
cat gpurun_out/c45/map2.txt
