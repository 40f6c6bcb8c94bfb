#!/bin/bash
mkdir -p gpurun_out/c46
timeout 600 python tools/placeprobe.py 144 > gpurun_out/c46/probe.txt 2>&1
cat gpurun_out/c46/probe.txt
