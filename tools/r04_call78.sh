#!/bin/bash
mkdir -p gpurun_out/fuzz
{
for s in 601 602 603 604; do timeout 1500 python tools/fuzz_fused.py 45 $s 2>&1 | tail -2; done
FMK_FLOW_SIDE_OHLCV=0 timeout 1500 python tools/fuzz_fused.py 30 605 2>&1 | tail -2
} > gpurun_out/fuzz/r04_fused.txt 2>&1
cut -c1-400 gpurun_out/fuzz/r04_fused.txt
