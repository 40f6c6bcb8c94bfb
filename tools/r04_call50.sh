#!/bin/bash
# fuzz of the paths touched after the first campaign: sharded step through the one-call entry, boundary bar in one launch, auxiliary stream priority
mkdir -p gpurun_out/fuzz
{
for s in 2 3 4; do timeout 900 python tools/fuzz_sharded.py 60 $s 2>&1 | tail -2; done
for s in 351 352; do timeout 1200 python tools/fuzz_parity.py $s 2500 2>&1 | tail -1; done
timeout 900 python tools/fuzz_longbars.py 150 361 2>&1 | tail -1
} > gpurun_out/fuzz/r04_campaign2.txt 2>&1
cat gpurun_out/fuzz/r04_campaign2.txt | cut -c1-220
