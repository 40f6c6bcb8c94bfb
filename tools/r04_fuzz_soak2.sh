#!/bin/bash
# soak: every fuzz tool with fresh seeds until ~20 minutes are spent; one line per run
mkdir -p gpurun_out/fuzz
T0=$(date +%s); s=900
{
while [ $(( $(date +%s) - T0 )) -lt 1200 ]; do
  s=$((s+1))
  timeout 900 python tools/fuzz_volume.py $s 250 3000000 volume 2>&1 | tail -1
  timeout 900 python tools/fuzz_volume.py $((s+1000)) 250 3000000 dollar 2>&1 | tail -1
  timeout 900 python tools/fuzz_whales.py $((s+2000)) 200 3000000 2>&1 | tail -1
  timeout 1200 python tools/fuzz_parity.py $((s+3000)) 2500 2>&1 | tail -1
  timeout 900 python tools/fuzz_longbars.py 60 $((s+4000)) 2>&1 | tail -1
  timeout 900 python tools/fuzz_longbars.py 50 $((s+5000)) short 2>&1 | tail -1
  timeout 900 python tools/fuzz_longbars.py 50 $((s+6000)) mid 2>&1 | tail -1
  timeout 900 python tools/fuzz_fused.py 30 $((s+7000)) 2>&1 | tail -1
  timeout 900 python tools/fuzz_sharded.py 30 $((s+8000)) 2>&1 | tail -1
done
} > gpurun_out/fuzz/r04_soak2.txt 2>&1
grep -c "" gpurun_out/fuzz/r04_soak2.txt; grep -v " 0 failures" gpurun_out/fuzz/r04_soak2.txt | cut -c1-300 | head
