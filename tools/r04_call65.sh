#!/bin/bash
for v in default pf1 pf2 default pf1; do
  if [ $v = default ]; then L=finmlkit_amd/lib/libfmk_hip.so; else L=finmlkit_amd/lib/ab/libfmk_hip_$v.so; fi
  echo "== $v"
  timeout 600 python tools/ab_lib.py $L tools/cfg4bench.py 1e9 2>&1 | tail -2 | cut -c1-75
  timeout 600 python tools/ab_lib.py $L tools/fpbench.py 1000000000 2>&1 | tail -1 | cut -c1-120
done
