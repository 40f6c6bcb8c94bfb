#!/bin/bash
# the one-pass CUSUM form against the fixed point over stream sizes (floor 1e-5): where does it start to pay?
O=gpurun_out/cs1; mkdir -p $O
for n in 3e5 1e6 3e6 1e7 3.9e7 1e8 3e8; do
  a=$(FMK_CUSUM_ONEPASS=1 timeout 300 python tools/cusumbench.py $n 1e-5 2>&1 | grep sigma_floor | awk '{print $3}' | sort -n | head -1)
  b=$(FMK_CUSUM_ONEPASS=0 timeout 300 python tools/cusumbench.py $n 1e-5 2>&1 | grep sigma_floor | awk '{print $3}' | sort -n | head -1)
  u=$(FMK_CUSUM_ONEPASS=1 timeout 300 python tools/cusumbench.py $n 1e-5 2>&1 | grep "one pass" | tail -1)
  echo "n=$n  one pass $a ms   fixed point $b ms   |$u"
done | tee $O/sizes.txt
