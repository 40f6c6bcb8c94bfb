#!/bin/bash
# pass A of the one-pass CUSUM form: the kernel, without its walk, with a subtraction for the logarithm (kernel times by rocprofv3)
R=$PWD; O=$R/gpurun_out/cs1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 0 1 2; do
  rm -rf /tmp/prof_v$v
  FMK_CS1_VARIANT=$v timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_v$v -o c -- env -C $R python tools/cusumbench.py 1e9 1e-5 > $O/variant_$v.txt 2>&1
  echo "variant $v: $(env -C $R python tools/rocpd_stats.py $(find /tmp/prof_v$v -name '*.db' | head -1) | grep k_cs1_pass | cut -c1-120)"
done
