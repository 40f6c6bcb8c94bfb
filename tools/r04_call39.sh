#!/bin/bash
mkdir -p gpurun_out/c39
for v in "1 1 0" "1 0 0" "1 1 1" ; do set -- $v
  echo "aux priority knob $3 (0 = default priority, 1 = lowest):" | tee -a gpurun_out/c39/distab.txt
  FMK_DIST_ONE_CALL=$1 FMK_TB_PIPE_EO_CENSUS=$2 FMK_AUX_PRIORITY=$3 timeout 200 python tools/distab.py 1e9 20 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee -a gpurun_out/c39/distab.txt
done
