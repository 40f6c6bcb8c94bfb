#!/bin/bash
mkdir -p gpurun_out/c40
for v in "1 1 0" "1 1 1" "0 1 1" "1 1 0" "0 1 1"; do set -- $v
  echo "aux priority knob $3 (0 = default priority, 1 = lowest):" | tee -a gpurun_out/c40/distab.txt
  FMK_DIST_ONE_CALL=$1 FMK_TB_PIPE_EO_CENSUS=$2 FMK_AUX_PRIORITY=$3 timeout 200 python tools/distab.py 1e9 20 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee -a gpurun_out/c40/distab.txt
done
for pr in 0 1 0 1; do
  FMK_AUX_PRIORITY=$pr timeout 300 python bench.py --no-extras --steps 20 > gpurun_out/c40/plain_pr$pr.json 2>/dev/null
  python - <<PY | tee -a gpurun_out/c40/plain.txt
import json
d=json.loads(open('gpurun_out/c40/plain_pr$pr.json').read().strip().splitlines()[-1]); r=d['roofline']
print('plain bench aux priority knob $pr: step', round(d['ms_per_step'],4), 'kernel', round(r['avg_kernel_ms'],4), 'diff', round(d['ms_per_step']-r['avg_kernel_ms'],4))
PY
done
