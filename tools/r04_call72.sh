#!/bin/bash
mkdir -p gpurun_out/c72
for rep in 1 2; do for v in "8 2" "12 2" "16 2" "12 4" "16 4" "10 3"; do set -- $v
  FMK_TB_PIPE_SPLIT=$1 FMK_TB_PIPE_IDX_BPC=$2 timeout 300 python bench.py --no-extras --cpu-sample 0 --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('split $1 idx_bpc $2: step %.4f kernel %.4f diff %.4f best probe %.3f' % (d['ms_per_step'], r['avg_kernel_ms'], d['ms_per_step']-r['avg_kernel_ms'], min(r['placement']['probe_kernel_ms'])))"
done; done | tee gpurun_out/c72/sweep.txt
