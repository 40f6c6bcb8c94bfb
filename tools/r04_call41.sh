#!/bin/bash
mkdir -p gpurun_out/c41
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_timebars_fused.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/c41/pytest.txt
for v in "1 1" "1 1" "0 1"; do set -- $v
  FMK_DIST_ONE_CALL=$1 FMK_TB_PIPE_EO_CENSUS=$2 timeout 200 python tools/distab.py 1e9 20 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee -a gpurun_out/c41/distab.txt
done
timeout 400 python bench.py --force-dist --no-extras --steps 20 > gpurun_out/c41/forcedist.json 2> gpurun_out/c41/forcedist.err
timeout 400 python bench.py > gpurun_out/c41/default.json 2> gpurun_out/c41/default.err
python - <<'PY'
import json
for f in ('forcedist','default'):
    d=json.loads(open(f'gpurun_out/c41/{f}.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f, 'step', round(d['ms_per_step'],4), 'kernel', round(r['avg_kernel_ms'],4), 'diff', round(d['ms_per_step']-r['avg_kernel_ms'],4), 'frac', round(r['frac'],4))
    o=d.get('other_configs',{})
    print({k:round(v,2) for k,v in o.items() if k.startswith('cfg4') and isinstance(v,float)})
PY
