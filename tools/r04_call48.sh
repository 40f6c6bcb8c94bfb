#!/bin/bash
mkdir -p gpurun_out/c48
timeout 600 python bench.py --force-dist > gpurun_out/c48/forcedist.json 2> gpurun_out/c48/forcedist.err
timeout 600 python bench.py > gpurun_out/c48/default.json 2> gpurun_out/c48/default.err
python - <<'PY'
import json
for f in ('forcedist','default'):
    d=json.loads(open(f'gpurun_out/c48/{f}.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f, 'step', round(d['ms_per_step'],4), 'kernel', round(r['avg_kernel_ms'],4), 'diff', round(d['ms_per_step']-r['avg_kernel_ms'],4), 'frac', round(r['frac'],4), [round(x,3) for x in r.get('placement',{}).get('probe_kernel_ms',[])])
PY
tail -3 gpurun_out/c48/*.err
