#!/bin/bash
mkdir -p gpurun_out/c42
timeout 400 python bench.py --force-dist > gpurun_out/c42/forcedist.json 2> gpurun_out/c42/forcedist.err
timeout 400 python bench.py --force-dist --no-extras --steps 50 > gpurun_out/c42/forcedist50.json 2>> gpurun_out/c42/forcedist.err
python - <<'PY'
import json
for f in ('forcedist','forcedist50'):
    d=json.loads(open(f'gpurun_out/c42/{f}.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f, 'step', round(d['ms_per_step'],4), 'kernel', round(r['avg_kernel_ms'],4), 'diff', round(d['ms_per_step']-r['avg_kernel_ms'],4), 'frac', round(r['frac'],4), r.get('placement',{}).get('probe_kernel_ms'))
PY
