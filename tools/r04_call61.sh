#!/bin/bash
mkdir -p gpurun_out/c61
for i in 1 2 3; do
timeout 400 python bench.py --force-dist --no-extras --cpu-sample 0 --steps 20 > gpurun_out/c61/fd$i.json 2> gpurun_out/c61/fd$i.err
python - $i <<'PY'
import json,sys
d=json.loads(open(f'gpurun_out/c61/fd{sys.argv[1]}.json').read().strip().splitlines()[-1]); r=d['roofline']; pl=r.get('placement',{})
print('force-dist', [round(x,3) for x in pl.get('probe_kernel_ms',[])], pl.get('chosen'), [round(x,3) for x in pl.get('settle_kernel_ms',[])], 'run kernel', round(r['avg_kernel_ms'],3), 'step', round(d['ms_per_step'],3))
PY
done
