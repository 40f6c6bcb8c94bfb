#!/bin/bash
# the headline kernel with its raw words pinned behind the last load (all loads complete, then all arithmetic): A/B on ONE allocation
cat > /tmp/pinab.py <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import finmlkit_amd._ffi as ffi
PY
for r in 1 2; do for v in default pin1 pin2; do
  if [ $v = default ]; then L=finmlkit_amd/lib/libfmk_hip.so; else L=finmlkit_amd/lib/ab/libfmk_hip_$v.so; fi
  timeout 300 python tools/ab_lib.py $L bench.py --no-extras --cpu-sample 0 --steps 20 --placements 7 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'step', round(d['ms_per_step'],4), 'kernel', round(r['avg_kernel_ms'],4), 'probes', sorted(round(x,3) for x in r['placement']['probe_kernel_ms'])[:3])"
done; done
