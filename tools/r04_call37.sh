#!/bin/bash
# sharded step through the one-call entry: with / without the census wait in enqueue-only mode, and a timeline of each
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c37
timeout 600 python -m pytest tests/test_gpu_dist.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/c37/pytest.txt
for c in 1 0; do
  FMK_TB_PIPE_EO_CENSUS=$c timeout 300 python bench.py --force-dist --no-extras > gpurun_out/c37/fd_census$c.json 2> gpurun_out/c37/fd_census$c.err
  cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/p37_$c
  FMK_TB_PIPE_EO_CENSUS=$c timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/p37_$c -o c -- env -C $R python bench.py --force-dist --no-extras --steps 6 > $R/gpurun_out/c37/prof$c.log 2>&1
  db=$(find /tmp/p37_$c -name '*_results.db' | head -1)
  python $R/tools/rocpd_timeline.py "$db" 260 > $R/gpurun_out/c37/timeline$c.txt
  cd $R
done
cat gpurun_out/c37/pytest.txt
python - <<'PY'
import json
for c in (1,0):
    d=json.loads(open(f'gpurun_out/c37/fd_census{c}.json').read().strip().splitlines()[-1])
    print(c, d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['per_rank']['exchange_ms'])
PY
