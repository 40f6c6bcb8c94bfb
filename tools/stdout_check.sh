#!/bin/bash
# bench.py's stdout must be exactly ONE line (the JSON line), also when librccl comes up (its version banner is a printf on fd 1).
mkdir -p gpurun_out/stdout_check; O=gpurun_out/stdout_check
timeout 900 python -m pytest tests/test_gpu_dist.py -q 2>&1 | tail -4 > $O/pytest_dist_tail.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --cpu-sample 0 > $O/plain.out 2> $O/plain.err
timeout 600 python bench.py --force-dist --steps 20 --warmup 5 > $O/forcedist.out 2> $O/forcedist.err
cat $O/pytest_dist_tail.txt
for f in plain forcedist; do echo "$f: stdout lines $(wc -l < $O/$f.out), parses as one JSON document: $(python -c "import json,sys; d=json.load(open('$O/$f.out')); print('yes, frac %.3f step %.3f ms' % (d['roofline']['frac'], d['ms_per_step']))" 2>&1 | tail -1)"; echo "   banner on stderr: $(grep -c 'RCCL version' $O/$f.err)"; done
