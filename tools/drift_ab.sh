#!/bin/bash
# Headline drift: on ONE box, interleave R runs each of the round-1, round-2 and round-3 end
# libraries (built from 2349744^, 80570ca^ and bee7b92^ into finmlkit_amd/lib/ab/) and the current library, each run a
# fresh process that allocates its own inputs:  bench.py --no-extras --cpu-sample 0 --steps 20 --warmup 5 --separate-index
# --placements 1 (the two-call step every library has; ONE allocation of the inputs, no choice).
# Prints one line per run: library, ms_per_step, avg dominant-kernel ms, roofline fraction.
R=${1:-6}
OUT=${2:-gpurun_out/r04_drift.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
LIBS="r1 r2 r3 head"
for r in $(seq 1 $R); do
  for l in $LIBS; do
    if [ $l = head ]; then lib=finmlkit_amd/lib/libfmk_hip.so; else lib=finmlkit_amd/lib/ab/libfmk_hip_$l.so; fi
    [ -f $lib ] || continue
    line=$(timeout 300 python tools/ab_lib.py $lib bench.py --no-extras --cpu-sample 0 --steps 20 --warmup 5 --separate-index --placements 1 2>/dev/null | tail -1)
    echo "$line" | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print('round $r lib %-4s step %.3f ms  kernel %.3f ms  frac %.3f' % ('$l', d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac']))
except Exception as e:
    print('round $r lib $l FAILED', e)
" | tee -a "$OUT"
  done
done
python - "$OUT" <<'PY' | tee -a "$OUT"
import sys, re, collections
rows = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m = re.match(r"round (\d+) lib (\S+)\s+step ([\d.]+) ms\s+kernel ([\d.]+) ms", ln)
    if m: rows[m.group(2)].append((float(m.group(3)), float(m.group(4))))
for k, v in rows.items():
    s = sorted(x[0] for x in v); q = sorted(x[1] for x in v)
    print("SUMMARY lib %-4s runs %d  step ms min %.3f median %.3f max %.3f | kernel ms min %.3f median %.3f max %.3f" %
          (k, len(v), s[0], s[len(s)//2], s[-1], q[0], q[len(q)//2], q[-1]))
PY
