#!/bin/bash
# A/B timing of one-pass cfg-4 kernel variants (tools/ab_variant.sh) on one box: tools/fu_ab.sh TAG ...
for v in "$@"; do echo "== $v"; python tools/ab_lib.py finmlkit_amd/lib/ab/libfmk_hip_$v.so tools/cfg4bench.py 1e9 2>&1 | grep -v "tick-order" | head -2; done
echo "== default"; python tools/cfg4bench.py 1e9 2>&1 | grep -v "tick-order" | head -2
