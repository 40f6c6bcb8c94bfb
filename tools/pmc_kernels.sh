#!/bin/bash
# HBM traffic of named kernels: two counter passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as gpurun requires) of one
# command, per-kernel means in bytes with the corrections of profiles/traffic_constants.json (FETCH_SIZE x2, WRITE_SIZE x1; KiB).
# usage (on the GPU box): tools/pmc_kernels.sh "<kernel substring>[,<substring>...]" <command ...>
K=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmck_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmck_$c -o p -- "$@" > $R/gpurun_out/pmck_$c.log 2>&1
done
python - "$K" "$R" <<'PY'
import csv, glob, sys, collections
keys, root = sys.argv[1].split(","), sys.argv[2]
for c, fac in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    f = sorted(glob.glob(f"{root}/gpurun_out/pmck_{c}/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for k in keys:
            if k in r["Kernel_Name"]:
                acc[k].append(float(r["Counter_Value"]) * 1024.0 * fac)
    for k, v in acc.items():
        print(f"{c:10s} {k:28s} launches {len(v):4d}  mean {sum(v) / len(v) / 1e9:9.3f} GB  max {max(v) / 1e9:9.3f} GB")
PY
rm -rf $R/gpurun_out/pmck_FETCH_SIZE $R/gpurun_out/pmck_WRITE_SIZE
