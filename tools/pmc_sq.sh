#!/bin/bash
# SQ counters of named kernels: two counter passes (--kernel-trace only, as gpurun requires) of one command, per-kernel means.
# usage (on the GPU box): tools/pmc_sq.sh <tag> "<kernel substring>[|<substring>...]" (demangled names) <command ...>   -> gpurun_out/sq_<tag>.txt
TAG=$1; K=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
P2="SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
i=0
for c in "$P1" "$P2"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/sqp_$i
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/sqp_$i -o p -- "$@" > $R/gpurun_out/sqp_$i.log 2>&1
done
python - "$K" "$R" > $R/gpurun_out/sq_$TAG.txt <<'PY'
import csv, glob, sys, collections
keys, root = sys.argv[1].split("|"), sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    for f in sorted(glob.glob(f"{root}/gpurun_out/sqp_{i}/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            for k in keys:
                if k in r["Kernel_Name"]:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(f"# kernel {k}")
    for c, v in acc[k].items():
        v = sorted(v)
        print(f"{c:24s} launches {len(v):4d}  mean {sum(v) / len(v):12.4g}  median {v[len(v) // 2]:12.4g}  max {v[-1]:12.4g}")
PY
rm -rf $R/gpurun_out/sqp_1 $R/gpurun_out/sqp_2
cat $R/gpurun_out/sq_$TAG.txt
