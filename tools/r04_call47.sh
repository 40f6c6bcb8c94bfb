#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/c47
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/c47/avail.txt 2>&1
grep -o "\b\(TCP_UTCL1\|TCC_EA0\|TCC_TAG\|TCC_HIT\|TCC_MISS\|TCC_REQ\|TCP_PENDING\|TCP_TCC\|TCP_TA\|TCC_BUBBLE\|TCC_NORMAL\|TCC_EA0_RD\)[A-Za-z0-9_]*" $R/gpurun_out/c47/avail.txt | sort -u | tr '\n' ' ' > $R/gpurun_out/c47/names.txt
i=0
for c in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
         "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" \
         "TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum" \
         "TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_REQ_sum TCC_EA0_RD_UNCACHED_32B_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pp_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pp_$i -o p -- env -C $R python tools/placepmc.py > $R/gpurun_out/c47/pass$i.log 2>&1
  python - $i >> $R/gpurun_out/c47/counters.txt <<'PY'
import csv, glob, sys, collections
i = sys.argv[1]
fs = sorted(glob.glob(f"/tmp/pp_{i}/**/*counter_collection.csv", recursive=True))
if not fs:
    print(f"pass {i}: no counter file"); sys.exit()
rows = [r for r in csv.DictReader(open(fs[-1])) if "k_bar_ohlcv_small" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows:
    by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)[-8:]
names = sorted({c for d in ids for c in by[d]})
print(f"pass {i}: last 8 dispatches of the kernel (4 on the fast window, then 4 on the slow one)")
for c in names:
    v = [by[d].get(c, float('nan')) for d in ids]
    a, b = sum(v[:4]) / 4, sum(v[4:]) / 4
    print(f"  {c:40s} fast {a:14.5g}  slow {b:14.5g}  ratio {b / a if a else float('nan'):.4f}")
PY
  grep -h "fast window\|kernel us" $R/gpurun_out/c47/pass$i.log | cut -c1-200 >> $R/gpurun_out/c47/counters.txt
done
cat $R/gpurun_out/c47/names.txt; echo; cat $R/gpurun_out/c47/counters.txt
