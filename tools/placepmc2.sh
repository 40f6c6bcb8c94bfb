#!/bin/bash
# tools/placepmc2.py under four counter passes; per pass: mean of every counter over the k_bar_ohlcv_small dispatches of the SLOW and of the FAST region
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1e9}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/placepmc2; mkdir -p $O
P[1]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
P[2]="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_BUSY_sum"
P[3]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
P[4]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
for i in 1 2 3 4; do
  rm -rf $O/p$i
  timeout 900 rocprofv3 --kernel-trace --pmc ${P[$i]} --output-format csv -d $O/p$i -o p -- env -C $R python tools/placepmc2.py $N 13 4 > $O/p$i.log 2>&1
  grep PLACE $O/p$i.log
  python - $O/p$i <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))
if not f:
    print("  (no counter file)"); sys.exit(0)
rs = list(csv.DictReader(open(f[-1])))
did = lambda r: int(r.get("Dispatch_Id") or r.get("Dispatch_ID") or 0)
marks = sorted({did(r): r["Kernel_Name"] for r in rs if "k_diag_marker" in r["Kernel_Name"]})
if len(marks) < 3:
    print("  markers:", marks); sys.exit(0)
m1, m2, m3 = marks[-3], marks[-2], marks[-1]
acc = {"slow": collections.defaultdict(list), "fast": collections.defaultdict(list)}
for r in rs:
    if "k_bar_ohlcv_small" not in r["Kernel_Name"]:
        continue
    d = did(r)
    reg = "slow" if m1 < d < m2 else "fast" if m2 < d < m3 else None
    if reg:
        acc[reg][r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in sorted(acc["slow"]):
    s, q = acc["slow"][c], acc["fast"].get(c, [0.0])
    ms, mf = sum(s) / len(s), sum(q) / max(1, len(q))
    print(f"  {c:48s} slow {ms:14.5g}  fast {mf:14.5g}  slow/fast {ms / mf if mf else float('nan'):7.3f}   ({len(s)} / {len(q)} dispatches)")
PY
  rm -rf $O/p$i
done
