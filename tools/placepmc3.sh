#!/bin/bash
# tools/placepmc3.py under counter passes of at most three TCC counters (more: "exceeds the capabilities of the hardware"); per position the mean
# duration (kernel trace) and the counters of the large k_bar_ohlcv_small launches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/placepmc3; mkdir -p $O
i=0
for P in "$@"; do
  i=$((i+1)); rm -rf $O/p$i
  timeout 800 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/p$i -o p -- env -C $R python tools/placepmc3.py 1e9 8 2 > $O/p$i.log 2>&1
  python - $O/p$i "$P" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
fc = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)); fk = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))
if not fc or not fk:
    print("  pass", sys.argv[2], ": no output"); sys.exit(0)
did = lambda r: int(r.get("Dispatch_Id") or r.get("Dispatch_ID") or 0)
dur = {}
for r in csv.DictReader(open(fk[-1])):
    dur[did(r)] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
marks = sorted(k for k, v in dur.items() if "k_diag_marker" in v[0])
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(fc[-1])):
    cnt[did(r)][r["Counter_Name"]] = float(r["Counter_Value"])
print("  pass:", sys.argv[2])
for a, b in zip(marks[:-1], marks[1:]):
    ds = [k for k, v in dur.items() if a < k < b and "k_bar_ohlcv_small" in v[0] and v[1] > 1_500_000]      # the 7/8 launches
    if not ds:
        continue
    ms = sum(dur[k][1] for k in ds) / len(ds) * 1e-6
    names = sorted(cnt[ds[0]])
    print(f"  position {marks.index(a)}: {ms:7.3f} ms  " + "  ".join(f"{c.replace('TCC_EA0_', '').replace('_sum', '')} {sum(cnt[k].get(c, 0.0) for k in ds) / len(ds):.5g}" for c in names))
PY
  rm -rf $O/p$i
done
