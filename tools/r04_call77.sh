#!/bin/bash
# the three bench lines again, with the traffic constants measured at HEAD in place (the end-of-round script measures them AFTER its bench runs)
R=$PWD; mkdir -p gpurun_out/final
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
for i in 1 2 3; do timeout 900 python bench.py --force-dist --steps 20 --warmup 5 --no-extras --cpu-sample 0 > gpurun_out/final/bench_forcedist_$i.json 2>> gpurun_out/final/bench_forcedist.err; done
timeout 900 python bench.py --force-dist --steps 20 --warmup 5 > gpurun_out/final/bench_forcedist.json 2>> gpurun_out/final/bench_forcedist.err
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_final
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o b -- env -C $R python bench.py --no-extras --cpu-sample 0 --steps 20 --warmup 5 > $R/gpurun_out/final/bench_under_rocprof.json 2> $R/gpurun_out/final/bench_under_rocprof.err
cd $R
python tools/rocpd_stats.py $(find /tmp/prof_final -name "*.db" | head -1) > gpurun_out/final/bench_cfg2_kernel_stats.csv
python tools/rocpd_stats.py $(find /tmp/prof_final -name "*.db" | head -1) "k_bar_ohlcv_smallILb0ELb1ELi21" --last 40 > gpurun_out/final/bench_cfg2_timed_region_kernel_stats.csv
for f in bench_default bench_forcedist_1 bench_forcedist_2 bench_forcedist_3 bench_forcedist bench_under_rocprof; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
d=json.loads(open(f"gpurun_out/final/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]; pl=r.get("placement",{})
print(f, "step %.3f kernel %.3f diff %.3f frac %.3f best probe %.3f launch avg %.4f" % (d["ms_per_step"], r["avg_kernel_ms"], d["ms_per_step"]-r["avg_kernel_ms"], r["frac"], min(pl.get("probe_kernel_ms",[0])), r.get("avg_launch_ms",0)), "STALE" if r.get("traffic_stale") else "")
PY
done
cat gpurun_out/final/bench_cfg2_timed_region_kernel_stats.csv | cut -c1-120
