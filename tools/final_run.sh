#!/bin/bash
# End-of-round evidence, ONE gpurun call at HEAD (everything lands in gpurun_out/final/, to be copied into profiles/ as rNN_*):
#   the full GPU suite, smoke, the default bench line, the --force-dist bench line, a rocprofv3 kernel trace of the SAME step
#   (--no-extras; the timed region = the last steps x 2 dispatches of the dominant kernel, summarised on its own), the two counter
#   passes of the dominant kernel (traffic_constants.json), and tools/cfgprof.sh: kernel stats + PMC traffic of every secondary config
#   (traffic_other_configs.json, rNN_<key>_kernel_stats.csv) -- so that no `traffic_stale` is left in the bench line the driver runs.
R=$PWD
mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/final/pytest_gpu_tail.txt
cp gpu_parity_counts.json gpurun_out/final/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1
# counters first: the bench lines below then read constants taken at this very code
bash tools/pmc_calibrate.sh > gpurun_out/final/pmc_calibrate.log 2>&1
python tools/pmc_summarize.py > gpurun_out/final/pmc_summarize.log 2>&1 || tail -5 gpurun_out/final/pmc_summarize.log
rm -rf gpurun_out/pmc_r02_FETCH_SIZE gpurun_out/pmc_r02_WRITE_SIZE
bash tools/cfgprof.sh 1e9 > gpurun_out/final/cfgprof.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
timeout 900 python bench.py --force-dist --steps 20 --warmup 5 > gpurun_out/final/bench_forcedist.json 2> gpurun_out/final/bench_forcedist.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_final
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o b -- env -C $R python bench.py --no-extras --cpu-sample 0 --placed-probe 0 --steps 20 --warmup 5 > $R/gpurun_out/final/bench_under_rocprof.json 2> $R/gpurun_out/final/bench_under_rocprof.err
cd $R
python tools/rocpd_stats.py $(find /tmp/prof_final -name "*.db" | head -1) > gpurun_out/final/bench_cfg2_kernel_stats.csv
python tools/rocpd_stats.py $(find /tmp/prof_final -name "*.db" | head -1) "k_bar_ohlcv_smallILb0ELb1ELi21" --last 40 > gpurun_out/final/bench_cfg2_timed_region_kernel_stats.csv
tail -4 gpurun_out/final/pytest_gpu_tail.txt; tail -2 gpurun_out/final/smoke.txt; grep CFGPROF gpurun_out/final/cfgprof.log
for f in bench_default bench_forcedist bench_under_rocprof; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/final/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, "step %.3f kernel %.3f diff %.3f frac %.3f launch avg %.4f" % (d["ms_per_step"], r["avg_kernel_ms"], d["ms_per_step"]-r["avg_kernel_ms"], r["frac"], r.get("avg_launch_ms", 0)), "STALE" if r.get("traffic_stale") else "", "placed %.3f" % r["placed"]["frac"] if r.get("placed", {}).get("frac") else "")
    oc = d.get("other_configs", {})
    print("   ", {k:(round(v,2) if isinstance(v,float) else v) for k,v in oc.items() if k.endswith("_ms")})
    print("   ", {k:(round(v["kernel_ms"],2), round(v["frac"],3), "STALE" if v.get("traffic_stale") else "") for k,v in (oc.get("roofline") or {}).items()})
except Exception as e:
    print(f, "FAILED", e)
PY
done
cat gpurun_out/final/bench_cfg2_timed_region_kernel_stats.csv | cut -c1-150
