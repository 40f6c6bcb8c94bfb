# end-of-round evidence run (one gpurun call): full GPU suite, smoke, default bench, --force-dist bench, a --no-extras rocprofv3 kernel
# trace of the SAME step (default placement probes; the timed region = the last 40 dispatches of the dominant kernel, summarised on its own
# next to the whole-run summary), the two counter passes for roofline.traffic
R=$PWD
mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/final/pytest_gpu_tail.txt
cp gpu_parity_counts.json gpurun_out/final/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
timeout 900 python bench.py --force-dist --steps 20 --warmup 5 > gpurun_out/final/bench_forcedist.json 2> gpurun_out/final/bench_forcedist.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_final
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o b -- env -C $R python bench.py --no-extras --cpu-sample 0 --steps 20 --warmup 5 > $R/gpurun_out/final/bench_under_rocprof.json 2> $R/gpurun_out/final/bench_under_rocprof.err
cd $R
python tools/rocpd_stats.py $(find /tmp/prof_final -name "*.db" | head -1) > gpurun_out/final/bench_cfg2_kernel_stats.csv
python tools/rocpd_stats.py $(find /tmp/prof_final -name "*.db" | head -1) "k_bar_ohlcv_smallILb0ELb1ELi21" --last 40 > gpurun_out/final/bench_cfg2_timed_region_kernel_stats.csv
bash tools/pmc_calibrate.sh > gpurun_out/final/pmc_calibrate.log 2>&1
python tools/pmc_summarize.py > gpurun_out/final/pmc_summarize.log 2>&1 || tail -5 gpurun_out/final/pmc_summarize.log
rm -rf gpurun_out/pmc_r02_FETCH_SIZE gpurun_out/pmc_r02_WRITE_SIZE
tail -24 gpurun_out/final/pytest_gpu_tail.txt; cat gpurun_out/final/smoke.txt | tail -2
for f in bench_default bench_forcedist bench_under_rocprof; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/final/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, "step %.3f kernel %.3f diff %.3f frac %.3f (min %s max %s) launch avg %.4f" % (d["ms_per_step"], r["avg_kernel_ms"], d["ms_per_step"]-r["avg_kernel_ms"], r["frac"], r.get("frac_min"), r.get("frac_max"), r.get("avg_launch_ms", 0)), "stale" if r.get("traffic_stale") else "")
    print("   ", {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.get("other_configs",{}).items() if k.endswith("_ms")})
except Exception as e:
    print(f, "FAILED", e)
PY
done
head -6 gpurun_out/final/bench_cfg2_kernel_stats.csv | cut -c1-150; cat gpurun_out/final/bench_cfg2_timed_region_kernel_stats.csv | cut -c1-150
python -c "import json; d = json.load(open('gpurun_out/final/traffic_constants.json')); print('traffic: read B/tick', d['read_bytes_per_tick'], 'write B/bar', d['write_bytes_per_bar'], d['kernel_source_sha256'][:12])"
