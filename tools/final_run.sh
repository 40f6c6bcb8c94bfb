R=$PWD
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/final/pytest_gpu_tail.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1
timeout 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
timeout 900 python bench.py --force-dist > gpurun_out/final/bench_forcedist.json 2> gpurun_out/final/bench_forcedist.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/final/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/final/prof -o b -- python $R/bench.py > $R/gpurun_out/final/bench_under_rocprof.json 2> $R/gpurun_out/final/bench_under_rocprof.err
cd $R
python tools/rocpd_stats.py $(find gpurun_out/final/prof -name "*.db" | head -1) > gpurun_out/final/bench_kernel_stats.csv
rm -rf gpurun_out/final/prof
cat gpurun_out/final/pytest_gpu_tail.txt gpurun_out/final/smoke.txt | tail -6
for f in bench_default bench_forcedist bench_under_rocprof; do python - $f <<'PY'
import json,sys
f=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/final/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], {k:v for k,v in d.get("other_configs",{}).items() if k.endswith("_ms")})
except Exception as e:
    print(f, "FAILED", e)
PY
done
head -8 gpurun_out/final/bench_kernel_stats.csv | cut -c1-140
# the two counter passes of the dominant kernel at THIS code (roofline.traffic): gpurun_out/final/traffic_constants.json is what
# goes to profiles/ afterwards
bash tools/pmc_calibrate.sh > gpurun_out/final/pmc_calibrate.log 2>&1
python tools/pmc_summarize.py > gpurun_out/final/pmc_summarize.log 2>&1 || tail -5 gpurun_out/final/pmc_summarize.log
rm -rf gpurun_out/pmc_r02_FETCH_SIZE gpurun_out/pmc_r02_WRITE_SIZE
python -c "import json; d = json.load(open('gpurun_out/final/traffic_constants.json')); print('traffic: read B/tick', d['read_bytes_per_tick'], 'write B/bar', d['write_bytes_per_bar'], d['kernel_source_sha256'][:12])"
