R=$PWD
mkdir -p gpurun_out/c3
timeout 900 python -m pytest tests/test_gpu_timebars_fused.py tests/test_gpu_core.py tests/test_gpu_dist.py -q -x 2>&1 | tail -8 > gpurun_out/c3/pytest.txt
for sp in 8 0; do FMK_TB_PIPE_SPLIT=$sp timeout 300 python bench.py --no-extras --cpu-sample 0 --steps 20 --warmup 5 > gpurun_out/c3/bench_split$sp.json 2> gpurun_out/c3/bench_split$sp.err; done
for sp in 4 6 12 16; do FMK_TB_PIPE_SPLIT=$sp timeout 300 python bench.py --no-extras --cpu-sample 0 --steps 20 --warmup 5 --placements 1 > gpurun_out/c3/bench_p1_split$sp.json 2>/dev/null; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_tl
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_tl -o c -- env -C $R python bench.py --no-extras --cpu-sample 0 --steps 6 --warmup 3 --placements 1 > $R/gpurun_out/c3/bench_tl.json 2> $R/gpurun_out/c3/bench_tl.err
cd $R
python tools/rocpd_timeline.py $(find /tmp/prof_tl -name '*_results.db' | head -1) 40 > gpurun_out/c3/step_timeline.txt 2>&1
cat gpurun_out/c3/pytest.txt; tail -24 gpurun_out/c3/step_timeline.txt
for f in gpurun_out/c3/bench_*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[1], 'step %.3f kernel %.3f diff %.3f frac %.3f' % (d['ms_per_step'], r['avg_kernel_ms'], d['ms_per_step']-r['avg_kernel_ms'], r['frac']), r.get('launches_per_step'), r.get('placement'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
PY
done
tail -3 gpurun_out/c3/bench_split8.err
