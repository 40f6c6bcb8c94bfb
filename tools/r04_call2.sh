R=$PWD
mkdir -p gpurun_out/c2
for v in "1 1" "0 1" "1 0" "0 0"; do timeout 300 python tools/placecheck.py 1e9 3 $v >> gpurun_out/c2/placecheck.txt 2>&1; echo "--" >> gpurun_out/c2/placecheck.txt; done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_tl
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/prof_tl -o c -- env -C $R python bench.py --no-extras --cpu-sample 0 --steps 6 --warmup 3 --placements 1 > $R/gpurun_out/c2/bench_tl.json 2> $R/gpurun_out/c2/bench_tl.err
cd $R
python tools/rocpd_timeline.py $(find /tmp/prof_tl -name '*_results.db' | head -1) 60 > gpurun_out/c2/step_timeline.txt 2>&1
cat gpurun_out/c2/placecheck.txt; tail -40 gpurun_out/c2/step_timeline.txt
