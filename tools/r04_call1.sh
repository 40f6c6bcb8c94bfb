mkdir -p gpurun_out/c1
bash tools/partition_probe.sh probe gpurun_out/c1/partition_probe.txt > /dev/null 2>&1
bash tools/drift_ab.sh 6 gpurun_out/c1/r04_drift.txt > /dev/null 2>&1
timeout 600 python tools/stepvar.py 1e9 4 20 > gpurun_out/c1/stepvar.txt 2>&1
timeout 900 python bench.py > gpurun_out/c1/bench_default.json 2> gpurun_out/c1/bench_default.err
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/c1/pytest_gpu_tail.txt
tail -12 gpurun_out/c1/r04_drift.txt; tail -8 gpurun_out/c1/stepvar.txt; cat gpurun_out/c1/pytest_gpu_tail.txt; tail -30 gpurun_out/c1/partition_probe.txt
