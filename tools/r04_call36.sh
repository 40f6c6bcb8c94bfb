#!/bin/bash
# sharded step through the pipelined one-call entry: parity, then the force-dist and plain bench lines
mkdir -p gpurun_out/c36
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_timebars_fused.py tests/test_gpu_ohlcv.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/c36/pytest.txt
timeout 400 python bench.py --force-dist > gpurun_out/c36/forcedist.json 2> gpurun_out/c36/forcedist.err
timeout 400 python bench.py > gpurun_out/c36/default.json 2> gpurun_out/c36/default.err
cat gpurun_out/c36/pytest.txt; cat gpurun_out/c36/forcedist.json; cat gpurun_out/c36/default.json
