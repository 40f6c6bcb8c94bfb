#!/bin/bash
mkdir -p gpurun_out/c75
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_features.py tests/test_gpu_kits.py tests/test_gpu_quickstart.py -m gpu -x -q 2>&1 | tail -3
for k in 1 0 1 0; do echo "FMK_FLOW_SIDE_OHLCV=$k"; FMK_FLOW_SIDE_OHLCV=$k timeout 300 python tools/cfg4bench.py 1e9 2>&1 | tail -2 | cut -c1-70; done | tee gpurun_out/c75/side.txt
