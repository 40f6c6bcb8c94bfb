#!/bin/bash
mkdir -p gpurun_out/c74
timeout 1500 python -m pytest tests/test_gpu_fused.py tests/test_gpu_features.py tests/test_gpu_kits.py -m gpu -x -q 2>&1 | tail -3
for k in 1 0 1 0; do echo "FMK_FLOW_SIDE_OHLCV=$k"; FMK_FLOW_SIDE_OHLCV=$k timeout 300 python tools/realcfg4.py 1e9 1.0 2>&1 | grep "cfg 4"; FMK_FLOW_SIDE_OHLCV=$k timeout 300 python tools/realcfg4.py 1e9 1.0 dyadic 2>&1 | grep "cfg 4"; done | tee gpurun_out/c74/side.txt
