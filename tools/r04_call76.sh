#!/bin/bash
mkdir -p gpurun_out/c76
timeout 1800 python -m pytest tests/test_gpu_fused.py tests/test_gpu_features.py tests/test_gpu_kits.py tests/test_gpu_quickstart.py tests/test_gpu_core.py -m gpu -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "cfg4" 2>&1 | tail -3
for x in 1 0; do echo "FMK_FLOW_SIDE_OHLCV=$x"; FMK_FLOW_SIDE_OHLCV=$x timeout 400 python tools/intervalbench.py 1e9 600 3600 86400 2>&1 | grep interval | cut -c1-40,150-240; done | tee gpurun_out/c76/long.txt
timeout 300 python tools/realcfg4.py 1e9 1.0 2>&1 | grep "cfg 4"
