#!/bin/bash
mkdir -p gpurun_out/c64
timeout 1800 python -m pytest tests/test_gpu_features.py tests/test_gpu_fused.py tests/test_gpu_volume_profile.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/cfg4bench.py 1e9 2>&1 | tail -2 | cut -c1-120
timeout 600 python tools/fpbench.py 1000000000 2>&1 | tail -4 | cut -c1-200
