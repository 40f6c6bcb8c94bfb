#!/bin/bash
mkdir -p gpurun_out/c59
echo "quarter rows 16..48"; FMK_OHLCV_QUARTER_MIN_MEAN=16 FMK_OHLCV_QUARTER_MAX_MEAN=48 timeout 600 python tools/shortbars.py 1e9 1,1.3,1.7,2,2.3 2>&1 | grep "median=True"
echo "default"; timeout 600 python tools/shortbars.py 1e9 1,1.3,1.7,2,2.3,3,4 2>&1 | grep "median=True"
FMK_OHLCV_QUARTER_MIN_MEAN=16 FMK_OHLCV_QUARTER_MAX_MEAN=48 timeout 1200 python tools/fuzz_longbars.py 60 411 short 2>&1 | tail -2
