#!/bin/bash
mkdir -p gpurun_out/c53
for m in 65 45; do echo "FMK_OHLCV_ROWS_MIN_MEAN=$m"; FMK_OHLCV_ROWS_MIN_MEAN=$m timeout 600 python tools/shortbars.py 1e9 2,2.5,3 2>&1 | grep "median=True"; done | tee gpurun_out/c53/rowsmin.txt
