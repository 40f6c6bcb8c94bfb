#!/bin/bash
mkdir -p gpurun_out/c56
for m in 57 20; do echo "FMK_OHLCV_ROWS_MIN_MEAN=$m"; FMK_OHLCV_ROWS_MIN_MEAN=$m timeout 600 python tools/shortbars.py 1e9 1,1.3,1.7,2,2.5 2>&1; done | tee gpurun_out/c56/rowsmin.txt
