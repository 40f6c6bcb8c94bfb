#!/bin/bash
echo "half rows up to 110 ticks (FMK_OHLCV_ROWS_MIN_MEAN=111)"; FMK_OHLCV_ROWS_MIN_MEAN=111 timeout 600 python tools/shortbars.py 1e9 4,4.5,5 2>&1 | grep "median=True"
echo "default"; timeout 600 python tools/shortbars.py 1e9 4,4.5,5 2>&1 | grep "median=True"
