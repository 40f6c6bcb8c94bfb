mkdir -p gpurun_out/c14
for rm in 65 33 41 49; do echo "FMK_OHLCV_ROWS_MIN_MEAN=$rm"; FMK_OHLCV_ROWS_MIN_MEAN=$rm timeout 300 python tools/shortbars.py 1e9 2,2.5,3,4,5,7.5,10,15 2>&1 | grep "median=True"; done > gpurun_out/c14/rows.txt 2>&1
cat gpurun_out/c14/rows.txt | cut -c1-150
