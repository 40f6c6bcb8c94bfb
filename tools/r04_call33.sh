mkdir -p gpurun_out/c33
for t in 2048 1024; do echo "FMK_OHLCV_LANES_TILE=$t"; FMK_OHLCV_LANES_TILE=$t timeout 300 python tools/shortbars.py 1e9 1.7,2,2.5,3 2>&1 | grep "median=True" | cut -c1-130; done > gpurun_out/c33/tile.txt 2>&1
cat gpurun_out/c33/tile.txt
timeout 600 python -m pytest tests/test_gpu_core.py tests/test_gpu_barlengths.py -q -x 2>&1 | tail -2
