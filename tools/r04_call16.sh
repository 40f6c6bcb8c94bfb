mkdir -p gpurun_out/c16
for s in 11 12 13 14; do timeout 900 python tools/fuzz_whales.py $s 250 2000000 2>&1 | tail -4; done > gpurun_out/c16/fuzz_whales.txt 2>&1
cat gpurun_out/c16/fuzz_whales.txt | cut -c1-260
{ timeout 900 python tools/whalebench.py 1e9 1e-4 1000 1 2>&1 | tail -3
timeout 900 python tools/whalebench.py 1e9 1e-4 10000 1 2>&1 | tail -3
timeout 900 python tools/whalebench.py 1e9 1e-3 3000 1 2>&1 | tail -3
FMK_DL_WHALE_TIER=0 timeout 900 python tools/whalebench.py 1e8 1e-4 1000 0 2>&1 | tail -2; } > gpurun_out/c16/whalebench.txt 2>&1
cat gpurun_out/c16/whalebench.txt
