mkdir -p gpurun_out/c23
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_barlengths.py tests/test_gpu_f32amounts.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -3
timeout 900 python tools/fuzz_longbars.py 60 4401 2>&1 | tail -3
timeout 900 python tools/fuzz_parity.py 4402 800 2>&1 | tail -3
timeout 600 python tools/intervalbench.py 1e9 60 3600 86400 2>&1 | grep interval | cut -c1-260
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "long_bars or full_mantissa" 2>&1 | tail -3
