mkdir -p gpurun_out/c35
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_core.py tests/test_gpu_refcalls.py tests/test_gpu_f32amounts.py tests/test_gpu_barlengths.py tests/test_gpu_fuzz.py tests/test_gpu_kits.py tests/test_gpu_volume_profile.py -q -x 2>&1 | tail -3
timeout 900 python tools/fuzz_parity.py 1500 4601 2>&1 | tail -2
timeout 300 python tools/realcfg4.py 1e9 0 2>&1 | tail -4
timeout 300 python tools/realcfg4.py 1e9 0 dy 2>&1 | tail -4 | head -2
timeout 300 python tools/realcfg4.py 1e9 1.0 2>&1 | tail -4 | head -2
