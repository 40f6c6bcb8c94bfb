mkdir -p gpurun_out/c30
timeout 1200 python -m pytest tests/test_gpu_fused.py tests/test_gpu_core.py tests/test_gpu_barlengths.py tests/test_gpu_f32amounts.py tests/test_gpu_refcalls.py tests/test_gpu_volume_profile.py -q -x 2>&1 | tail -3
for ss in 1 0; do echo "FMK_FP_SIDE_STREAM=$ss"; export FMK_FP_SIDE_STREAM=$ss; timeout 300 python tools/realcfg4.py 1e9 1.0 2>&1 | tail -4 | head -2; timeout 300 python tools/realcfg4.py 1e9 0 2>&1 | tail -4 | head -2; timeout 400 python tools/intervalbench.py 1e9 600 3600 86400 2>&1 | grep interval | cut -c1-250; done > gpurun_out/c30/side.txt 2>&1
cat gpurun_out/c30/side.txt
timeout 600 python tools/fuzz_longbars.py 60 4501 2>&1 | tail -1
timeout 600 python tools/widebench.py 2>&1 | tail -6 | cut -c1-200
