mkdir -p gpurun_out/c26
timeout 900 python -m pytest tests/test_gpu_timebars_fused.py tests/test_gpu_core.py tests/test_gpu_dist.py tests/test_gpu_refcalls.py -q -x 2>&1 | tail -3
for sc in 1 0; do echo "FMK_TIME_INDEX_SECANT=$sc"; FMK_TIME_INDEX_SECANT=$sc timeout 300 python tools/indexbench.py 2>&1 | grep interval; FMK_TIME_INDEX_SECANT=$sc timeout 300 python tools/pipeab.py 1e9 20 2; done > gpurun_out/c26/secant.txt 2>&1
cat gpurun_out/c26/secant.txt
