mkdir -p gpurun_out/c6
timeout 600 python -m pytest tests/test_gpu_threshold.py -q -x 2>&1 | tail -3
for d in 0 8 1 9 2 4 11 15; do echo "FMK_VX_DBG=$d"; FMK_VX_DBG=$d timeout 300 python tools/thrbench.py 1e9 600,865 volume 2>&1 | grep "mean bar"; done > gpurun_out/c6/vx_phases.txt 2>&1
cat gpurun_out/c6/vx_phases.txt
