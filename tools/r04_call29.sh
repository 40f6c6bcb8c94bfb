mkdir -p gpurun_out/c29
timeout 900 python -m pytest tests/test_gpu_fused.py tests/test_gpu_kits.py tests/test_gpu_quickstart.py -q -x 2>&1 | tail -3
for sm in 0 1 d; do echo "FMK_FLOW_SORT=$sm"; if [ $sm = d ]; then unset FMK_FLOW_SORT; else export FMK_FLOW_SORT=$sm; fi; timeout 300 python tools/realcfg4.py 1e9 1.0 2>&1 | tail -4 | head -1; timeout 300 python tools/realcfg4.py 1e9 1.0 dy 2>&1 | tail -4 | head -1; timeout 300 python tools/realcfg4.py 1e9 0 2>&1 | tail -4 | head -1; timeout 300 python tools/realcfg4.py 1e9 0.5 2>&1 | tail -4 | head -1; done > gpurun_out/c29/sort.txt 2>&1
cat gpurun_out/c29/sort.txt
