mkdir -p gpurun_out/c27
timeout 900 python -m pytest tests/test_gpu_fused.py -q -x 2>&1 | tail -3
for sm in 0 1; do echo "FMK_FLOW_SORT=$sm"; FMK_FLOW_SORT=$sm timeout 300 python tools/realcfg4.py 1e9 1.0 2>&1 | tail -4; done > gpurun_out/c27/sort.txt 2>&1
echo "uniform:" >> gpurun_out/c27/sort.txt
for sm in 0 1; do echo "FMK_FLOW_SORT=$sm"; FMK_FLOW_SORT=$sm timeout 300 python tools/realcfg4.py 1e9 0 2>&1 | tail -4 | head -1; done >> gpurun_out/c27/sort.txt 2>&1
cat gpurun_out/c27/sort.txt
FMK_FLOW_SORT=1 bash tools/prof.sh c27/sorted python tools/realcfg4.py 1e9 1.0 2>&1 | tail -16 | cut -c1-150
