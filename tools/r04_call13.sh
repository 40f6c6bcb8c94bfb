mkdir -p gpurun_out/c13
for fl in 1 0; do echo "FMK_FLOW_LANES=$fl"; FMK_FLOW_LANES=$fl timeout 300 python tools/realcfg4.py 1e9 1.0 2>&1 | tail -4 | head -1; done > gpurun_out/c13/flow.txt 2>&1
FMK_FLOW_LANES=0 bash tools/prof.sh c13/cfg4_ln_fused python tools/realcfg4.py 1e9 1.0 >> gpurun_out/c13/flow.txt 2>&1
cat gpurun_out/c13/flow.txt | cut -c1-200 | head -30
