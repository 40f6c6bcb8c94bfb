mkdir -p gpurun_out/c12
for cfg in "4 0" "1 0" "4 1" "1 1" "1 2"; do set -- $cfg; echo "FMK_DIR_WPB=$1 FMK_FP_WPB=$2"; FMK_DIR_WPB=$1 FMK_FP_WPB=$2 timeout 300 python tools/realcfg4.py 1e9 1.0 2>&1 | tail -4; done > gpurun_out/c12/wpb.txt 2>&1
echo "uniform:" >> gpurun_out/c12/wpb.txt
for cfg in "4 0" "1 1"; do set -- $cfg; echo "FMK_DIR_WPB=$1 FMK_FP_WPB=$2"; FMK_DIR_WPB=$1 FMK_FP_WPB=$2 timeout 300 python tools/realcfg4.py 1e9 0 2>&1 | tail -4; done >> gpurun_out/c12/wpb.txt 2>&1
cat gpurun_out/c12/wpb.txt
