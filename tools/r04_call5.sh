mkdir -p gpurun_out/c5
bash tools/prof.sh c5/vol865 python tools/thrbench.py 1e9 865 volume > gpurun_out/c5/vol865.txt 2>&1
FMK_VOL_EXACT_TIER=0 bash tools/prof.sh c5/vol865_old python tools/thrbench.py 1e9 865 volume > gpurun_out/c5/vol865_old.txt 2>&1
bash tools/prof.sh c5/vol600 python tools/thrbench.py 1e9 600 volume > gpurun_out/c5/vol600.txt 2>&1
cat gpurun_out/c5/vol865.txt gpurun_out/c5/vol865_old.txt gpurun_out/c5/vol600.txt
