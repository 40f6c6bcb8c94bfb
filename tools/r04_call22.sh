mkdir -p gpurun_out/c22
for l in finmlkit_amd/lib/ab/libfmk_hip_r3.so finmlkit_amd/lib/libfmk_hip.so; do echo "== $l"; timeout 600 python tools/ab_lib.py $l tools/intervalbench.py 1e9 3600 86400 2>&1 | grep interval; done > gpurun_out/c22/daily_ab.txt 2>&1
cat gpurun_out/c22/daily_ab.txt | cut -c1-260
