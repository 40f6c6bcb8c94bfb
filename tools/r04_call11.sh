mkdir -p gpurun_out/c11
bash tools/prof.sh c11/cfg4_lognormal python tools/realcfg4.py 1e9 1.0 > gpurun_out/c11/cfg4_lognormal.txt 2>&1
bash tools/prof.sh c11/cfg4_uniform python tools/realcfg4.py 1e9 0 > gpurun_out/c11/cfg4_uniform.txt 2>&1
head -30 gpurun_out/c11/cfg4_lognormal.txt | cut -c1-180; head -24 gpurun_out/c11/cfg4_uniform.txt | cut -c1-180
