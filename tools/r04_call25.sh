mkdir -p gpurun_out/c25
for i in 1 2 3; do timeout 300 python tools/pipeab.py 1e9 20 3; echo "--"; done > gpurun_out/c25/pipeab.txt 2>&1
cat gpurun_out/c25/pipeab.txt
