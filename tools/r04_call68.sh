#!/bin/bash
mkdir -p gpurun_out/final
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/final/pytest_gpu_tail.txt
cp gpu_parity_counts.json gpurun_out/final/ 2>/dev/null
tail -3 gpurun_out/final/pytest_gpu_tail.txt
