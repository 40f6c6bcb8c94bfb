#!/bin/bash
O=gpurun_out/cs1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cusum.py -q -x 2>&1 | tail -3 > $O/pytest.txt; cat $O/pytest.txt
timeout 600 python tools/cusumbench.py 1e9 1e-5 1e-4 > $O/bench_on.txt 2>&1; cat $O/bench_on.txt
FMK_CUSUM_CHAIN=0 timeout 900 python tools/fuzz_cusum.py 300 8803 300000 > $O/fuzz_nochain.txt 2>&1; tail -2 $O/fuzz_nochain.txt
bash tools/cs1_prof.sh
