#!/bin/bash
# the one-pass CUSUM form: tests, the 1e9-tick bench with the form on / off (closes and checksum must agree), fuzz with and without the chain tier
O=gpurun_out/cs1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cusum.py -q -x 2>&1 | tail -6 > $O/pytest.txt; cat $O/pytest.txt
timeout 600 python tools/cusumbench.py 1e9 1e-5 1e-4 5e-4 > $O/bench_on.txt 2>&1; cat $O/bench_on.txt
FMK_CUSUM_ONEPASS=0 timeout 600 python tools/cusumbench.py 1e9 1e-5 1e-4 > $O/bench_off.txt 2>&1; grep sigma_floor $O/bench_off.txt
FMK_CUSUM_CHAIN=0 timeout 600 python tools/cusumbench.py 1e9 5e-4 > $O/bench_nochain.txt 2>&1; tail -4 $O/bench_nochain.txt
timeout 900 python tools/fuzz_cusum.py 300 8801 300000 > $O/fuzz_default.txt 2>&1; tail -3 $O/fuzz_default.txt
FMK_CUSUM_CHAIN=0 timeout 900 python tools/fuzz_cusum.py 300 8802 300000 > $O/fuzz_nochain.txt 2>&1; tail -3 $O/fuzz_nochain.txt
