#!/bin/bash
# wider fuzz of the CUSUM tiers after the one-pass form went in: who answered, and 0 failures expected
O=gpurun_out/cs1; mkdir -p $O
FMK_CUSUM_CHAIN=0 timeout 1500 python tools/fuzz_cusum.py 1200 8811 1000000 > $O/fuzz_a.txt 2>&1; tail -3 $O/fuzz_a.txt
timeout 1200 python tools/fuzz_cusum.py 600 8812 1000000 > $O/fuzz_b.txt 2>&1; tail -3 $O/fuzz_b.txt
FMK_CUSUM_CHAIN=0 timeout 1200 python tools/fuzz_cusum.py 120 8813 20000000 > $O/fuzz_c.txt 2>&1; tail -3 $O/fuzz_c.txt
