#!/bin/bash
# differential fuzz of every kernel family at HEAD, fresh seeds; each tool exits non-zero on a failure
O=gpurun_out/soak; mkdir -p $O; rc=0
run() { name=$1; shift; timeout 1500 "$@" > $O/$name.txt 2>&1; r=$?; echo "$name rc=$r: $(tail -1 $O/$name.txt | cut -c1-200)"; [ $r -ne 0 ] && rc=1; }
run parity        python tools/fuzz_parity.py 3000 9101 20000
run volume        python tools/fuzz_volume.py 9102 500 300000 volume
run dollar        python tools/fuzz_volume.py 9103 500 300000 dollar
run whales        python tools/fuzz_whales.py 9104 150 300000
run fused         python tools/fuzz_fused.py 300 9105
run longbars      python tools/fuzz_longbars.py 60 9106
run longbars_mid  python tools/fuzz_longbars.py 60 9107 mid
run sharded       python tools/fuzz_sharded.py 40 9108
run cusum         python tools/fuzz_cusum.py 500 9109 1000000
FMK_CUSUM_CHAIN=0 run cusum_nochain python tools/fuzz_cusum.py 500 9110 1000000
echo "SOAK rc=$rc"
