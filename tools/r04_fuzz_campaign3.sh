# round-4 fuzz campaign over the code that changed this round (volume exact-sum tier, dollar block-trade walk, pipelined time-bar step,
# secant indexer, order-flow tie bound): every line ends "N failures"
mkdir -p gpurun_out/fuzz
{
for s in 501 502 503 504 505 506; do timeout 900 python tools/fuzz_volume.py $s 400 3000000 volume 2>&1 | tail -1; done
for s in 511 512 513 514; do timeout 900 python tools/fuzz_volume.py $s 300 3000000 dollar 2>&1 | tail -1; done
for s in 521 522 523 524; do timeout 900 python tools/fuzz_whales.py $s 300 3000000 2>&1 | tail -1; done
for s in 531 532 533 534; do timeout 1200 python tools/fuzz_parity.py $s 2500 2>&1 | tail -1; done
timeout 900 python tools/fuzz_longbars.py 200 541 2>&1 | tail -1
timeout 900 python tools/fuzz_sharded.py 2>&1 | tail -2
} > gpurun_out/fuzz/r04_campaign3.txt 2>&1
cat gpurun_out/fuzz/r04_campaign3.txt | cut -c1-220
{
for s in 551 552 553 554; do timeout 1200 python tools/fuzz_longbars.py 100 $s short 2>&1 | tail -1; done
for s in 561 562; do timeout 1200 python tools/fuzz_longbars.py 120 $s mid 2>&1 | tail -1; done
for s in 5 6; do timeout 900 python tools/fuzz_sharded.py 60 $s 2>&1 | tail -1; done
} >> gpurun_out/fuzz/r04_campaign3.txt 2>&1
tail -12 gpurun_out/fuzz/r04_campaign3.txt | cut -c1-200
