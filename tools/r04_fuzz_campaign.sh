# round-4 fuzz campaign over the code that changed this round (volume exact-sum tier, dollar block-trade walk, pipelined time-bar step,
# secant indexer, order-flow tie bound): every line ends "N failures"
mkdir -p gpurun_out/fuzz
{
for s in 301 302 303 304 305 306 307 308; do timeout 900 python tools/fuzz_volume.py $s 400 3000000 volume 2>&1 | tail -1; done
for s in 311 312 313 314 315 316; do timeout 900 python tools/fuzz_volume.py $s 300 3000000 dollar 2>&1 | tail -1; done
for s in 321 322 323 324 325 326 327 328; do timeout 900 python tools/fuzz_whales.py $s 300 3000000 2>&1 | tail -1; done
for s in 331 332 333; do timeout 1200 python tools/fuzz_parity.py $s 2500 2>&1 | tail -1; done
timeout 900 python tools/fuzz_longbars.py 150 341 2>&1 | tail -1
timeout 900 python tools/fuzz_sharded.py 2>&1 | tail -2
} > gpurun_out/fuzz/r04_campaign.txt 2>&1
cat gpurun_out/fuzz/r04_campaign.txt | cut -c1-220
