#!/bin/bash
# why does TimeBarKit(39 M trades).build_ohlcv() read 16.0 ms inside bench.py and 14.7 ms in tools/apibench.py on its own?
O=gpurun_out/api; mkdir -p $O
timeout 300 python tools/apibench.py > $O/standalone.json 2>&1
timeout 600 python bench.py --steps 5 --warmup 2 --placed-probe 0 > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python bench.py --steps 5 --warmup 2 --placed-probe 0 --cpu-sample 0 > $O/bench_nocpu.json 2> $O/bench_nocpu.err
timeout 300 python tools/apibench.py > $O/standalone2.json 2>&1
python - <<'PY'
import json
for f in ("standalone", "standalone2"):
    d = json.loads(open(f"gpurun_out/api/{f}.json").read().strip().splitlines()[-1]); print(f, "warm %.2f" % d["warm_ms"], d["warm_all_ms"], "upload %.2f ms %.1f GB/s" % (d["upload_ms"], d["h2d_GBps"]))
for f in ("bench_default", "bench_nocpu"):
    d = json.loads(open(f"gpurun_out/api/{f}.json").read().strip().splitlines()[-1]); o = d["other_configs"]; print(f, "api warm %.2f cold %.2f h2d %.1f GB/s" % (o["api_39M_build_ohlcv_ms"], o["api_39M_build_ohlcv_cold_ms"], o["h2d_GBps"]))
PY
