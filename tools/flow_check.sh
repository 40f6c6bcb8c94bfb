#!/bin/bash
# after the order-flow tie test learnt about bars of mixed sign: the feature tests, the fused fuzz, and cfg 4 / order-flow timings
O=gpurun_out/flow; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_features.py tests/test_gpu_fused.py -q -x 2>&1 | tail -4 | tee $O/pytest.txt
timeout 900 python tools/fuzz_fused.py 400 7720 > $O/fuzz_fused.txt 2>&1; tail -2 $O/fuzz_fused.txt
timeout 900 python tools/fuzz_longbars.py 60 372 > $O/fuzz_longbars.txt 2>&1; tail -2 $O/fuzz_longbars.txt
timeout 900 python bench.py --steps 5 --warmup 2 --placed-probe 0 --cpu-sample 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
o = json.loads(open("gpurun_out/flow/bench.json").read().strip().splitlines()[-1])["other_configs"]
print({k: round(v, 2) for k, v in o.items() if k.startswith(("cfg4", "directional")) and k.endswith("_ms")})
PY
