#!/bin/bash
mkdir -p gpurun_out/c67
FMK_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --ticks 200000000 --no-extras --cpu-sample 0 --steps 5 > gpurun_out/c67/two_ranks.json 2> gpurun_out/c67/two_ranks.err; echo "rc $?"
tail -3 gpurun_out/c67/two_ranks.err | cut -c1-300
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/c67/two_ranks.json').read().strip().splitlines()[-1])
    print('n_gpus', d['n_gpus'], 'step', d['ms_per_step'], d['config']['parallelism'][:60], d['per_rank']['ms_per_step'], d['config'].get('n_bars_total'))
except Exception as e: print('no line', e)
PY
FMK_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --ticks 200000000 --no-extras --cpu-sample 0 --steps 5 > gpurun_out/c67/torchrun.json 2> gpurun_out/c67/torchrun.err; echo "rc $?"
tail -2 gpurun_out/c67/torchrun.json | cut -c1-400
