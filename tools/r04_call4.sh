R=$PWD
mkdir -p gpurun_out/c4
timeout 900 python -m pytest tests/test_gpu_threshold.py tests/test_gpu_timebars_fused.py -q -x 2>&1 | tail -8 > gpurun_out/c4/pytest.txt
cat gpurun_out/c4/pytest.txt
for s in 101 102 103; do timeout 600 python tools/fuzz_volume.py $s 300 3000000 volume 2>&1 | tail -3; done > gpurun_out/c4/fuzz_volume.txt
cat gpurun_out/c4/fuzz_volume.txt
for on in 1 2 0; do echo "FMK_VOL_EXACT_TIER=$on"; FMK_VOL_EXACT_TIER=$on timeout 300 python tools/thrbench.py 1e9 300,600,865,1200,1400 volume; done > gpurun_out/c4/thrbench.txt 2>&1
cat gpurun_out/c4/thrbench.txt
for cfg in "8 2" "8 1" "8 4" "8 0" "6 2" "4 2" "12 2" "0 2"; do set -- $cfg; for rep in 1 2; do
FMK_TB_PIPE_SPLIT=$1 FMK_TB_PIPE_IDX_BPC=$2 timeout 300 python bench.py --no-extras --cpu-sample 0 --steps 20 --warmup 5 --placements 1 2>gpurun_out/c4/err.txt | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('split $1 idx_bpc $2: step %.3f kernel %.3f diff %.3f frac %.3f lps %s' % (d['ms_per_step'], r['avg_kernel_ms'], d['ms_per_step']-r['avg_kernel_ms'], r['frac'], r.get('launches_per_step')))
except Exception as e: print('split $1 idx_bpc $2 FAILED', e)
"; done; done > gpurun_out/c4/pipe_sweep.txt 2>&1
cat gpurun_out/c4/pipe_sweep.txt; tail -3 gpurun_out/c4/err.txt
