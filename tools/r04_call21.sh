mkdir -p gpurun_out/c21
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 > gpurun_out/c21/pytest_gpu_tail.txt
timeout 900 python bench.py > gpurun_out/c21/bench_default.json 2> gpurun_out/c21/bench_default.err
cp gpu_parity_counts.json gpurun_out/c21/ 2>/dev/null
tail -34 gpurun_out/c21/pytest_gpu_tail.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c21/bench_default.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("step %.3f kernel %.3f diff %.3f frac %.3f min %.3f max %.3f" % (d["ms_per_step"], r["avg_kernel_ms"], d["ms_per_step"]-r["avg_kernel_ms"], r["frac"], r.get("frac_min",0), r.get("frac_max",0)), r.get("placement"))
print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d["other_configs"].items() if k.endswith("_ms") or k=="error"})
PY
