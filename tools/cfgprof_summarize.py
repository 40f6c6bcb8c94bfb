#!/usr/bin/env python3
"""The offline half of bench.py's per-config roofline objects.  Input: what tools/cfgprof.sh left in gpurun_out/cfgprof/ for each
KEY -- a rocpd database of `rocprofv3 --kernel-trace --stats` (KEY_trace) and the counter CSVs of two `--pmc` runs (KEY_FETCH_SIZE,
KEY_WRITE_SIZE), each of the process `tools/cfgprof.py KEY`.  Only the dispatches between the two k_diag_marker kernels count.
Output: profiles/r06_KEY_kernel_stats.csv (per kernel: calls per call of the config, total / average ns) and
profiles/traffic_other_configs.json {KEY: {traffic_bytes (2 x FETCH_SIZE + WRITE_SIZE per call of the config, KiB counters,
profiles/pmc_calibration.txt), dominant_kernel, dominant_kernel_ms_summed, kernels_ms, device_ms, csrc_sha256}}."""
import csv, glob, json, os, re, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import csrc_sha256  # noqa: E402
D = os.path.join(ROOT, "gpurun_out", "cfgprof")


def trace(key):
    dbs = glob.glob(os.path.join(D, f"{key}_trace", "**", "*_results.db"), recursive=True)
    if not dbs:
        return None
    db = sqlite3.connect(dbs[-1]); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    sym = [t for t in tabs if "info_kernel_symbol" in t][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {sym} s on d.kernel_id=s.id order by d.start"))
    marks = [i for i, r in enumerate(rows) if "k_diag_marker" in r[0]]
    assert len(marks) == 2, (key, marks)
    return rows[marks[0] + 1:marks[1]]


def counter(key, name):
    fs = glob.glob(os.path.join(D, f"{key}_{name}", "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return None
    rs = list(csv.DictReader(open(fs[-1])))
    rs.sort(key=lambda r: int(r.get("Dispatch_Id") or r.get("Dispatch_ID") or 0))
    marks = [i for i, r in enumerate(rs) if "k_diag_marker" in r["Kernel_Name"]]
    assert len(marks) == 2, (key, name, marks)
    m = re.search(r"CFGPROF \S+ n=\d+ reps=(\d+)", open(os.path.join(D, f"{key}_{name}.log")).read())
    reps = int(m.group(1)) if m else 1
    return sum(float(r["Counter_Value"]) for r in rs[marks[0] + 1:marks[1]] if r["Counter_Name"] == name) * 1024.0 / reps


def main():
    out = {}
    try:
        out = json.load(open(os.path.join(ROOT, "profiles", "traffic_other_configs.json")))
    except (OSError, ValueError):
        pass
    sha = csrc_sha256()
    for log in sorted(glob.glob(os.path.join(D, "*_trace.log"))):
        key = os.path.basename(log)[:-len("_trace.log")]
        m = re.search(r"CFGPROF (\S+) n=(\d+) reps=(\d+) device_ms=([\d.]+)", open(log).read())
        rows = trace(key)
        if not m or rows is None:
            print("skip", key); continue
        reps = int(m.group(3))
        by = {}
        for name, st, en in rows:
            by.setdefault(name, []).append(en - st)
        os.makedirs(os.path.join(ROOT, "gpurun_out", "final"), exist_ok=True)
        for dst in (os.path.join(ROOT, "profiles"), os.path.join(ROOT, "gpurun_out", "final")):
          with open(os.path.join(dst, f"r06_{key}_kernel_stats.csv"), "w") as fh:
            fh.write(f'# tools/cfgprof.py {key} {m.group(2)} {reps} under rocprofv3 --kernel-trace --stats: the dispatches of {reps} calls of the config (between the k_diag_marker kernels); device time per call by HIP events in the same run: {m.group(4)} ms\n')
            fh.write('"Name","CallsPerConfigCall","TotalDurationNs","AverageNs","MsPerConfigCall"\n')
            for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
                fh.write(f'"{name}",{len(v) / reps:g},{sum(v)},{sum(v) / len(v):.1f},{sum(v) / reps * 1e-6:.4f}\n')
        dom = max(by.items(), key=lambda kv: sum(kv[1]))
        short = re.sub(r"\(.*", "", dom[0]).replace(".kd", "")
        f, w = counter(key, "FETCH_SIZE"), counter(key, "WRITE_SIZE")
        out[key] = {"traffic_bytes": 2.0 * f + w if f is not None and w is not None else None,
                    "fetch_size_bytes_raw": f, "write_size_bytes": w,
                    "dominant_kernel": short, "dominant_kernel_ms_summed": sum(dom[1]) / reps * 1e-6,
                    "kernels_ms": sum(sum(v) for v in by.values()) / reps * 1e-6, "device_ms": float(m.group(4)),
                    "n_ticks": int(m.group(2)), "csrc_sha256": sha}
        print(key, json.dumps(out[key]))
    json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_other_configs.json"), "w"), indent=1)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "final"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "final", "traffic_other_configs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
