#!/usr/bin/env python3
"""gpurun_out/pmc_r02_{FETCH_SIZE,WRITE_SIZE}/p_counter_collection.csv (tools/pmc_calibrate.sh) -> profiles/traffic_constants.json
(what bench.py reads for roofline.traffic) + profiles/pmc_calibration.txt (the calibration table).  The constants carry the SHA-256
of the dominant kernel's sources (bench.kernel_source_sha256): bench.py marks them `stale` when the sources have changed since.  On
the GPU box (no .git, nothing but gpurun_out/ comes back) a copy goes to gpurun_out/final/ for the round's hand-over."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_TICKS = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
PROBE_BYTES = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4_000_000_000


sys.path.insert(0, ROOT)
from bench import kernel_source_sha256  # noqa: E402


def rows(counter):
    f = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_r02_{counter}", "**", "*counter_collection.csv"), recursive=True))[-1]
    return list(csv.DictReader(open(f)))


def value(rs, key):
    v = [float(r["Counter_Value"]) for r in rs if key in r["Kernel_Name"]]
    assert len(v) == 1, (key, v)
    return v[0] * 1024.0                                  # the counters are in KiB


f, w = rows("FETCH_SIZE"), rows("WRITE_SIZE")
probes = {"16 B/lane loads": value(f, "k_diag_read<0>"), "8 B/lane loads": value(f, "k_diag_read<1>"),
          "4 B/lane loads": value(f, "k_diag_read4"), "8 B/lane stores (WRITE_SIZE)": value(w, "k_diag_write8")}
fetch_factor = {k: PROBE_BYTES / v for k, v in probes.items() if "loads" in k}
write_factor = PROBE_BYTES / probes["8 B/lane stores (WRITE_SIZE)"]
dom_fetch, dom_write = value(f, "k_bar_ohlcv_small"), value(w, "k_bar_ohlcv_small")
nb = [int(r["Grid_Size"]) for r in f if "k_time_bar_index" in r["Kernel_Name"]]
bars = 833323 if N_TICKS == 1_000_000_000 else None
ff = sum(fetch_factor.values()) / len(fetch_factor)
commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
const = {
    "kernel": "k_bar_ohlcv_small<false, true> (fused OHLCV + median, float32 amounts)",
    "n_ticks": N_TICKS, "n_bars": bars, "commit": commit or "(measured on the GPU box: see kernel_source_sha256)",
    "kernel_source_sha256": kernel_source_sha256(),
    "fetch_size_correction": round(ff, 4), "write_size_correction": round(write_factor, 4),
    "fetch_size_correction_by_width": {k: round(v, 4) for k, v in fetch_factor.items()},
    "read_bytes_per_launch": dom_fetch * ff, "write_bytes_per_launch": dom_write * write_factor,
    "read_bytes_per_tick": dom_fetch * ff / N_TICKS, "write_bytes_per_bar": dom_write * write_factor / bars if bars else None,
    "source": "tools/pmc_calibrate.sh: two rocprofv3 passes (--kernel-trace --pmc FETCH_SIZE, --pmc WRITE_SIZE), ONE launch each; "
              "corrections measured in the same passes on 4e9-byte probes (profiles/pmc_calibration.txt)",
}
json.dump(const, open(os.path.join(ROOT, "profiles", "traffic_constants.json"), "w"), indent=1)
os.makedirs(os.path.join(ROOT, "gpurun_out", "final"), exist_ok=True)
json.dump(const, open(os.path.join(ROOT, "gpurun_out", "final", "traffic_constants.json"), "w"), indent=1)
for cal in (os.path.join(ROOT, "profiles", "pmc_calibration.txt"), os.path.join(ROOT, "gpurun_out", "final", "pmc_calibration.txt")):
  with open(cal, "w") as fh:
      fh.write("# rocprofv3 FETCH_SIZE / WRITE_SIZE against known byte counts on gfx950 (tools/pmc_calibrate.py: one launch of each\n"
               f"# probe over a {PROBE_BYTES}-byte buffer, two separate counter passes).  counter (bytes) and bytes / counter:\n")
      for k, v in probes.items():
          fh.write(f"{k:32s} {v:16.0f}  x{PROBE_BYTES / v:.4f}\n")
      fh.write("# -> FETCH_SIZE counts exactly half of a coalesced read stream at 16, 8 AND 4 bytes per lane (the guide documents\n"
               "#    16 B/lane); WRITE_SIZE counts coalesced 8 B/lane stores in full.\n"
               f"# dominant kernel at N = {N_TICKS}: FETCH_SIZE {dom_fetch:.0f} B -> {dom_fetch * ff:.6g} B read, WRITE_SIZE {dom_write:.0f} B\n"
               f"#   -> traffic {dom_fetch * ff + dom_write * write_factor:.6g} B per launch (algorithmic 12 N + 76 B + 8 = {12 * N_TICKS + 76 * (bars or 0) + 8:.6g})\n")
print(json.dumps(const, indent=1))
