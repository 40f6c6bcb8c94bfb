"""How much each full-size parity test actually compared: `record()` collects (test, config) -> counts, the
session hook in conftest.py prints them after the `-q` dots (so they are in the tail the driver keeps) and writes them to
gpu_parity_counts.json at the repository root and under gpurun_out/ (what a gpurun call brings back)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTS = {}


def record(name, **counts):
    COUNTS[name] = {k: (int(v) if isinstance(v, (int, bool)) or hasattr(v, "__index__") else v) for k, v in counts.items()}


def dump():
    if not COUNTS:
        return None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "gpu_parity_counts.json"), "w") as fh:
                json.dump(COUNTS, fh, indent=1, sort_keys=True)
        except OSError:
            pass
    return COUNTS
