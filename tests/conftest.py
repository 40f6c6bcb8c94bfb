import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """The full-size parity tests say how many bars / closes / levels they compared: printed after the dots (also with -q) and
    written to gpu_parity_counts.json (tests/_counts.py)."""
    from tests import _counts
    c = _counts.dump()
    if c:
        terminalreporter.write_line("gpu_parity_counts (also in gpu_parity_counts.json):")
        for k in sorted(c):
            terminalreporter.write_line("  %s: %s" % (k, ", ".join("%s=%s" % kv for kv in sorted(c[k].items()))))


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """A fresh checkout has no finmlkit_amd/lib/libfmk_hip.so (build products are git-ignored): build it once
    (hipcc cross-compiles gfx950 without a GPU).  The product itself never builds or falls back on its own."""
    lib = os.path.join(ROOT, "finmlkit_amd", "lib", "libfmk_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
    yield


@pytest.fixture(scope="session")
def orc():
    """The CPU parity oracle (oracle/fmk_oracle.c through ctypes)."""
    from oracle import oracle as o
    o.build()
    return o
