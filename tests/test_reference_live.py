"""CPU, build container only: the oracle against the REFERENCE ITSELF on a fixed-seed slice of tools/fuzz_reference.py's campaign.
Skipped where /root/reference does not exist (the GPU box; a user's checkout) -- the committed fixtures are what pins the oracle
there; this test is the live cross-check while the reference is at hand."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/finmlkit"), reason="the reference is only present in the build container")
def test_oracle_against_the_reference_on_random_cases():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_reference.py"), "600", "20260929", "3000"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " 0 failures" in r.stdout, tail
