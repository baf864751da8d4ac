"""A short run of the randomised differential parity driver (tools/fuzz_parity.py) as a GPU test: random robot shapes, horizons, gaits,
VO rates / latencies and solver switches, device against oracle at every tick; a case over the tolerance is arbitrated against the
exact optimum of the oracle's own QP and fails only if the DEVICE is the inaccurate one."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_a_dozen_random_configurations_against_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "12", "6", "120"], capture_output=True, text=True, timeout=400)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0, (lines[-3:], r.stderr[-1500:])
    import json
    last = json.loads(lines[-1])
    assert last["cases_run"] == 12 and last["all_passed_or_explained"] is True
